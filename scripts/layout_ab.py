"""A/B of the step layouts on mid-size contexts: the library's choice vs track-wise Kalman kernels + DBSCAN workers on the side stream
(vs the one-workgroup step where it applies).  python scripts/layout_ab.py S [S ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch

N, T = 512, 8
F, W = int(os.environ.get('F', 120)), int(os.environ.get('W', 20))
dev = torch.device("cuda:0")
for S in [int(a) for a in sys.argv[1:]] or [640, 768, 896]:
    pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=16)
    P = torch.from_numpy(pts).to(dev).double(); C = torch.from_numpy(cnt).to(dev); D = torch.from_numpy(dts).to(dev)
    for rnd in range(2):
        for name, kw in (("default", {}), ("track-wise + side stream", dict(kalman_dense_min_units=1, chain_side_stream=1)),
                         ("track-wise", dict(kalman_dense_min_units=1, chain_side_stream=-1)), ("one workgroup", dict(fused_step=1))):
            try:
                sb = SceneBatch(_lib.default_config(tr_max_tracks=T, **kw), S, N)
            except Exception as e:
                print(f"S={S} {name}: {e}"); continue
            st = torch.cuda.Stream(); sb.follow_torch_stream(st)
            torch.cuda.synchronize()
            for f in range(W):
                sb.step_dev(P[f].data_ptr(), C[f].data_ptr(), D[f].data_ptr())
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for f in range(W, F):
                sb.step_dev(P[f].data_ptr(), C[f].data_ptr(), D[f].data_ptr())
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / (F - W) * 1e3
            print(f"S={S} {name:26s}: {ms:.4f} ms per step; kind {sb.step_kind()} kalman {sb.kalman_layout()} side {sb.side_workers()} tracks {int(sb.num_tracks().sum())}")
            sb.check(); sb.close()
