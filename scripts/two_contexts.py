"""Experiment: 4096 scenes as ONE context vs TWO contexts of 2048 scenes on two streams of the same GPU (the tail of one context's
launches filled by the other's): ms per step of all 4096 scenes.  python scripts/two_contexts.py [G ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch

S, N, T, F, W = 4096, 512, 8, 120, 20
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=16)
dev = torch.device("cuda:0")
P = torch.from_numpy(pts).to(dev)           # [F, S, N, 8] fp32
Pd = P.double()
C = torch.from_numpy(cnt).to(dev); D = torch.from_numpy(dts).to(dev)
for G in [int(a) for a in sys.argv[1:]] or [1, 2, 1, 2, 4]:
    per = S // G
    ctxs = []
    for g in range(G):
        sb = SceneBatch(_lib.default_config(tr_max_tracks=T), per, N)
        st = torch.cuda.Stream()
        sb.follow_torch_stream(st)
        ctxs.append((sb, st, g * per))
    torch.cuda.synchronize()
    def step(f):
        for sb, st, lo in ctxs:
            sb.step_dev(Pd[f, lo:lo + per].data_ptr(), C[f, lo:lo + per].data_ptr(), D[f, lo:lo + per].data_ptr())
    for f in range(W):
        step(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(W, F):
        step(f)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (F - W) * 1e3
    ntr = sum(int(sb.num_tracks().sum()) for sb, _, _ in ctxs)
    print(f"G={G}: {ms:.4f} ms per step of {S} scenes = {S / ms / 1e3:.2f} M scene-frames/s; tracks {ntr}; side workers {[sb.side_workers() for sb, _, _ in ctxs]}; kinds {[sb.step_kind() for sb, _, _ in ctxs]}")
    for sb, _, _ in ctxs:
        sb.check(); sb.close()
