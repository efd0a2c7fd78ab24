#!/usr/bin/env python3
"""Experiment: the 4096-scene batch as L independent contexts ("lanes") on L streams, stepped interleaved, so one lane's
kernel ramps and tails overlap the other lanes' kernels.  Prints ms per frame of all scenes for L = 1, 2, 4."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch

S, N, T, W, K = 4096, 512, 8, 10, 40
F = W + K
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=8)
dev = torch.device("cuda:0")
d_pts = torch.empty((F, S, N, 8), dtype=torch.float64, device=dev)
for f in range(F):
    d_pts[f] = torch.from_numpy(pts[f]).to(dev).double()
d_cnt = torch.from_numpy(cnt).to(dev); d_dt = torch.from_numpy(dts).to(dev)
side_modes = (0, 1, -1)
for L in (1, 2, 4):
    for side in side_modes:
        per = S // L
        lanes = []
        for l in range(L):
            sb = SceneBatch(_lib.default_config(tr_max_tracks=T, chain_side_stream=side), per, N, device=0)
            st = torch.cuda.Stream(device=dev)
            sb.follow_torch_stream(st)
            a = torch.empty((per, N), dtype=torch.int32, device=dev); lab = torch.empty((per, sb.UM), dtype=torch.int32, device=dev)
            dbn = torch.empty((per,), dtype=torch.int32, device=dev)
            lanes.append((sb, st, a, lab, dbn))
        torch.cuda.synchronize()
        def step(f):
            for l, (sb, st, a, lab, dbn) in enumerate(lanes):
                lo = l * per
                sb.step_dev(d_pts[f, lo:lo + per].data_ptr(), d_cnt[f, lo:lo + per].data_ptr(), d_dt[f, lo:lo + per].data_ptr(),
                            a.data_ptr(), lab.data_ptr(), dbn.data_ptr())
        for f in range(W): step(f)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for f in range(W, F): step(f)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / K * 1e3
        for sb, *_ in lanes: sb.check(); sb.close()
        print(f"lanes {L}  side_stream {side:2d}: {ms:.4f} ms/frame  {S / ms / 1e3:.2f} M scene-frames/s", flush=True)
