#!/usr/bin/env python3
"""Experiment: one context of S scenes vs C contexts of S/C scenes (own streams) on one GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch

S, N, T, W, K = 4096, 512, 8, 10, 40
F = W + K
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=32)
dev = torch.device("cuda:0")
d_pts = torch.from_numpy(pts).to(dev).double()
d_cnt = torch.from_numpy(cnt).to(dev)
d_dt = torch.from_numpy(dts).to(dev)
for C in (1, 2, 4):
    Sc = S // C
    sbs = [SceneBatch(_lib.default_config(tr_max_tracks=T), Sc, N) for _ in range(C)]
    outs = [(torch.empty((Sc, N), dtype=torch.int32, device=dev), torch.empty((Sc, sbs[0].UM), dtype=torch.int32, device=dev),
             torch.empty((Sc,), dtype=torch.int32, device=dev)) for _ in range(C)]
    def step(f):
        for c, sb in enumerate(sbs):
            a, l, n = outs[c]
            sb.step_dev(d_pts[f, c * Sc:(c + 1) * Sc].data_ptr(), d_cnt[f, c * Sc:(c + 1) * Sc].data_ptr(),
                        d_dt[f, c * Sc:(c + 1) * Sc].data_ptr(), a.data_ptr(), l.data_ptr(), n.data_ptr())
    for f in range(W):
        step(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(W, F):
        step(f)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for sb in sbs:
        sb.check()
    print(f"{C} context(s): {S * K / el / 1e6:.2f} M scene-frames/s, {el / K * 1e3:.4f} ms/step")
    for sb in sbs:
        sb.close()
