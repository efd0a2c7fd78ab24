#!/usr/bin/env python3
"""Diagnostic: k_features alone (feature map of every live track of 4096 scenes), wall time per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch
S, N, T, F = int(os.environ.get("S", 4096)), 512, 8, 8
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=16)
dev = torch.device("cuda:0")
sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
side = torch.cuda.Stream(); torch.cuda.set_stream(side); sb.follow_torch_stream(side)
for f in range(F):
    sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
cap = S * 16
feat = torch.empty((cap, 3, 8, 8, 5), dtype=torch.float32, device=dev)
owner = torch.empty((cap, 2), dtype=torch.int32, device=dev)
for _ in range(3): n = sb.features_dev(feat.data_ptr(), owner.data_ptr(), cap)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): n = sb.features_dev(feat.data_ptr(), owner.data_ptr(), cap)
torch.cuda.synchronize()
print(f"{os.environ.get('MMW_LIB_NAME', 'libmmw_hip.so')}: {n} rows, {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per call (count + scan + k_features + the row count's read-back)")
