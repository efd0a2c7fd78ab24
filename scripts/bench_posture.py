#!/usr/bin/env python3
"""Posture leg only (features kernel -> MARS CNN -> keypoint scatter) on the live tracks of a batch:
used under rocprofv3 for the CNN's kernel list and MFMA counters (profiles/README.md)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch
from mmwave_msc_amd.mars import MarsCNN, random_keras_weights

S, N, T, F = int(os.environ.get("S", 4096)), 512, 8, 8
POP = os.environ.get("POP", "full")   # the e2e leg of bench.py runs SURVEY 8(d)'s K = T population (31.7 k samples); POP=mixed: 1 + s mod 8 targets (18.3 k)
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=1, population=POP)
dev = torch.device("cuda:0")
sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
side = torch.cuda.Stream(); torch.cuda.set_stream(side); sb.follow_torch_stream(side)  # one stream for torch and the context
for f in range(F):
    sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
cap = S * 16
feat = torch.empty((cap, 3, 8, 8, 5), dtype=torch.float32, device=dev)
owner = torch.empty((cap, 2), dtype=torch.int32, device=dev)
model = MarsCNN.from_keras_weights(random_keras_weights(0, 3)).to(dev)
def it():
    n = sb.features_dev(feat.data_ptr(), owner.data_ptr(), cap)
    with torch.no_grad():
        kp = model(feat[:n])
    sb.set_keypoints_dev(kp.data_ptr(), owner.data_ptr(), n)
    return n
for _ in range(2): n = it()
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 10
for _ in range(K): n = it()
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / K
print(f"posture leg ({POP} population): {n} samples, {el*1e3:.3f} ms/iter, {n/el:.0f} samples/s, CNN {n*25.187328e6/el/1e12:.2f} TFLOP/s of 157.3 fp32-MFMA peak")
