#!/usr/bin/env python3
"""Exploration: throughput of the posture leg (features kernel + MARS CNN) on the live tracks of a batch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch
from mmwave_msc_amd.mars import MarsCNN, random_keras_weights

S, N, T, F = int(os.environ.get("S", 4096)), 512, 8, 12
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=16)
dev = torch.device("cuda:0")
sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
side = torch.cuda.Stream(); torch.cuda.set_stream(side); sb.follow_torch_stream(side)  # one stream for torch and the context
for f in range(F):
    sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
ntr = int(sb.num_tracks().sum())
print("live tracks", ntr)
cap = S * 16
feat = torch.empty((cap, 3, 8, 8, 5), dtype=torch.float32, device=dev)
owner = torch.empty((cap, 2), dtype=torch.int32, device=dev)
model = MarsCNN.from_keras_weights(random_keras_weights(0, 3)).to(dev)
torch.backends.cudnn.benchmark = True
def it():
    n = sb.features_dev(feat.data_ptr(), owner.data_ptr(), cap)
    with torch.no_grad():
        kp = model(feat[:n])
    sb.set_keypoints_dev(kp.data_ptr(), owner.data_ptr(), n)
    return n
for _ in range(3): n = it()
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 10
for _ in range(K): n = it()
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / K
print(f"posture leg: {n} samples, {el*1e3:.3f} ms/iter, {n/el:.0f} samples/s, CNN {n*25.19e6/el/1e12:.2f} TFLOP/s")
# pieces
def timeit(fn, K=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K
x = feat[:n]
print("features only: %.3f ms" % (timeit(lambda: sb.features_dev(feat.data_ptr(), owner.data_ptr(), cap)) * 1e3))
with torch.no_grad():
    print("cnn only: %.3f ms" % (timeit(lambda: model(x)) * 1e3))
    h = x.permute(0, 4, 1, 2, 3)
    print("conv1: %.3f ms" % (timeit(lambda: torch.relu(model.conv1(h))) * 1e3))
    h1 = torch.relu(model.conv1(h))
    print("conv2: %.3f ms" % (timeit(lambda: torch.relu(model.conv2(h1))) * 1e3))
    h2 = torch.relu(model.conv2(h1)).flatten(1)
    print("dense1: %.3f ms" % (timeit(lambda: torch.relu(model.dense1(h2))) * 1e3))
    xcl = h.contiguous(memory_format=torch.channels_last_3d)
    m2 = model.to(memory_format=torch.channels_last_3d)
    print("cnn channels_last_3d: %.3f ms" % (timeit(lambda: m2(x)) * 1e3))
