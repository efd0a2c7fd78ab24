#!/usr/bin/env python3
"""Time the split-fp16 conv kernel (mmw_mars_conv_split) alone: B samples, ms per launch, algorithmic TFLOP/s."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_msc_amd.mars import MarsCNN, random_keras_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 18304
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
m = MarsCNN.from_keras_weights(random_keras_weights(0, frames)).to(dev)
x = torch.randn((B, 3, 8, 8, 5) if frames == 3 else (B, 8, 8, 5), device=dev)
for _ in range(3):
    m._hip_convs_split(x)
torch.cuda.synchronize()
t = time.perf_counter()
n = 20
for _ in range(n):
    m._hip_convs_split(x)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / n * 1e3
flop = (0.83e6 + 5.31e6) if frames == 3 else (0.83e6 + 5.31e6) / 9  # conv1 + conv2 per sample (3-D)
print(f"conv16 frames={frames} B={B}: {ms:.4f} ms/launch, {B * flop / ms / 1e9:.1f} algorithmic TFLOP/s")
if frames == 3:
    for _ in range(3):
        m._hip_convs(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        m._hip_convs(x)
    torch.cuda.synchronize()
    print(f"fp32-MFMA kernel: {(time.perf_counter() - t) / n * 1e3:.4f} ms/launch")
