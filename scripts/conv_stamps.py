#!/usr/bin/env python3
"""Diagnostic (STAMPS build): cycles of wave 0 / block 0 of k_mars_conv16 per phase."""
import os, sys
os.environ["MMW_LIB_NAME"] = "libmmw_hip_stamps.so"
import ctypes as C
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_msc_amd import _lib
from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
B = 18304
dev = torch.device("cuda:0")
m = MarsCNN.from_keras_weights(random_keras_weights(0, 3)).to(dev)
x = torch.randn((B, 3, 8, 8, 5), device=dev)
L = _lib.load()
fn = L.mmw_diag_conv_stamps
fn.argtypes = [C.c_void_p, C.c_int]
out = np.zeros(8, dtype=np.uint64)
m._hip_convs_split(x); torch.cuda.synchronize()
fn(out.ctypes.data, 1)
m._hip_convs_split(x); torch.cuda.synchronize()
fn(out.ctypes.data, 0)
names = ["set-up", "input staging", "conv1", "conv2 MFMA loops", "conv2 epilogue+stores"]
samples = (B + 1023) // 1024
tot = float(out[:5].sum())
print("launches", int(out[7]), "samples of the wave ~", samples, "total cycles", tot)
for i, n in enumerate(names):
    print(f"  {n:24s} {float(out[i]):12.0f} cyc  {100 * float(out[i]) / tot:5.1f} %   per sample {float(out[i]) / samples:9.0f}")
