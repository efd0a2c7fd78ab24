#!/usr/bin/env python3
"""Experiment: Dense-1 of the MARS CNN (6144 -> 1536, 75 % of the CNN's MACs) as a split-fp16 GEMM on the fp16 matrix
cores with fp32 accumulation -- a = hi + 2^-11 lo', three of the four partial products -- against the plain fp32 GEMM:
speed, and error of the 57 outputs against the fp64 oracle."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
from oracle.mars_np import mars_forward_np

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 18304
w = random_keras_weights(0, 3)
model = MarsCNN.from_keras_weights(w).to(dev)
rng = np.random.default_rng(0)
feat = rng.normal(0, 0.5, size=(B, 3, 8, 8, 5)).astype(np.float32)
feat[:, :, :, :, 4] = rng.normal(-0.3, 0.4, size=(B, 3, 8, 8))
feat[rng.random(size=(B, 3, 8, 8)) < 0.3] = 0.0          # zero-padded rows
x = torch.from_numpy(feat).to(dev)
S = 2048.0  # 2^11

def split(t):
    hi = t.half()
    lo = ((t - hi.float()) * S).half()
    return hi, lo

with torch.no_grad():
    act = model._hip_convs(x)                      # (B, 6144) fp32
    W = model.dense1_dhwc.weight.t().contiguous()  # (6144, 1536)
    bias = model.dense1_dhwc.bias
    W_hi, W_lo = split(W)
    W2 = torch.cat([W_lo, W_hi], 0).contiguous()   # (2K, N)

    def d1_fp32():
        return torch.addmm(bias, act, W)

    def d1_split():
        a_hi, a_lo = split(act)
        g1 = torch.mm(a_hi, W_hi, out_dtype=torch.float32)
        a2 = torch.cat([a_hi, a_lo], 1)
        g2 = torch.mm(a2, W2, out_dtype=torch.float32)
        return g1 + g2 * (1.0 / S) + bias

    def d1_split_pre(a_hi, a2):
        g1 = torch.mm(a_hi, W_hi, out_dtype=torch.float32)
        g2 = torch.mm(a2, W2, out_dtype=torch.float32)
        return g1 + g2 * (1.0 / S) + bias

    def d1_fp16_plain():
        return torch.mm(act.half(), W_hi, out_dtype=torch.float32) + bias

    def tm(f, n=10):
        for _ in range(3): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

    ref32 = d1_fp32()
    sp = d1_split()
    pl = d1_fp16_plain()
    ref64 = (act[:512].double() @ W.double() + bias.double())
    print("dense1 out scale", float(ref64.abs().mean()), float(ref64.abs().max()))
    print("fp32  vs fp64 max", float((ref32[:512].double() - ref64).abs().max()))
    print("split vs fp64 max", float((sp[:512].double() - ref64).abs().max()))
    print("fp16  vs fp64 max", float((pl[:512].double() - ref64).abs().max()))
    a_hi, a_lo = split(act); a2 = torch.cat([a_hi, a_lo], 1).contiguous()
    print("ms fp32 %.3f  split(all) %.3f  split(gemms only) %.3f  fp16 plain %.3f" % (tm(d1_fp32), tm(d1_split), tm(lambda: d1_split_pre(a_hi, a2)), tm(d1_fp16_plain)))
    flop = 2.0 * B * 6144 * 1536
    print("TFLOP/s fp32 %.1f  split-gemms(3x) %.1f" % (flop / tm(d1_fp32) / 1e9, 3 * flop / tm(lambda: d1_split_pre(a_hi, a2)) / 1e9))
    # end-to-end keypoints with the split Dense-1
    import torch.nn.functional as F
    kp32 = model(x[:512]).double().cpu().numpy()
    kps = model.dense2(F.relu(d1_split()[:512])).double().cpu().numpy()
    want = mars_forward_np(w, feat[:512].astype(np.float64))
    print("keypoints: fp32 path max err %.3e   split Dense-1 max err %.3e" % (np.abs(kp32 - want).max(), np.abs(kps - want).max()))
