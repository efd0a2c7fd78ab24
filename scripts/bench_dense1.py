#!/usr/bin/env python3
"""Dense-1 of the split-fp16 CNN alone: the fused kernel (k_dense.hip) against the formulation it replaced -- two hipBLASLt
GEMMs (hi.W_hi and [hi | lo'].[W_lo' ; W_hi]) side by side on two streams + a merge pass."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_msc_amd.mars import MarsCNN, SPLIT_SCALE, deinterleave_split, random_keras_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 18304
dev = torch.device("cuda:0")
mk = MarsCNN.from_keras_weights(random_keras_weights(0, 3)).to(dev)
x = torch.randn((B, 3, 8, 8, 5), device=dev)
side = torch.cuda.Stream(device=dev)
with torch.no_grad():
    a2 = mk._hip_convs_split(x)
    hi, lo = deinterleave_split(a2)
    acat = torch.cat([hi, lo], 1).contiguous()
    w_hi_t, w_lo_t = (t.contiguous() for t in deinterleave_split(mk.d1_w2_t))      # (N, K), K contiguous
    w2_t = torch.cat([w_lo_t, w_hi_t], 1).contiguous()
    bias = mk.dense1_dhwc.bias

    def two_gemms():
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            g2 = torch.mm(acat, w2_t.t(), out_dtype=torch.float32)
        g1 = torch.addmm(bias, acat[:, : hi.shape[1]], w_hi_t.t(), out_dtype=torch.float32)
        cur.wait_stream(side)
        return torch.relu_(g1.add_(g2, alpha=1.0 / SPLIT_SCALE))

    for name, fn in (("k_mars_dense1", lambda: mk._dense1_split(a2)), ("two GEMMs + merge", two_gemms),
                     ("conv pair", lambda: mk._hip_convs_split(x)), ("whole CNN", lambda: mk(x))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
        fl = 3 * 2.0 * B * 6144 * 1536
        print(f"{name:22s} {ms:7.3f} ms" + (f"   {fl / ms / 1e9:7.1f} TFLOP/s issued" if "CNN" not in name and "conv" not in name else ""))
    print("max |kernel - gemm| =", (mk._dense1_split(a2) - two_gemms()).abs().max().item())
