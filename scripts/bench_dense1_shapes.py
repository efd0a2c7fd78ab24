#!/usr/bin/env python3
"""Dense-1 of the split arithmetic at the shapes that exercise each tile kernel: (frames, rows) -> ms per launch.
   3 x 31 744: 256 x 192 tiles only; 3 x 18 304: + a tail of 128 x 128 tiles; 1 x 31 744 (define_CNN: K 2048, N 512): 256 x 128 tiles (+ tail)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
dev = torch.device("cuda:0")
for frames, B in ((3, 31744), (3, 18304), (1, 31744), (1, 18304), (3, 8500)):
    mk = MarsCNN.from_keras_weights(random_keras_weights(0, frames)).to(dev)
    x = torch.randn((B, 3, 8, 8, 5) if frames == 3 else (B, 8, 8, 5), device=dev)
    with torch.no_grad():
        a2 = mk._hip_convs_split(x)
        for _ in range(5): mk._dense1_split(a2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): mk._dense1_split(a2)
        torch.cuda.synchronize()
    print(f"frames={frames} rows={B}: {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms")
