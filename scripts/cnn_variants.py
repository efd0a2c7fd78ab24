import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
dev = torch.device("cuda:0")
B = 18000
x = torch.randn(B, 3, 8, 8, 5, device=dev)
model = MarsCNN.from_keras_weights(random_keras_weights(0, 3)).to(dev)
def timeit(fn, K=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
with torch.no_grad():
    hv = x.permute(0, 4, 1, 2, 3)
    hc = hv.contiguous()
    print("permute->contiguous copy: %.3f ms" % timeit(lambda: hv.contiguous()))
    print("conv1 on permuted view  : %.3f ms" % timeit(lambda: model.conv1(hv)))
    print("conv1 on contiguous     : %.3f ms" % timeit(lambda: model.conv1(hc)))
    h1 = torch.relu(model.conv1(hc))
    print("conv2 (from contiguous) : %.3f ms" % timeit(lambda: model.conv2(h1)))
    h1v = torch.relu(model.conv1(hv))
    print("conv2 (from view path)  : %.3f ms  strides %s" % (timeit(lambda: model.conv2(h1v)), h1v.stride()))
    h2 = torch.relu(model.conv2(h1)).flatten(1)
    print("dense1                  : %.3f ms" % timeit(lambda: model.dense1(h2)))
    print("full model (view)       : %.3f ms" % timeit(lambda: model(x)))
    # conv as unfold + GEMM (im2col) for conv1
    w1 = model.conv1.weight.reshape(16, -1)  # [16, 5*27]
    def conv1_gemm():
        p = torch.nn.functional.pad(hc, (1, 1, 1, 1, 1, 1))
        cols = p.unfold(2, 3, 1).unfold(3, 3, 1).unfold(4, 3, 1)  # B,5,3,8,8,3,3,3
        cols = cols.permute(0, 2, 3, 4, 1, 5, 6, 7).reshape(B * 192, 135)
        return (cols @ w1.t()).view(B, 3, 8, 8, 16)
    print("conv1 as im2col GEMM    : %.3f ms" % timeit(conv1_gemm))
