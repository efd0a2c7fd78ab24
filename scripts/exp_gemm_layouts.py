#!/usr/bin/env python3
"""Experiment: layouts / fusions of the split Dense-1 GEMMs (M = 18304, N = 1536, K = 6144, fp16 in, fp32 out)."""
import os, time, torch
dev = torch.device("cuda:0")
M, N, K = 18304, 1536, 6144
S = 2048.0
torch.manual_seed(0)
a2 = (torch.randn(M, 2 * K, device=dev) * 0.5).half()
W = torch.randn(K, N, device=dev) * 0.02
w_hi = W.half(); w_lo = ((W - w_hi.float()) * S).half()
w2 = torch.cat([w_lo, w_hi], 0).contiguous()
bias = torch.randn(N, device=dev)
w_hi_t = w_hi.t().contiguous(); w2_t = w2.t().contiguous()          # [N, K] / [N, 2K]: K contiguous
w3 = torch.cat([w_hi * S, w_lo, w_hi], 0).contiguous()                # single GEMM: [a_hi | a_hi | a_lo'] @ w3 / S
a3 = torch.cat([a2[:, :K], a2], 1).contiguous()
w3_t = w3.t().contiguous()

def tm(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

def two_nn():
    g1 = torch.addmm(bias, a2[:, :K], w_hi, out_dtype=torch.float32)
    return torch.addmm(g1, a2, w2, out_dtype=torch.float32, alpha=1.0 / S)
def two_nt():
    g1 = torch.addmm(bias, a2[:, :K], w_hi_t.t(), out_dtype=torch.float32)
    return torch.addmm(g1, a2, w2_t.t(), out_dtype=torch.float32, alpha=1.0 / S)
def one_nn():
    return torch.addmm(bias, a3, w3, out_dtype=torch.float32, alpha=1.0 / S)
def one_nt():
    return torch.addmm(bias, a3, w3_t.t(), out_dtype=torch.float32, alpha=1.0 / S)
ref = (a2[:, :K].double() + a2[:, K:].double() / S)[:256] @ (w_hi.double() + w_lo.double() / S) + bias.double()
for name, f in (("two GEMMs, W [K,N]", two_nn), ("two GEMMs, W^T [N,K]", two_nt), ("one GEMM K'=3K, W [K,N]", one_nn), ("one GEMM K'=3K, W^T", one_nt)):
    out = f()
    err = float((out[:256].double() - ref).abs().max())
    print(f"{name:28s} {tm(f):.3f} ms   max err vs fp64 {err:.2e}")

print("preferred BLAS library by default:", torch.backends.cuda.preferred_blas_library())
for lib in ("cublas", "cublaslt", "ck"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
        for name, f in (("two GEMMs, W [K,N]", two_nn), ("two GEMMs, W^T [N,K]", two_nt)):
            out = f()
            err = float((out[:256].double() - ref).abs().max())
            print(f"{lib:9s} {name:28s} {tm(f):.3f} ms   max err vs fp64 {err:.2e}")
    except Exception as e:
        print(lib, "failed:", str(e)[:200])
torch.backends.cuda.preferred_blas_library("default")

# the same with PyTorch's TunableOp picking among the hipBLASLt / rocBLAS solutions
import torch.cuda.tunable as tun
tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(int(os.environ.get('TUNE_MS', '3000'))); tun.set_max_tuning_iterations(100)
tun.set_filename("gpurun_out/tunableop_dense1.csv")
t0 = time.perf_counter()
for name, f in (("two GEMMs, W [K,N]", two_nn), ("two GEMMs, W^T [N,K]", two_nt)):
    out = f()
    err = float((out[:256].double() - ref).abs().max())
    print(f"tuned {name:28s} {tm(f):.3f} ms   max err vs fp64 {err:.2e}   (tuning so far {time.perf_counter() - t0:.0f} s)")
for r in tun.get_results(): print(r)

# the two GEMMs side by side on two streams, merged by one elementwise pass
tun.enable(False)
s2 = torch.cuda.Stream(device=dev)
def two_concurrent():
    cur = torch.cuda.current_stream(dev)
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        g2 = torch.mm(a2, w2_t.t(), out_dtype=torch.float32)
    g1 = torch.addmm(bias, a2[:, :K], w_hi_t.t(), out_dtype=torch.float32)
    cur.wait_stream(s2)
    g2.record_stream(cur)
    return torch.relu_(g1.add_(g2, alpha=1.0 / S))
def two_serial_relu():
    return torch.relu_(two_nt())
a = two_concurrent(); b = two_serial_relu()
print("concurrent == serial:", bool(torch.equal(a, b)), float((a - b).abs().max()))
for name, f in (("serial + relu", two_serial_relu), ("two streams + add + relu", two_concurrent)):
    print(f"{name:28s} {tm(f, 20):.3f} ms")
