// Micro-benchmark (diagnostic, not part of the product): latency of lu6_inverse_cols for one wave,
// alone on a CU and with 16 waves per CU, in s_memtime ticks and in wall-clock ns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mmwave_msc_amd/csrc/mmw_math.hpp"
using namespace mmw;

__global__ void k_lu(const double *in, double *out, unsigned long long *ticks, int reps)
{
    const int lane = threadIdx.x & 63;
    const int c = lane & 15;
    double v[6], det = 0, acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < reps; it++) {
#pragma unroll
        for (int i = 0; i < 6; i++) v[i] = c < 6 ? in[(it & 7) * 36 + i * 6 + c] + acc * 1e-300 : 0.0;
        lu6_inverse_cols(v, lane, det);
        acc += v[0] + det;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main()
{
    double h[8 * 36];
    for (int m = 0; m < 8; m++)
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) h[m * 36 + i * 6 + j] = (i == j ? 4.0 + m : 0.0) + 0.1 * ((i * 7 + j * 3 + m) % 5);
    double *din, *dout; unsigned long long *dt;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 4096 * 256 * 8); hipMalloc(&dt, 4096 * 8);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    const int reps = 200;
    struct { int grid, block; const char *what; } cfgs[] = {{1, 64, "1 wave on the chip"}, {256, 64, "1 wave per CU"}, {256, 256, "4 waves per CU"}, {1024, 256, "16 waves per CU"}, {2048, 256, "32 waves per CU (2 rounds?)"}};
    for (auto &c : cfgs) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k_lu, dim3(c.grid), dim3(c.block), 0, 0, din, dout, dt, reps);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_lu, dim3(c.grid), dim3(c.block), 0, 0, din, dout, dt, reps);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long t; hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
        printf("%-28s: %8.1f ticks/LU  %8.1f ns/LU (kernel %.3f ms)\n", c.what, (double)t / reps, ms * 1e6 / reps, ms);
    }
    return 0;
}
