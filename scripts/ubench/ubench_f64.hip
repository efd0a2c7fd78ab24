// Micro-benchmark (diagnostic, not part of the product): fp64 VALU on gfx950 as the tracker kernels use it --
// cycles per DEPENDENT v_add_f64 (the sequential column sums of PointCluster), per independent v_fma_f64 with
// 1 / 2 / 4 waves per SIMD, and per LDS-fed dependent add (ds_read_b64 / ds_read2_b64 rows 16 in flight).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_dep(const double *in, double *out, unsigned long long *ticks, int n)
{
    double a = in[threadIdx.x & 63], x = in[64];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int u = 0; u < 16; u++) a = a + x;
        asm volatile("" : "+v"(a));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
__global__ void k_ind(const double *in, double *out, unsigned long long *ticks, int n)
{
    double a[8], x = in[64], y = in[65];
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = in[(threadIdx.x + u) & 63];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = __builtin_fma(a[u], x, y);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) s += a[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
__global__ void k_lds(const double *in, double *out, unsigned long long *ticks, int n)
{
    __shared__ double col[6 * 1030];
    for (int i = threadIdx.x; i < 6 * 1030; i += blockDim.x) col[i] = in[i & 63];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double sum = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (lane < 12) {
        const double *c = col + (lane % 6) * 1030 + (lane / 6);
        double v[16], w[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = c[u];
        int r;
        for (r = 16; r + 16 <= n; r += 16) {
#pragma unroll
            for (int u = 0; u < 16; u++) w[u] = c[r + u];
#pragma unroll
            for (int u = 0; u < 16; u++) sum += v[u];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = w[u];
        }
#pragma unroll
        for (int u = 0; u < 16; u++) sum += v[u];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main()
{
    double h[128];
    for (int i = 0; i < 128; i++) h[i] = 1.0 + 1e-3 * i;
    double *din, *dout; unsigned long long *dt;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 8192 * 256 * 8); hipMalloc(&dt, 8192 * 8);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    const int n = 4096;
    struct { int grid, block; const char *what; } cfgs[] = {{256, 64, "1 wave per CU"}, {256, 256, "1 wave per SIMD"}, {512, 256, "2 waves per SIMD"}, {1024, 256, "4 waves per SIMD"}, {2048, 256, "8 waves per SIMD"}};
    for (auto &c : cfgs) {
        unsigned long long t;
        hipLaunchKernelGGL(k_dep, dim3(c.grid), dim3(c.block), 0, 0, din, dout, dt, n); hipDeviceSynchronize();
        hipLaunchKernelGGL(k_dep, dim3(c.grid), dim3(c.block), 0, 0, din, dout, dt, n); hipDeviceSynchronize();
        hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
        printf("%-18s dependent v_add_f64: %6.2f cycles each", c.what, (double)t / n);
        hipLaunchKernelGGL(k_ind, dim3(c.grid), dim3(c.block), 0, 0, din, dout, dt, n); hipDeviceSynchronize();
        hipLaunchKernelGGL(k_ind, dim3(c.grid), dim3(c.block), 0, 0, din, dout, dt, n); hipDeviceSynchronize();
        hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
        printf(" | 8 independent v_fma_f64 chains: %6.2f cycles per instruction", (double)t / n);
        {   // the same loop 64 x longer between two events: wave instructions per SIMD and nanosecond (wall clock, whatever the shader clock does)
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int nl = n * 64;
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_ind, dim3(c.grid), dim3(c.block), 0, 0, din, dout, dt, nl);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
            const double waves_per_simd = (double)c.grid * (c.block / 64) / 1024.0;
            printf(" (long run: %6.2f cycles, %.3f ns per instruction and wave, %.3f ns per instruction and SIMD = %.1f TFLOP/s fp64 chip-wide)",
                   (double)t / nl, ms * 1e6 / nl, ms * 1e6 / nl / (waves_per_simd < 1 ? 1 : waves_per_simd),
                   2.0 * 64 * nl * (double)c.grid * (c.block / 64) / (ms * 1e-3) * 1e-12);
        }
        hipLaunchKernelGGL(k_lds, dim3(c.grid), dim3(c.block), 0, 0, din, dout, dt, 1024); hipDeviceSynchronize();
        hipLaunchKernelGGL(k_lds, dim3(c.grid), dim3(c.block), 0, 0, din, dout, dt, 1024); hipDeviceSynchronize();
        hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
        printf(" | LDS-fed sequential sum (12 lanes): %6.2f cycles per row\n", (double)t / 1024);
    }
    return 0;
}
