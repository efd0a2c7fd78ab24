// Does vector work hide in the shadow of a wave's own MFMAs?  One wave per SIMD (256-thread workgroups, one per CU), a loop of six
// v_mfma_f32_32x32x16_f16 and 42 independent v_fma_f32 per iteration, (A) the MFMAs first and the vector work behind them -- what the
// compiler makes of an epilogue chunk placed behind a tap --, (B) dealt out, seven behind each MFMA, (C) MFMAs only, (D) vector work only.
//   hipcc --offload-arch=gfx950 -O3 ubench_mfma_valu.hip -o ubench_mfma_valu && ./ubench_mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define MF(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)
#define V7(x0, x1, x2, x3, x4, x5, x6) \
    asm volatile("v_fma_f32 %0, %0, %7, %8\n v_fma_f32 %1, %1, %7, %8\n v_fma_f32 %2, %2, %7, %8\n v_fma_f32 %3, %3, %7, %8\n v_fma_f32 %4, %4, %7, %8\n v_fma_f32 %5, %5, %7, %8\n v_fma_f32 %6, %6, %7, %8" \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6) : "v"(m), "v"(c))
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters)
{
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0}, c4 = {0}, c5 = {0};
    float x[7];
    for (int i = 0; i < 7; i++) x[i] = threadIdx.x + i;
    const float m = 0.999f, c = 0.001f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {   // A: six MFMAs, then 42 VALU
            MF(c0); MF(c1); MF(c2); MF(c3); MF(c4); MF(c5);
            __builtin_amdgcn_sched_barrier(0);
            for (int r = 0; r < 6; r++) { V7(x[0], x[1], x[2], x[3], x[4], x[5], x[6]); }
        } else if (MODE == 1) {   // B: one MFMA, seven VALU
            MF(c0); __builtin_amdgcn_sched_barrier(0); V7(x[0], x[1], x[2], x[3], x[4], x[5], x[6]); __builtin_amdgcn_sched_barrier(0);
            MF(c1); __builtin_amdgcn_sched_barrier(0); V7(x[0], x[1], x[2], x[3], x[4], x[5], x[6]); __builtin_amdgcn_sched_barrier(0);
            MF(c2); __builtin_amdgcn_sched_barrier(0); V7(x[0], x[1], x[2], x[3], x[4], x[5], x[6]); __builtin_amdgcn_sched_barrier(0);
            MF(c3); __builtin_amdgcn_sched_barrier(0); V7(x[0], x[1], x[2], x[3], x[4], x[5], x[6]); __builtin_amdgcn_sched_barrier(0);
            MF(c4); __builtin_amdgcn_sched_barrier(0); V7(x[0], x[1], x[2], x[3], x[4], x[5], x[6]); __builtin_amdgcn_sched_barrier(0);
            MF(c5); __builtin_amdgcn_sched_barrier(0); V7(x[0], x[1], x[2], x[3], x[4], x[5], x[6]); __builtin_amdgcn_sched_barrier(0);
        } else if (MODE == 2) {
            MF(c0); MF(c1); MF(c2); MF(c3); MF(c4); MF(c5);
        } else {
            for (int r = 0; r < 6; r++) { V7(x[0], x[1], x[2], x[3], x[4], x[5], x[6]); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i] + c4[i] + c5[i];
    for (int i = 0; i < 7; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
float run(float *d, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    float *d; hipMalloc(&d, 256 * 256 * 4);
    const int iters = 20000;
    const char *names[4] = {"A: 6 MFMA then 42 VALU", "B: (1 MFMA, 7 VALU) x 6", "C: 6 MFMA", "D: 42 VALU"};
    float t[4] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters)};
    for (int i = 0; i < 4; i++) printf("%-28s %8.3f ms  = %6.1f ns per iteration\n", names[i], t[i], t[i] * 1e6 / iters);
    return 0;
}
