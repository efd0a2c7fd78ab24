// Micro-benchmark (diagnostic, not part of the product): the Mahalanobis gate of k_track as it is (fp64 chains, gate records as
// SGPR operands through the scalar cache) against a packed-fp32 evaluation of the same quadratic form (symmetric 21-term form,
// two points per v_pk_fma_f32, records as fp32 SGPR operands) -- what an fp32 screen in front of the fp64 gate would issue.
// 4096 workgroups x 256 threads x 2 points x T tracks, as the 4096-scene step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kRec = 44, kRec32 = 32;

template <int PPT>
__global__ __launch_bounds__(256, 5) void gate64(const double *__restrict__ pts, const double *__restrict__ rec, int T, int *out)
{
    typedef const double __attribute__((address_space(4))) *gp;
    const int s = blockIdx.x, tid = threadIdx.x;
    double2 pr[PPT][3];
    const double2 *src = reinterpret_cast<const double2 *>(pts + (size_t)s * 512 * 8);
#pragma unroll
    for (int q = 0; q < PPT; q++)
#pragma unroll
        for (int u = 0; u < 3; u++) pr[q][u] = src[(q * 256 + tid) * 4 + u];
    gp gb = (gp)(rec + (size_t)__builtin_amdgcn_readfirstlane(s) * 8 * kRec);
    double bestd[PPT];
    int bestj[PPT];
#pragma unroll
    for (int q = 0; q < PPT; q++) { bestd[q] = 0; bestj[q] = -1; }
    for (int j = 0; j < T; j++) {
        gp G = gb + j * kRec;
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const double y0 = pr[q][0].x - G[37], y1 = pr[q][0].y - G[38], y2 = pr[q][1].x - G[39], y3 = pr[q][1].y - G[40], y4 = pr[q][2].x - G[41],
                         y5 = pr[q][2].y - G[42];
            double v[6];
#pragma unroll
            for (int k = 0; k < 6; k++) v[k] = y0 * G[k];
#pragma unroll
            for (int k = 0; k < 6; k++) v[k] = __builtin_fma(y1, G[6 + k], v[k]);
#pragma unroll
            for (int k = 0; k < 6; k++) v[k] = __builtin_fma(y2, G[12 + k], v[k]);
#pragma unroll
            for (int k = 0; k < 6; k++) v[k] = __builtin_fma(y3, G[18 + k], v[k]);
#pragma unroll
            for (int k = 0; k < 6; k++) v[k] = __builtin_fma(y4, G[24 + k], v[k]);
#pragma unroll
            for (int k = 0; k < 6; k++) v[k] = __builtin_fma(y5, G[30 + k], v[k]);
            double quad = v[0] * y0;
            quad = __builtin_fma(v[1], y1, quad);
            quad = __builtin_fma(v[2], y2, quad);
            quad = __builtin_fma(v[3], y3, quad);
            quad = __builtin_fma(v[4], y4, quad);
            quad = __builtin_fma(v[5], y5, quad);
            const double d = G[36] + quad;
            if (d < 4.5 && (bestj[q] < 0 || d < bestd[q])) { bestj[q] = j; bestd[q] = d; }
        }
    }
    int acc = 0;
#pragma unroll
    for (int q = 0; q < PPT; q++) acc += bestj[q];
    out[s * 256 + tid] = acc;
}

// record32[j]: [0..20] S_ab (a <= b, off-diagonals doubled), [21..26] -hx, [27] G_lo, [28] G_hi
__global__ __launch_bounds__(256, 5) void gate32pk(const double *__restrict__ pts, const float *__restrict__ rec, int T, int *out)
{
    typedef const float __attribute__((address_space(4))) *gp;
    const int s = blockIdx.x, tid = threadIdx.x;
    f2 p[6];
    {
        const double2 *src = reinterpret_cast<const double2 *>(pts + (size_t)s * 512 * 8);
        double2 a[2][3];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int u = 0; u < 3; u++) a[q][u] = src[(q * 256 + tid) * 4 + u];
#pragma unroll
        for (int u = 0; u < 3; u++) { p[2 * u] = f2{(float)a[0][u].x, (float)a[1][u].x}; p[2 * u + 1] = f2{(float)a[0][u].y, (float)a[1][u].y}; }
    }
    gp gb = (gp)(rec + (size_t)__builtin_amdgcn_readfirstlane(s) * 8 * kRec32);
    f2 best = f2{3.0e38f, 3.0e38f}, second = best;
    int bj0 = -1, bj1 = -1, n0 = 0, n1 = 0;
    for (int j = 0; j < T; j++) {
        gp G = gb + j * kRec32;
        f2 y[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { const float h = G[21 + k]; y[k] = p[k] + f2{h, h}; }
        f2 q2 = f2{0.f, 0.f};
        int e = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const float s0 = G[e++];
            f2 t = y[a] * f2{s0, s0};
#pragma unroll
            for (int b = a + 1; b < 6; b++) { const float sv = G[e++]; t = __builtin_elementwise_fma(y[b], f2{sv, sv}, t); }
            q2 = a == 0 ? t * y[0] : __builtin_elementwise_fma(t, y[a], q2);
        }
        const float ghi = G[28];
        // bookkeeping of a screen: best and runner-up of the possible passers, their count
        const bool p0 = q2.x < ghi, p1 = q2.y < ghi;
        const float d0 = q2.x, d1 = q2.y;
        if (p0) { n0++; if (d0 < best.x) { second.x = best.x; best.x = d0; bj0 = j; } else if (d0 < second.x) second.x = d0; }
        if (p1) { n1++; if (d1 < best.y) { second.y = best.y; best.y = d1; bj1 = j; } else if (d1 < second.y) second.y = d1; }
    }
    out[s * 256 + tid] = bj0 + bj1 + n0 + n1 + (second.x < 1e30f) + (second.y < 1e30f);
}

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 8, S = 4096, reps = 30;
    std::vector<double> pts((size_t)S * 512 * 8), rec((size_t)S * 8 * kRec);
    std::vector<float> rec32((size_t)S * 8 * kRec32);
    srand(1);
    for (auto &v : pts) v = (rand() % 2000) / 250.0 - 4.0;
    for (size_t i = 0; i < rec.size(); i++) rec[i] = (i % kRec) < 36 ? (((i % kRec) % 7 == 0) ? 5.0 : 0.01) : ((i % kRec) == 36 ? -6.0 : ((rand() % 2000) / 250.0 - 4.0));
    for (size_t i = 0; i < rec32.size(); i++) rec32[i] = (i % kRec32) < 21 ? 0.5f : ((i % kRec32) < 27 ? (float)((rand() % 2000) / 250.0 - 4.0) : 10.5f);
    double *d_pts, *d_rec; float *d_rec32; int *d_out;
    hipMalloc(&d_pts, pts.size() * 8); hipMalloc(&d_rec, rec.size() * 8); hipMalloc(&d_rec32, rec32.size() * 4); hipMalloc(&d_out, (size_t)S * 256 * 4);
    hipMemcpy(d_pts, pts.data(), pts.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_rec, rec.data(), rec.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_rec32, rec32.data(), rec32.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; mode++) {
        for (int w = 0; w < 3; w++) {
            if (mode == 0) hipLaunchKernelGGL(gate64<2>, dim3(S), dim3(256), 0, 0, d_pts, d_rec, T, d_out);
            else hipLaunchKernelGGL(gate32pk, dim3(S), dim3(256), 0, 0, d_pts, d_rec32, T, d_out);
        }
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < reps; r++) {
            if (mode == 0) hipLaunchKernelGGL(gate64<2>, dim3(S), dim3(256), 0, 0, d_pts, d_rec, T, d_out);
            else hipLaunchKernelGGL(gate32pk, dim3(S), dim3(256), 0, 0, d_pts, d_rec32, T, d_out);
        }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        printf("%s T=%d: %.2f us per launch (%d scenes x 512 points)\n", mode == 0 ? "gate64 (SGPR fp64 chains)" : "gate32pk (packed fp32, symmetric form)", T, ms / reps * 1e3, S);
    }
    return 0;
}
