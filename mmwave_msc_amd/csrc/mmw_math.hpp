// mmw_math.hpp -- small fp64 device helpers with a FIXED operation order.
// Build with -ffp-contract=off: every a*b+c below is two roundings, on purpose
// (bit-reproducibility against the CPU restatement used by the parity tests).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmw {

// log() with the fdlibm/musl reduction + polynomial (Tracking.py:558 uses np.log;
// device libm and host libm differ in the last bit, so the kernels carry their own).
__device__ inline double dlog(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (x != x) return x;
    if (x < 0.0) return __longlong_as_double(0x7ff8000000000000LL);
    if (x == 0.0) return __longlong_as_double(0xfff0000000000000LL);
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    if (u == 0x7ff0000000000000ULL) return x;
    int k = 0;
    if ((u >> 52) == 0) {
        x *= 18014398509481984.0;
        k -= 54;
        u = (unsigned long long)__double_as_longlong(x);
    }
    unsigned int hx = (unsigned int)(u >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((unsigned long long)hx << 32) | (u & 0xffffffffULL);
    x = __longlong_as_double((long long)u);
    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

// altered_EuclideanDist (Utils.py:242-247), operation order kept.
__device__ inline double alt_dist(double ax, double ay, double az, double bx, double by, double bz,
                                  double range_w, double z_w)
{
    double w = 1 - ((ay + by) / 2) * range_w;
    double dx = ax - bx, dy = ay - by, dz = az - bz;
    return w * ((dx * dx + dy * dy) + z_w * (dz * dz));
}

__device__ inline unsigned long long lanemask_lt()
{
    unsigned lane = __lane_id();
    return lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
}

// Ordering point between phases that exchange data through LDS inside ONE wave: DS operations
// of a wave execute in issue order, so only the compiler has to be kept from moving accesses.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 6x6 partial-pivot LU, determinant and inverse (stands for np.linalg.det / np.linalg.inv at
// Tracking.py:558-560 and filterpy's inv(S)), up to EIGHT matrices per wave: lanes 8g..8g+5 hold
// rows 0..5 of matrix g (lanes 8g+6, 8g+7 idle).  First-max partial pivoting, L stored in place, det = signed product of pivots; the
// inverse is solved for all six right-hand sides at once in axpy form: forward substitution
// subtracts L[r][k]*Y[k] for k ascending, back substitution subtracts U[r][k]*X[k] for k
// DESCENDING and divides by U[r][r] last (the order the CPU oracle restates).  One call costs
// about what the one-matrix-per-wave version costs, so a scene's tracks share it.
// All 64 lanes must call; `ok` is per group.
__device__ __forceinline__ bool lu6_inverse_rows(double (&a)[6], int lane, double (&inv)[6], double &det)
{
    const int r = lane & 7, gb = lane & ~7;
    int prow = r;
    bool neg = false, ok = true;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        int p = k;
        double best = fabs(__shfl(a[k], gb + k));
#pragma unroll
        for (int r2 = k + 1; r2 < 6; r2++) {
            const double v = fabs(__shfl(a[k], gb + r2));
            if (v > best) { best = v; p = r2; }
        }
        if (!(best > 0.0)) ok = false;
        const int src = gb + ((r == k) ? p : ((r == p) ? k : r));
#pragma unroll
        for (int c = 0; c < 6; c++) a[c] = __shfl(a[c], src);
        prow = __shfl(prow, src);
        if (p != k) neg = !neg;
        double piv[6];
#pragma unroll
        for (int c = k; c < 6; c++) piv[c] = __shfl(a[c], gb + k);
        if (r > k && r < 6) {
            const double l = a[k] / piv[k];
            a[k] = l;
#pragma unroll
            for (int c = k + 1; c < 6; c++) a[c] = a[c] - l * piv[c];
        }
    }
    double d = __shfl(a[0], gb);
#pragma unroll
    for (int k = 1; k < 6; k++) d = d * __shfl(a[k], gb + k);
    det = neg ? -d : d;
    double y[6];
#pragma unroll
    for (int c = 0; c < 6; c++) y[c] = (prow == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        double yk[6];
#pragma unroll
        for (int c = 0; c < 6; c++) yk[c] = __shfl(y[c], gb + k);
        if (r > k && r < 6) {
#pragma unroll
            for (int c = 0; c < 6; c++) y[c] = y[c] - a[k] * yk[c];
        }
    }
#pragma unroll
    for (int k = 5; k >= 0; k--) {
        const double ukk = __shfl(a[k], gb + k);
        if (r == k) {
#pragma unroll
            for (int c = 0; c < 6; c++) y[c] = y[c] / ukk;
        }
        if (k > 0) {
            double xk[6];
#pragma unroll
            for (int c = 0; c < 6; c++) xk[c] = __shfl(y[c], gb + k);
            if (r < k) {
#pragma unroll
                for (int c = 0; c < 6; c++) y[c] = y[c] - a[k] * xk[c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 6; c++) inv[c] = y[c];
    return ok;
}

}  // namespace mmw
