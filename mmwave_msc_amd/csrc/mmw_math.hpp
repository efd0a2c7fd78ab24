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

// Workgroup barrier for phases that hand data over through LDS only.  __syncthreads() is also a fence for
// global memory, i.e. every wave first waits until its outstanding global stores have completed (a round
// trip of a thousand cycles and more); rows, track records and labels written for LATER kernels need no
// such wait.  Use only where no thread reads global memory another thread of the workgroup wrote.
__device__ __forceinline__ void lds_barrier()
{
    // release / acquire at workgroup scope for the LDS address space only: the compiler emits
    // s_waitcnt lgkmcnt(0) ; s_barrier and leaves outstanding global stores (vmcnt) alone
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// 6x6 partial-pivot LU, determinant and inverse (stands for np.linalg.det / np.linalg.inv at
// Tracking.py:558-560 and filterpy's inv(S)), FOUR matrices per wave, column-parallel Gauss-Jordan
// on the augmented matrix [A | I]: in every 16-lane group lane c < 6 holds column c of A in v[0..5],
// lanes 6..11 hold the columns of the identity (set here), lanes 12..15 idle.
//
// Why columns: the pivot search of step k and the multipliers l_i = A[i][k] / A[k][k] only involve
// column k, i.e. ONE lane's registers.  Every lane runs the step on its own column; what lane k found
// (pivot row, 1/pivot, the multipliers) is broadcast in one batch of cross-lane reads, then every lane
// swaps and eliminates locally: one cross-lane round trip per pivot step, and one more for the whole
// back substitution -- the wave spends its time on 6x6 arithmetic, not on the LDS crossbar.
//
// The arithmetic per element is that of the CPU oracle's lu6(): first-max pivoting; 1/pivot formed
// once (rp); l = a*rp; a[i][c] -= l_i*a[k][c]; the right-hand sides see the same row operations
// (forward substitution, k ascending); back substitution subtracts U[r][k]*x[k] for k DESCENDING and
// scales by rp[r] last; det = signed product of the pivots in order.
// On return lanes 6..11 hold columns 0..5 of the inverse in v; det is valid in every lane of the group.
// All 64 lanes must call; the result (false = a zero pivot) is per group.
// Lane k (0..5, a constant once the loops below are unrolled) of every 16-lane row to all lanes of that row, as one DPP move
// (row_newbcast: v_mov_b64_dpp for a double, v_mov_b32_dpp for an int) -- not a trip through the LDS crossbar (ds_bpermute), which
// is what __shfl(x, row base + k) compiles to.  All 64 lanes must be active.
template <typename T>
__device__ __forceinline__ T lu_row_bcast(T x, int k)
{
    switch (k) {
    case 0: return __builtin_amdgcn_mov_dpp(x, 0x150, 0xf, 0xf, false);
    case 1: return __builtin_amdgcn_mov_dpp(x, 0x151, 0xf, 0xf, false);
    case 2: return __builtin_amdgcn_mov_dpp(x, 0x152, 0xf, 0xf, false);
    case 3: return __builtin_amdgcn_mov_dpp(x, 0x153, 0xf, 0xf, false);
    case 4: return __builtin_amdgcn_mov_dpp(x, 0x154, 0xf, 0xf, false);
    default: return __builtin_amdgcn_mov_dpp(x, 0x155, 0xf, 0xf, false);
    }
}
__device__ __forceinline__ bool lu6_inverse_cols(double (&v)[6], int lane, double &det)
{
    const int c = lane & 15;
    if (c >= 6) {
#pragma unroll
        for (int i = 0; i < 6; i++) v[i] = (c - 6 == i) ? 1.0 : 0.0;
    }
    bool neg = false, ok = true;
    double rp[6], d = 1.0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        // this lane's column as if it were column k: pivot row, 1/pivot, swapped column, multipliers
        int p = k;
        double pv = v[k];
#pragma unroll
        for (int r2 = k + 1; r2 < 6; r2++) {
            const bool gt = fabs(v[r2]) > fabs(pv);  // first maximum of |column|
            p = gt ? r2 : p;
            pv = gt ? v[r2] : pv;
        }
        const double rpl = 1.0 / pv;
        double l[6];
#pragma unroll
        for (int i = k + 1; i < 6; i++) l[i] = ((p == i) ? v[k] : v[i]) * rpl;  // row i after the swap k <-> p
        // what lane k found, for the whole group
        p = lu_row_bcast(p, k);
        pv = lu_row_bcast(pv, k);
        rp[k] = lu_row_bcast(rpl, k);
#pragma unroll
        for (int i = k + 1; i < 6; i++) l[i] = lu_row_bcast(l[i], k);
        if (!(fabs(pv) > 0.0)) ok = false;
        if (p != k) neg = !neg;
        d = (k == 0) ? pv : d * pv;  // the pivots in order
        {  // rows k <-> p of this lane's column
            const double tk = v[k];
#pragma unroll
            for (int r2 = k + 1; r2 < 6; r2++)
                if (p == r2) { v[k] = v[r2]; v[r2] = tk; }
        }
        // every lane eliminates: in columns <= k this only overwrites entries below the diagonal (the L
        // factors), which nothing reads again -- the right-hand sides see the row operations as they happen
#pragma unroll
        for (int i = k + 1; i < 6; i++) v[i] = v[i] - l[i] * v[k];
    }
    det = neg ? -d : d;
    // U above the diagonal, all at once (one cross-lane round trip); after that the lanes that held U may
    // overwrite it -- only lanes 6..11 carry a result
    double u[6][6];
#pragma unroll
    for (int k = 1; k < 6; k++) {
#pragma unroll
        for (int r = 0; r < k; r++) u[r][k] = lu_row_bcast(v[r], k);
    }
#pragma unroll
    for (int k = 5; k >= 0; k--) {
        v[k] = v[k] * rp[k];
#pragma unroll
        for (int r = 0; r < k; r++) v[r] = v[r] - u[r][k] * v[k];
    }
    return ok;
}

}  // namespace mmw
