// mmw_cloud.hpp -- the DBSCAN candidate cloud (the scene's global BatchedData ring, Tracking.py:51,689-697)
// as seen by k_track and k_dbscan, and the exact "apply_DBscan cannot find a core point" screen.
#pragma once

#include "mmw_device.hpp"
#include "mmw_math.hpp"

namespace mmw {

__device__ inline unsigned long long sortable(double v)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
__device__ inline double unsortable(unsigned long long k)
{
    unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
    return __longlong_as_double((long long)u);
}

// lane ^ J of a 64-bit value: DPP inside quads (J = 1, 2) and inside rows of 16 (J = 4, 8: the two row shifts, picked by the
// lane's bit J), ds_bpermute for 16 and 32 -- four of the six butterfly steps of a wave reduction without the LDS crossbar
// (a strided BallTree level reduces sixteen columns per 64 points: with six ds_bpermute steps each that pass was a third of
// a large cloud's build)
template <int J>
__device__ __forceinline__ double xor_lane_d(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    int lo = (int)(unsigned)u, hi = (int)(unsigned)(u >> 32);
    if constexpr (J == 1) { lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); }
    else if constexpr (J == 2) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); }
    else if constexpr (J == 4 || J == 8) {
        const bool up = (__lane_id() & J) != 0;
        const int lu = __builtin_amdgcn_update_dpp(0, lo, 0x100 + J, 0xF, 0xF, true), ld = __builtin_amdgcn_update_dpp(0, lo, 0x110 + J, 0xF, 0xF, true);
        const int hu = __builtin_amdgcn_update_dpp(0, hi, 0x100 + J, 0xF, 0xF, true), hd = __builtin_amdgcn_update_dpp(0, hi, 0x110 + J, 0xF, 0xF, true);
        lo = up ? ld : lu;
        hi = up ? hd : hu;
    } else { lo = __shfl_xor(lo, J); hi = __shfl_xor(hi, J); }
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ inline double wave_min_d(double v)
{
    double t;
    t = xor_lane_d<1>(v); v = t < v ? t : v;
    t = xor_lane_d<2>(v); v = t < v ? t : v;
    t = xor_lane_d<4>(v); v = t < v ? t : v;
    t = xor_lane_d<8>(v); v = t < v ? t : v;
    t = xor_lane_d<16>(v); v = t < v ? t : v;
    t = xor_lane_d<32>(v); v = t < v ? t : v;
    return v;
}
__device__ inline double wave_max_d(double v)
{
    double t;
    t = xor_lane_d<1>(v); v = t > v ? t : v;
    t = xor_lane_d<2>(v); v = t > v ? t : v;
    t = xor_lane_d<4>(v); v = t > v ? t : v;
    t = xor_lane_d<8>(v); v = t > v ? t : v;
    t = xor_lane_d<16>(v); v = t > v ? t : v;
    t = xor_lane_d<32>(v); v = t > v ? t : v;
    return v;
}

// Row source: the candidate cloud is either the concatenation (oldest first) of the
// scene's global ring frames (Tracking.py:51) or a caller-provided [n][8] block.
// Pure arithmetic on scalars, passed by value: a struct of four base pointers picked by
// comparisons gets turned into an indexed load from a private (scratch) copy by the compiler.
struct RowSrc {
    const double *gb;     // base of the scene's ring storage (or of the caller's block)
    size_t stride;        // doubles per physical frame slot
    unsigned slots;       // physical slot of frame k in byte k
    int c1, c2, c3;       // first point index of frames 1..3 (INT_MAX when absent)
    __device__ __forceinline__ const double *row(int i) const
    {
        const int k = (i >= c1 ? 1 : 0) + (i >= c2 ? 1 : 0) + (i >= c3 ? 1 : 0);
        int c = 0;
        c = i >= c1 ? c1 : c;
        c = i >= c2 ? c2 : c;
        c = i >= c3 ? c3 : c;
        const unsigned slot = (slots >> (8 * k)) & 255u;
        return gb + (size_t)slot * stride + (size_t)(i - c) * 8;
    }
};

__device__ __forceinline__ RowSrc ring_rows_of(const DevCfg &cfg, const DevState &st, const SceneHdr *hdr, int s)
{
    RowSrc src;
    const int nfr = hdr->g_len, NP = cfg.max_pts;
    const int big = 0x7fffffff;
    src.gb = st.g_ring + (size_t)s * cfg.ring * (size_t)NP * 8;
    src.stride = (size_t)NP * 8;
    src.slots = (unsigned)hdr->g_slot[0] | ((unsigned)hdr->g_slot[1] << 8) | ((unsigned)hdr->g_slot[2] << 16) | ((unsigned)hdr->g_slot[3] << 24);
    const int n0 = hdr->g_n[0], n1 = hdr->g_n[1], n2 = hdr->g_n[2];
    src.c1 = nfr > 1 ? n0 : big;
    src.c2 = nfr > 2 ? n0 + n1 : big;
    src.c3 = nfr > 3 ? n0 + n1 + n2 : big;
    return src;
}

// sklearn's input validation in front of DBSCAN.fit (apply_DBscan, Utils.py:272-278 -> check_array): every value of the
// cloud's 8 columns must be finite.  Bit 0: some value is NaN, bit 1: some value is infinite; uniform over the workgroup
// (all its threads call: two barriers).  The global ring keeps these bits per frame (SceneHdr.skipped) and never scans; this
// is for clouds that come from elsewhere (a track's ring in k_inner, a caller's block in mmw_dbscan).
__device__ __forceinline__ int cloud_nonfinite_bits(const RowSrc &src, int U)
{
    int nfb = 0;
    for (int i = threadIdx.x; i < U; i += blockDim.x) {
        const double2 *r2 = reinterpret_cast<const double2 *>(src.row(i));
        const double2 rr[4] = {r2[0], r2[1], r2[2], r2[3]};
        nfb |= row_nonfinite_bits(rr);
    }
    const int any_nan = __syncthreads_or(nfb & 1), any_inf = __syncthreads_or(nfb & 2);
    return (any_nan ? 1 : 0) | (any_inf ? 2 : 0);
}

// apply_DBscan found (or can find) nothing: labels -1, bookkeeping as after a full run with 0 clusters
// (all threads of the workgroup call)
__device__ __forceinline__ void cloud_finish_empty(const DevState &st, SceneHdr *hdr, int s, int U, int UM_out,
                                                   int32_t *__restrict__ labels_out, int32_t *__restrict__ db_n_out)
{
    const int tid = threadIdx.x;
    if (labels_out)
        for (int i = tid; i < U; i += blockDim.x) labels_out[(size_t)s * UM_out + i] = -1;
    if (tid == 0) {
        if (db_n_out) db_n_out[s] = U;
        hdr->need_db = 0;
        if (st.stats) {  // algorithmic bytes: ring rows in, labels out
            unsigned long long *sl = stats_slot(st, s);
            atomicAdd(&sl[1], (unsigned long long)(64 * U + 4 * U));
            atomicAdd(&sl[3], 1ULL);
            atomicAdd(&sl[4], (unsigned long long)U);
#ifdef MMW_STAMPS
            atomicAdd(&sl[31], 1ULL);
#endif
        }
    }
}

// ---- The "no point can be a core point" screen of apply_DBscan, for clouds of U <= 256 points ------------------
// Both stages return true (uniformly over the 256 calling threads) only when NO point can be a core point of
// sklearn's BallTree DBSCAN, i.e. every label is -1 (Utils.py:268-275 then returns no clusters).
//
// Bound (proof in dbscan_core, k_dbscan.hip): every BallTree neighbour q of p has
//   E(p,q)^2 = dx^2 + dy^2 + z_w dz^2 <= 2 eps / wmin =: R^2,   wmin = min over the cloud of 1 - range_w * y.
// Counting the points inside that ellipsoid is a SUPERSET count, so it may be formed in fp32 as long as the
// radius is widened by everything fp32 can lose: coordinates (z pre-scaled by sqrt(z_w)) are rounded once,
// |error| <= M 2^-24 per coordinate with M the largest magnitude in the cloud, i.e. at most 2 sqrt(3) M 2^-24
// on E (a norm), and the fp32 evaluation of the squared norm is good to a few 2^-24 relative.  R below carries
// 4 M 6e-8 absolute and 1e-5 relative slack, far more than both.
constexpr int kCloudGrid = 1024;

// Stage 1, O(U): the cell count, as k_track runs it on registers at the end of its frame (false = undecided).
// The points are binned into square xy cells of side h = R/2 (a hair more); everything within R of a point
// lies in the 5x5 cells around its own, so if no 5x5 block holds min_samples points nothing can be a core
// point.  Two more relaxations, both on the safe side of a superset count: wmin is bounded by
// 1 - |range_w| * M instead of the exact y range (one reduction instead of three), and the cells live on a
// 32 x 32 torus (cell = floor(coordinate / h) mod 32), so no bounding box is needed and far apart points can
// only ADD to each other's blocks.  (px, py, pz) = point `tid` of the cloud (threads tid >= U pass anything).
// `grid` [1024], mm[0] and *flag must be zero on entry (the caller does that before an earlier barrier);
// three barriers inside.
__device__ __forceinline__ bool cloud_cells_prove_no_core(const DevCfg &cfg, int U, double px, double py, double pz,
                                                          unsigned long long *mm, int *flag, int *grid)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const double rw = cfg.db_range_weight, zw = cfg.db_z_weight, eps = cfg.db_eps;
    const int min_samples = cfg.db_min_samples;
    if (!(min_samples > 1 && zw >= 0.0 && eps >= 0.0)) return false;
    double mag = 0.0;
    if (tid < U) {
        mag = fmax(fmax(fabs(px), fabs(py)), fabs(pz * sqrt(zw)));
        if (!(mag <= 1e15)) mag = __longlong_as_double(0x7ff0000000000000LL);  // NaN/inf/huge: give up below
    }
    mag = wave_max_d(mag);
    if (lane == 0) atomicMax(&mm[0], (unsigned long long)__double_as_longlong(mag));  // mag >= 0: bit order == value order
    lds_barrier();
    const double M = __longlong_as_double((long long)mm[0]);
    const double wmin = 1.0 - fabs(rw) * M;
    if (!(wmin > 0.0) || !(M <= 1e15)) return false;  // uniform
    const double R = sqrt(2.0 * (eps / wmin)) * (1.0 + 1e-9) + 4.0 * M * 6e-8;
    // |dx| <= R  =>  |dx| / h <= 2 - 2e-3, so the fp32 cell coordinates of two such points differ by less than
    // 2 - 1e-3 even after rounding (error ~1e-3 cells at most, enforced by M / h < 1e4) and the cells by at most 2
    const double h = 0.5 * R * (1.0 + 1e-3);
    if (!(h > 0.0) || !(M / h < 1e4)) return false;
    const float invh = (float)(1.0 / h);
    int cx = 0, cy = 0;
    if (tid < U) {
        cx = (int)floorf((float)px * invh) & 31;
        cy = (int)floorf((float)py * invh) & 31;
        atomicAdd(&grid[cy * 32 + cx], 1);
    }
    lds_barrier();
    bool maybe = false;
    if (tid < U) {
        int c = 0;
#pragma unroll
        for (int dy = -2; dy <= 2; dy++)
#pragma unroll
            for (int dx = -2; dx <= 2; dx++) c += grid[((cy + dy) & 31) * 32 + ((cx + dx) & 31)];
        maybe = c >= min_samples;
    }
    if (maybe) *flag = 1;
    lds_barrier();
    return *flag == 0;
}

// Stage 1 for clouds of more than 256 points (U <= 2048; NT = threads of the workgroup), rows read from the scene's global
// ring: a scene without tracks keeps three full frames of clutter in the ring, and without this its apply_DBscan
// would pay the O(U^2) pair count of dbscan_core in every frame.  Same contract as above.
template <int NT = 256>
__device__ __forceinline__ bool cloud_cells_prove_no_core_rows(const DevCfg &cfg, const RowSrc src, int U, unsigned long long *mm,
                                                               int *flag, int *grid)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const double rw = cfg.db_range_weight, zw = cfg.db_z_weight, eps = cfg.db_eps;
    const int min_samples = cfg.db_min_samples;
    if (!(min_samples > 1 && zw >= 0.0 && eps >= 0.0) || U > 8 * 256) return false;
    const double sqzw = sqrt(zw);
    double mag = 0.0;
    for (int p = tid; p < U; p += NT) {
        const double *r = src.row(p);
        const double2 a = *reinterpret_cast<const double2 *>(r);
        double m = fmax(fmax(fabs(a.x), fabs(a.y)), fabs(r[2] * sqzw));
        if (!(m <= 1e15)) m = __longlong_as_double(0x7ff0000000000000LL);
        mag = fmax(mag, m);
    }
    mag = wave_max_d(mag);
    if (lane == 0) atomicMax(&mm[0], (unsigned long long)__double_as_longlong(mag));
    lds_barrier();
    const double M = __longlong_as_double((long long)mm[0]);
    const double wmin = 1.0 - fabs(rw) * M;
    if (!(wmin > 0.0) || !(M <= 1e15)) return false;  // uniform
    const double R = sqrt(2.0 * (eps / wmin)) * (1.0 + 1e-9) + 4.0 * M * 6e-8;
    const double h = 0.5 * R * (1.0 + 1e-3);
    if (!(h > 0.0) || !(M / h < 1e4)) return false;
    const float invh = (float)(1.0 / h);
    for (int p = tid; p < U; p += NT) {
        const double2 a = *reinterpret_cast<const double2 *>(src.row(p));
        const int cx = (int)floorf((float)a.x * invh) & 31, cy = (int)floorf((float)a.y * invh) & 31;
        atomicAdd(&grid[cy * 32 + cx], 1);
    }
    lds_barrier();
    bool maybe = false;
    for (int p = tid; p < U; p += NT) {  // (rows again: they are in L2, and eight cell indices per thread would spill)
        const double2 a = *reinterpret_cast<const double2 *>(src.row(p));
        const int cx = (int)floorf((float)a.x * invh) & 31, cy = (int)floorf((float)a.y * invh) & 31;
        int c = 0;
#pragma unroll
        for (int dy = -2; dy <= 2; dy++)
#pragma unroll
            for (int dx = -2; dx <= 2; dx++) c += grid[((cy + dy) & 31) * 32 + ((cx + dx) & 31)];
        maybe = maybe || c >= min_samples;
    }
    if (maybe) *flag = 1;
    lds_barrier();
    return *flag == 0;
}

// Stage 2, O(U^2): the exact superset count over all pairs, for the clouds stage 1 left undecided (k_post: rows read from
// the scene's global ring; k_scene: from the registers that fed stage 1).  P4 [256] float4, cnt [256] ints, mm [3] u64, flag [1] are LDS scratch.  NT = threads of
// the workgroup (256 in k_post, 512 in k_chain); the count -- and with it the answer -- does not depend on it.
// (px, py, pz) = point `tid` of the cloud (threads tid >= U pass anything).
template <int NT = 256>
__device__ __forceinline__ bool cloud_pairs_prove_no_core_xyz(const DevCfg &cfg, int U, double px, double py, double pz, float4 *P4, int *cnt,
                                                              unsigned long long *mm, int *flag)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const double rw = cfg.db_range_weight, zw = cfg.db_z_weight, eps = cfg.db_eps;
    const int min_samples = cfg.db_min_samples;
    if (!(min_samples > 1 && zw >= 0.0 && eps >= 0.0)) return false;
    const double sqzw = sqrt(zw);
    double y = 0.0, mag = 0.0;
    if (tid < U) {
        const double zs = pz * sqzw;
        y = py;
        P4[tid] = make_float4((float)px, (float)py, (float)zs, 0.f);
        mag = fmax(fmax(fabs(px), fabs(py)), fabs(zs));
        if (!(mag <= 1e15)) mag = __longlong_as_double(0x7ff0000000000000LL);  // NaN/inf/huge: give up below
        cnt[tid] = 0;
    }
    if (tid == 0) { mm[0] = ~0ULL; mm[1] = 0ULL; mm[2] = 0ULL; *flag = 0; }
    lds_barrier();
    double ylo = tid < U ? y : 1.7976931348623157e308, yhi = tid < U ? y : -1.7976931348623157e308;
    ylo = wave_min_d(ylo);
    yhi = wave_max_d(yhi);
    mag = wave_max_d(mag);
    if (lane == 0 && ylo <= yhi) {
        atomicMin(&mm[0], sortable(ylo));
        atomicMax(&mm[1], sortable(yhi));
        atomicMax(&mm[2], (unsigned long long)__double_as_longlong(mag));  // mag >= 0: bit order == value order
    }
    lds_barrier();
    const double ymin = unsortable(mm[0]), ymax = unsortable(mm[1]), M = __longlong_as_double((long long)mm[2]);
    const double wa = 1 - ymax * rw, wb = 1 - ymin * rw;
    const double wmin = wa < wb ? wa : wb;
    if (!(wmin > 0.0) || !(M <= 1e15)) return false;  // uniform: same LDS values for every thread
    const double R = sqrt(2.0 * (eps / wmin)) * (1.0 + 1e-9) + 4.0 * M * 6e-8;
    const float R2f = (float)(R * R * (1.0 + 1e-5));
    // tasks (point p, slice `part` of the partners), about four per thread
    int parts = 1024 / U;
    parts = parts < 1 ? 1 : (parts > 16 ? 16 : parts);
    const int q256 = NT / U, r256 = NT - q256 * U, ntask = U * parts;
    int part = tid / U, p = tid - part * U;
    for (int t = tid; t < ntask; t += NT) {
        const float4 a = P4[p];
        int c = 0;
#pragma unroll 4
        for (int q = part; q < U; q += parts) {
            const float4 b = P4[q];
            const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
            const float e = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            c += (e <= R2f) ? 1 : 0;
        }
        atomicAdd(&cnt[p], c);
        p += r256;
        part += q256;
        if (p >= U) { p -= U; part++; }
    }
    lds_barrier();
    if (tid < U && cnt[tid] >= min_samples) *flag = 1;
    lds_barrier();
    return *flag == 0;
}
// ... with the rows read from the scene's global ring
template <int NT = 256>
__device__ __forceinline__ bool cloud_pairs_prove_no_core(const DevCfg &cfg, const RowSrc src, int U, float4 *P4, int *cnt,
                                                          unsigned long long *mm, int *flag)
{
    double px = 0.0, py = 0.0, pz = 0.0;
    if ((int)threadIdx.x < U) {
        const double *r = src.row(threadIdx.x);
        const double2 a = *reinterpret_cast<const double2 *>(r);
        px = a.x; py = a.y; pz = r[2];
    }
    return cloud_pairs_prove_no_core_xyz<NT>(cfg, U, px, py, pz, P4, cnt, mm, flag);
}

}  // namespace mmw
