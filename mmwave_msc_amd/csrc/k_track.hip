// k_track.hip -- one workgroup (256 threads = 4 wave64) per scene: the association part of
// TrackBuffer.track (Tracking.py:664-703), between the two batched Kalman kernels (k_kalman.hip, k_post):
//   _calc_dist_fun gating/association (gate matrices from k_predict) -> _get_gated_clouds ->
//   associate_pointcloud estimators -> _maintain_tracks -> batch.add_frame(unassigned) on the global ring
//   -> first stage of the apply_DBscan screen; scenes that may hold a cluster go to the BallTree work lists.
//
// Data movement per scene-frame: the frame's points are read ONCE from HBM (16-byte loads of whole rows
// into registers), gated from registers, then parked class-sorted in an LDS SoA tile (6 columns) for the
// cluster statistics; rows go from registers to their ring; track records (1.5 KB each) live in HBM/L2.
// All arithmetic fp64 with a fixed operation order (see mmw_math.hpp) -- the order the parity oracle
// restates from the reference.  Phase list, LDS budget and what the probes say: DESIGN.md §5.
#include <cstdlib>

#include "mmw_device.hpp"
#include "mmw_math.hpp"
#include "mmw_cloud.hpp"
#include "mmw_kalman.hpp"
#include "mmw_launch.hpp"

namespace mmw {

// Diagnostic build only (make STAMPS=1 -> libmmw_hip_stamps.so): per-phase cycle sums of wave 0,
// accumulated into stats[8 + phase].  Never compiled into the product library.
#ifdef MMW_STAMPS
#define STAMP(k)                                                                              \
    do {                                                                                      \
        if (tid == 0) {                                                                       \
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();                   \
            atomicAdd(&stats_slot(st, blockIdx.x)[8 + (k)], t_now - t_prev);                                    \
            t_prev = t_now;                                                                   \
        }                                                                                     \
    } while (0)
// PROBE(id): raw clock of lane 0 of every wave of ONE workgroup (scene kProbeScene), for timelines
constexpr int kProbeScene = 460;   // (a block index: st.perm puts the scenes with the most tracks first)
#define PROBE(id)                                                                             \
    do {                                                                                      \
        if (blockIdx.x == kProbeScene && (threadIdx.x & 63) == 0)                             \
            st.stats[kStatSlots * kStatWords + (threadIdx.x >> 6) * 64 + (id)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
// WGTIME(k): s_memrealtime (100 MHz, chip-wide) and s_memtime of every workgroup's start (k = 0) and end (k = 1)
// (-DMMW_STAMPS_POST: k_post's workgroups write these words instead, k_dbscan.hip)
#ifdef MMW_STAMPS_POST
#define WGTIME(k)
#else
// (2048 slots: a launch of more workgroups stamps every second / fourth ... one)
#define WGTIME(k)                                                                             \
    do {                                                                                      \
        int wg_sh = 0;                                                                        \
        while (((int)gridDim.x >> wg_sh) > 2048) wg_sh++;                                     \
        if (threadIdx.x == 0 && (blockIdx.x & ((1u << wg_sh) - 1)) == 0) {                    \
            const unsigned wg_slot = blockIdx.x >> wg_sh;                                     \
            st.stats[kStatSlots * kStatWords + 256 + wg_slot * 4 + (k) * 2] = __builtin_amdgcn_s_memrealtime(); \
            st.stats[kStatSlots * kStatWords + 256 + wg_slot * 4 + (k) * 2 + 1] = __builtin_amdgcn_s_memtime(); \
        }                                                                                     \
    } while (0)
#endif
#else
#define STAMP(k)
#define PROBE(id)
#define WGTIME(k)
#endif


// Columns of the point tile are NP + 2 doubles apart: with a power-of-two stride the same row of all six
// columns -- what the lanes of one track read together -- would sit in one LDS bank (6-way conflicts).
constexpr int kTilePad = 2;
// per track: gate record in (352) + spread, N_est, group dispersion, ring state in (392) + centroid, min, max,
// spread, group dispersion, N_est, lifetime, counters, ring state out (540)
constexpr int kTrackBytesPerTrack = 352 + 392 + 540;

// numpy's pairwise split point and the bound on leaves per frame (see pw_* below): a leaf that comes from a
// split holds at least 57 rows, so a frame has at most max_pts/57 of them.
__host__ __device__ inline int pw_split(int n) { const int h = n / 2; return h - h % 8; }
__host__ __device__ inline int pw_max_leaves(int np) { return np > 128 ? np / 57 + 1 : 0; }

struct TrackLds {
    double *p6;      // [6][NP + kTilePad] point columns x,y,z,vx,vy,vz, class-sorted (see the split)
    double *work;    // union: gate[kGateChunk][kGateStride] | point tile + pairwise stack
    double *cen;     // [t_cap][6] centroid of this frame's cloud per track
    unsigned short *cnt;  // [NB][CLS] per 64-point block: class counts, then their exclusive prefix (< max_pts <= 1024)
    int *cls_n;      // [CLS]
    int *cls_off;    // [CLS+1]
    long long *seg_dst; // [CLS] where this frame's rows go: track j's ring slot (j < T), the global ring slot (j == T); in doubles
    double *wmm;     // [kWaves][24] per-wave min/max hand-over of the cluster statistics
    double *nest;    // [t_cap] N_est after this frame's estimate (stats -> dispersion phase)
    int *ml;         // multi-leaf clouds (n > 128): [0] leaves, [1] clouds, then per leaf (track, off, len), per cloud (track, first leaf)
    int *slot;       // [t_cap]
    int *slot2;      // [t_cap]
    int *misc;       // [16]
};

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// WRITE=false only sizes the layout.  (No `if (L)` null test: in the private address space a
// null check on an alloca cannot be folded and would pin the struct in scratch memory.)
template <bool WRITE>
__host__ __device__ __forceinline__ size_t track_lds_layout(const DevCfg &c, char *base, TrackLds *L)
{
    const int NP = c.max_pts, NB = (NP + 63) / 64, CLS = c.t_cap + 1;
    // One region, two lives: (1) the SoA point tile + the leaf sums of the pairwise recursion while cluster
    // statistics are formed; (2) the cell grid of the DBSCAN screen at the very end.  (The gate records of the
    // tracks, once staged here, are read through the scalar cache now.)
    const int work_a = 0, work_b = 6 * (NP + kTilePad) + pw_max_leaves(NP) * 21;
    size_t off = 0;
#define CARVE(field, type, count)                            \
    if constexpr (WRITE) L->field = (type *)(base + off);    \
    off = align16(off + sizeof(type) * (size_t)(count));
    if constexpr (WRITE) L->p6 = (double *)(base + off);
    const int work_c = (12288 + 256 * 4 + 64) / 8;  // (3): [P4 of the pair count 4096 | grid 4096 + mm | ... | pair counts at 12288 + mm2]
    const int work_ab = work_a > work_b ? work_a : work_b;
    CARVE(work, double, work_ab > work_c ? work_ab : work_c)
    CARVE(cen, double, c.t_cap * 6)
    {   // two lives of one region: [cnt | seg_dst] until the points are parked, [wmm | nest | slot2] afterwards
        const size_t base_off = off;
        CARVE(cnt, unsigned short, NB *CLS)
        CARVE(seg_dst, long long, CLS + 1)
        const size_t end_a = off;
        off = base_off;
        CARVE(wmm, double, kWaves * 24)
        CARVE(nest, double, c.t_cap)
        CARVE(slot2, int, c.t_cap)
        off = off > end_a ? off : end_a;
    }
    CARVE(cls_n, int, CLS)
    CARVE(cls_off, int, CLS + 1)
    CARVE(ml, int, 2 + 5 * (pw_max_leaves(NP) + 1))
    CARVE(slot, int, c.t_cap)
    CARVE(misc, int, 16)
#undef CARVE
    return off;
}

size_t track_lds_bytes(const DevCfg &c)
{
    size_t b = track_lds_layout<false>(c, nullptr, nullptr);
    if (pred_in_track(c) && b < (size_t)16 * kPredScratch * sizeof(double)) b = (size_t)16 * kPredScratch * sizeof(double);  // the predict stage's scratch
#ifdef MMW_STAMPS
    // diagnostic build only: MMW_DIAG_LDS_EXTRA=<bytes> inflates the allocation to lower the number of
    // resident workgroups per CU (separates latency from contention in the phase stamps)
    if (const char *x = getenv("MMW_DIAG_LDS_EXTRA")) b += (size_t)atoi(x);
#endif
    return b;
}

// numpy pairwise_sum_DOUBLE (the summation order of the 1-D np.mean in ClusterTrack._get_D, Tracking.py:286):
//   n < 8      : one by one
//   n <= 128   : eight interleaved accumulators r[k] += x[i+k], ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the
//                n%8 leftovers one by one                                   -- a LEAF
//   otherwise  : n2 = n/2 - (n/2)%8 ;  pairwise(x, n2) + pairwise(x+n2, n-n2)
// The leaves of one sum are independent, so they are spread over lanes (pw_leaf) and only the few adds of
// the recursion (pw_combine) stay serial.  D = recursion depth budget: 4 levels cover n <= 2048.

template <int D, typename F>
__device__ __forceinline__ void pw_for_each_leaf(int off, int n, F f)
{
    if constexpr (D == 0) f(off, n);
    else {
        if (n <= 128) f(off, n);
        else { const int n2 = pw_split(n); pw_for_each_leaf<D - 1>(off, n2, f); pw_for_each_leaf<D - 1>(off + n2, n - n2, f); }
    }
}

// sums of the leaves, in leaf order, back into the value numpy returns
template <int D>
__device__ __forceinline__ double pw_combine(int n, const double *leafsum, int stride, int &idx)
{
    if constexpr (D == 0) { const double v = leafsum[idx * stride]; idx++; return v; }
    else {
        if (n <= 128) { const double v = leafsum[idx * stride]; idx++; return v; }
        const int n2 = pw_split(n);
        const double l = pw_combine<D - 1>(n2, leafsum, stride, idx);
        const double r = pw_combine<D - 1>(n - n2, leafsum, stride, idx);
        return l + r;
    }
}
constexpr int kPwDepth = 4;

// one leaf (n <= 128) of sum_r (pa[r]-ca)*(pb[r]-cb)
__device__ __forceinline__ double pw_leaf(const double *pa, const double *pb, double ca, double cb, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += (pa[i] - ca) * (pb[i] - cb);
        return res;
    }
    const int lim = n - (n & 7);
    double r[8], xa[8], xb[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { xa[u] = pa[u]; xb[u] = pb[u]; }
#pragma unroll
    for (int u = 0; u < 8; u++) r[u] = (xa[u] - ca) * (xb[u] - cb);
    for (int i = 8; i < lim; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) { xa[u] = pa[i + u]; xb[u] = pb[i + u]; }
#pragma unroll
        for (int u = 0; u < 8; u++) r[u] += (xa[u] - ca) * (xb[u] - cb);
    }
    const int left = n - lim;  // the n%8 leftovers, loaded together, added one by one
#pragma unroll
    for (int u = 0; u < 7; u++) { xa[u] = (u < left) ? pa[lim + u] : 0.0; xb[u] = (u < left) ? pb[lim + u] : 0.0; }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int u = 0; u < 7; u++) if (u < left) res += (xa[u] - ca) * (xb[u] - cb);
    return res;
}

// PPT = points per thread = ceil(max_pts / 256): a template so that per-point registers are not
// reserved for points a configuration can never have.
// INNER = mmw_config.seek_inner (per-track ring sizes, see k_inner in k_dbscan.hip): a template so that the default
// instantiation carries none of it.
// PRED = _predict_all at the head of this kernel instead of in k_predict (small contexts, mmw_kalman.hpp: pred_in_track).
// F32 = the frame's rows are fp32 (mmw_step_f32), promoted to fp64 as they are loaded (mmw_device.hpp: load_point_row).
template <int PPT, bool INNER, bool PRED = false, bool F32 = false>
__global__ __launch_bounds__(kThreads, (PRED ? 2 : (PPT == 2 ? 5 : (PPT == 1 ? 4 : 3)))) void k_track(DevCfg cfg, DevState st, const void *__restrict__ pts_all,
                                                    const int32_t *__restrict__ n_pts, const double *__restrict__ dt_all,
                                                    int32_t *__restrict__ assoc_out, int32_t *__restrict__ db_n_out,
                                                    int32_t *__restrict__ db_labels_out, int UM_out, int parity)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    TrackLds L;
    track_lds_layout<true>(cfg, lds_raw, &L);

    // Workgroups are dispatched in index order, so the order of the scenes is a schedule: st.perm lists the
    // scenes with the most tracks (the longest workgroups) first, which shortens the tail of the launch.
    const int s = st.perm[(size_t)parity * cfg.n_scenes + blockIdx.x];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    PROBE(19);
    WGTIME(0);
    // The single-wave job below (track maintenance) goes to wave `role == 0`, rotated by scene so that the
    // resident workgroups of a CU do not all queue it on the same SIMD.
    const int role = (wave + s) & (kWaves - 1);
    const int NP = cfg.max_pts, CLS = cfg.t_cap + 1;
    const int NPs = NP + kTilePad;  // column stride of the point tile
    // Everything the first phases read goes out in ONE batch of independent global loads: the scene's
    // track order first (its completion is all the first barrier has to wait for -- vmcnt retires in
    // order), then the points, WITHOUT waiting for the point count (rows past the count are allocated
    // memory, loaded speculatively and ignored), each row's 8 columns straight into registers: they are
    // gated from registers and only later parked in the LDS tile.
    const int my_slot = tid < cfg.t_cap ? st.order[(size_t)s * cfg.t_cap + tid] : 0;
    // ... and the three scalars the frame's first decisions hang on (point count, dt, track count), AHEAD of the
    // points: a count read after them retires after them, and a track count read only once the count has been
    // looked at is a second full round trip before the first barrier -- every workgroup paid both.
    const int n_raw = n_pts[s];
    SceneHdr *hdr = st.hdr + s;
    const double dt = dt_all[s];
    int T = hdr->n_tracks;
    double2 pr[PPT][4];
    {
        const void *frame = frame_of(pts_all, s, NP, F32);
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = q * kThreads + tid;
            load_point_row<F32>(frame, i, i < NP, pr[q]);
        }
    }
    const int n = n_raw < 0 ? 0 : n_raw;  // MMW_EMPTY_FRAME: track() on an empty cloud
    if (tid == 0 && db_n_out) db_n_out[s] = -1;
    if (s == 0 && tid < 4) st.db_count[(parity ^ 1) * 4 + tid] = 0;  // next step's work-list lengths
    if (s == 0 && tid >= 4 && tid < 7) st.q[(parity ^ 1) * 8 + (tid - 4)] = tid - 4 == kQHead ? q_tag(cfg.epoch + 1) : 0;  // ... and its queue counters (kQCount, kQHead -- tagged with the step it will serve --, kQDone)
    if (s == 0 && tid >= 8 && tid < 11) st.q[kQBig + (parity ^ 1) * 8 + (tid - 8)] = tid - 8 == kQHead ? q_tag(cfg.epoch + 1) : 0;  // ... and those of the large clouds' queue
    if (s == 0 && tid < kUpdWords) st.upd_count[(parity ^ 1) * kUpdWords + tid] = 0;       // ... and the lengths of its update lists
    if (s == 0 && tid == kThreads - 1) st.spc_count[parity ^ 1] = 0;
    if (!frame_reaches_track(n_raw, NP)) {  // offline_main.py:56: empty frames never reach track()
        if (tid == 0) {
            hdr->need_db = 0;
            hdr->skipped = (hdr->skipped & ~255) | 1;  // not in this frame's update lists: the next k_predict finds its tracks by this flag (the ring's size and non-finite flags stay)
            if (n_raw != 0) atomicOr(&hdr->err, ERR_BADCOUNT);  // a count the context was not sized for
        }
        return;
    }
    int32_t *order = st.order + (size_t)s * cfg.t_cap;
    TrackRec *trk = st.trk + (size_t)s * cfg.t_cap;
    const int Tin = T;
    int err = 0;
#ifdef MMW_STAMPS
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
    PROBE(16);
    if constexpr (PRED) {
        // _predict_all (Tracking.py:591-596) + the gate matrices, as k_predict does them: a 16-lane group per track, sixteen
        // tracks per round over the four waves; scratch = the head of this kernel's LDS (nothing else lives there yet).
        // What the groups store (track records, gate_buf) is read below by every wave of THIS workgroup -- gate_buf through
        // the SCALAR cache, which has not seen these lines in this launch but may hold a neighbour scene's share of a line:
        // stores acknowledged by the L2 (vmcnt(0): the vector L1 writes through), workgroup fence + barrier, scalar-cache
        // invalidate.  (A device-scope fence here writes the L2 back -- the XCDs' L2s are not coherent -- and cost 20 us.)
        double *Wj = reinterpret_cast<double *>(lds_raw) + (size_t)(wave * 4 + (lane >> 4)) * kPredScratch;
        int perr = 0;
        for (int j0 = wave * 4; j0 < T; j0 += 4 * kWaves) {
            const int j = j0 + (lane >> 4);
            const bool live = j < T;
            TrackRec *rec = trk + (live ? order[j] : 0);
            if (cfg.dx == 9) predict_one_track<9>(cfg, st, rec, live, s, j, dt, Wj, lane, lane & 15, perr);
            else predict_one_track<6>(cfg, st, rec, live, s, j, dt, Wj, lane, lane & 15, perr);
        }
        PROBE(17);
        if (perr) atomicOr(&hdr->err, perr);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_s_dcache_inv();
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the invalidate has completed before the first gate record is requested
        PROBE(18);
    }

    if (tid < cfg.t_cap) L.slot[tid] = my_slot;
    for (int j = tid + kThreads; j < cfg.t_cap; j += kThreads) L.slot[j] = order[j];
    if (tid == 0) { L.ml[0] = 0; L.ml[1] = 0; L.misc[15] = 0; }
    lds_barrier();
    STAMP(0);  // issue point loads
    PROBE(0);

    double bestd[PPT];
    int bestj[PPT];
#pragma unroll
    for (int q = 0; q < PPT; q++) { bestd[q] = 0.0; bestj[q] = -1; }

    // ---- gate every point against the scene's tracks (Tracking.py:553-572) ----
    // The gate record of a track (C^-1, log|det C|, predicted position: 43 doubles, k_predict -> gate_buf) is the same
    // for every point, i.e. wave-uniform: it is read through the SCALAR cache (constant address space, uniform address
    // -> s_load) and enters the fp64 VALU ops as their SGPR operand (no LDS staging, no barriers in this phase).
    // y' C^-1 y as k-ordered FUSED chains, v_k = fma(y_a, Ci[a][k], v_k) row by row over C^-1, then q = fma(v_k, y_k, q):
    // the arithmetic definition the oracle shares (oracle/c/mmw_oracle.c, _calc_dist_fun); with the operands in SGPRs the
    // phase is bound by fp64 issue, and the fused form is 49 instead of 84 instructions per (point, track).
    {
#ifdef MMW_DIAG_VGATE   // (diagnostic build, scripts/dual_run.py: the records by VECTOR loads -- volatile global -- instead of through the scalar cache)
        typedef const volatile double *gate_ptr;
#else
        typedef const double __attribute__((address_space(4))) *gate_ptr;
#endif
        const int su = __builtin_amdgcn_readfirstlane(s), Tu = __builtin_amdgcn_readfirstlane(T);
        gate_ptr gb = (gate_ptr)(st.gate_buf + (size_t)su * cfg.t_cap * kGateRec);
        if constexpr (PRED) {
            // The records were written by this launch.  The constant address space promises the compiler memory that does not
            // change, so the pointer itself is made opaque HERE, behind the invalidate: no load through it can be moved above
            // this statement.
            asm volatile("; mmw: gate pointer opaque from here" : "+s"(gb) : : "memory");
#ifndef MMW_DIAG_VGATE
            // Warm the scalar cache: one dword of every 64-byte line of the scene's gate records, all requests in flight
            // together.  The loop below then takes its records (six s_loads per track, waited for as a batch) from the
            // scalar cache instead of paying an L2 round trip per track -- with two or three workgroups per CU nothing hides
            // that (19.7 k -> 15.4 k cycles for eight tracks).  (Not in the bulk instantiations: five workgroups per CU hide
            // it, and their 96-register budget has no room for the sixteen addresses.)
            typedef const int __attribute__((address_space(4))) *line_ptr;
            const unsigned long long a0 = (unsigned long long)gb & ~63ULL;
            const int lines = (int)((((unsigned long long)gb + (unsigned long long)Tu * kGateRec * 8 + 63ULL) & ~63ULL) - a0) >> 6;
            line_ptr w = (line_ptr)a0;
            int acc = 0;
            for (int l0 = 0; l0 < lines; l0 += 16) {
                int t[16];
#pragma unroll
                for (int u = 0; u < 16; u++) { const int l = l0 + u < lines ? l0 + u : lines - 1; t[u] = w[l * 16]; }
#pragma unroll
                for (int u = 0; u < 16; u++) acc |= t[u];
            }
            asm volatile("" : : "s"(acc));
#endif
        }
        PROBE(32);
        for (int j = 0; j < Tu; j++) {
            if (j < 8) PROBE(33 + j);
            gate_ptr G = gb + j * kGateRec;
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = q * kThreads + tid;
                if (q * kThreads < n) {  // wave-uniform
                    const double y0 = pr[q][0].x - G[37], y1 = pr[q][0].y - G[38], y2 = pr[q][1].x - G[39], y3 = pr[q][1].y - G[40],
                                 y4 = pr[q][2].x - G[41], y5 = pr[q][2].y - G[42];
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
                    if (i < n && d < cfg.tr_gate) {
                        if (bestj[q] < 0 || d < bestd[q]) { bestj[q] = j; bestd[q] = d; }
                    }
                }
            }
        }
    }
    STAMP(1);
    STAMP(2);  // gating
    PROBE(2);
    // ---- _get_gated_clouds: order-preserving split by class (Tracking.py:605-629) ----
    int gp_len, gp_skw, gp_s[MMW_RING_MAX], gp_n[MMW_RING_MAX];   // (thread kThreads - 1: the global ring's header words, see below)
    {
        const int NB = (n + 63) / 64;
        unsigned long long mybal[PPT];
        // ring state of track `tid` (thread T: of the global ring), needed after the scans below: requested now so
        // that the global round trip runs under the ballots and scans (T <= t_cap <= 64 < 256 threads)
        int sd_len = 0, sd_rs[MMW_RING_MAX] = {0, 0, 0, 0}, sd_size = cfg.ring;
        gp_len = 0; gp_skw = 0;
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) { gp_s[k] = 0; gp_n[k] = 0; }
        if (tid < T) {
            const TrackRec *rec = trk + L.slot[tid];
            sd_len = rec->ring_len;
            if (INNER) sd_size = rec->inner & 255;  // track.batch.size after change_buffer_size (Tracking.py:60-64)
#pragma unroll
            for (int k = 0; k < MMW_RING_MAX; k++) sd_rs[k] = rec->ring_slot[k];
        } else if (tid == T) {
            sd_len = hdr->g_len;
            if (INNER && hdr_ring_size(hdr->skipped)) sd_size = hdr_ring_size(hdr->skipped);  // BatchedData.change_buffer_size on the global ring
#pragma unroll
            for (int k = 0; k < MMW_RING_MAX; k++) sd_rs[k] = hdr->g_slot[k];
        } else if (tid == kThreads - 1) {
            // the thread that pushes this frame into the global ring (behind the split): the ring's header words, requested NOW --
            // as one thread's fresh round trip plus a serial walk over LDS-indexed copies behind the barrier the push held its wave,
            // and with it the whole workgroup at the statistics' barrier, for ~7 k of a workgroup's 77 k cycles (probe timeline,
            // profiles/NOTEBOOK.md round 5)
            gp_len = hdr->g_len;
            gp_skw = hdr->skipped;
#pragma unroll
            for (int k = 0; k < MMW_RING_MAX; k++) { gp_s[k] = hdr->g_slot[k]; gp_n[k] = hdr->g_n[k]; }
        }
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = q * kThreads + tid;
            const int blk = q * kWaves + wave;
            mybal[q] = 0;
            if (q == 0) PROBE(25);
            if (blk < NB) {  // wave-uniform
                const int cls = (i < n) ? bestj[q] + 1 : -1;
                if (i < n && assoc_out) assoc_out[(size_t)s * NP + i] = bestj[q];
                for (int c = 0; c <= T; c++) {
                    unsigned long long b = __ballot(cls == c);
                    if (cls == c) mybal[q] = b;
                    if (lane == 0) L.cnt[blk * CLS + c] = __popcll(b);
                }
            }
        }
        PROBE(26);
        lds_barrier();
        PROBE(27);
        for (int c = tid; c <= T; c += kThreads) {  // exclusive scan over the 64-point blocks: all counts first, then the adds
            int t[PPT * kWaves];
#pragma unroll
            for (int b = 0; b < PPT * kWaves; b++) t[b] = b < NB ? L.cnt[b * CLS + c] : 0;
            int run = 0;
#pragma unroll
            for (int b = 0; b < PPT * kWaves; b++) { if (b < NB) L.cnt[b * CLS + c] = run; run += t[b]; }
            L.cls_n[c] = run;
        }
        lds_barrier();
        PROBE(28);
        for (int c = tid; c <= T + 1; c += kThreads) {  // class offsets: every class adds up the sizes before it
            int run = 0;
            for (int k = 0; k < c; k++) run += L.cls_n[k];
            L.cls_off[c] = run;
        }
        // where this frame's rows go, from the ring state BEFORE this frame's push (requested at the start of the
        // split, see there): a full ring recycles its oldest slot (BatchedData.add_frame, Tracking.py:43-51)
        if (tid <= T) {
            int phys = sd_rs[0];
            if (INNER) {
                // a ring whose size has shrunk pops p > 1 frames (add_frame's loop, Tracking.py:47-48); each pop rotates the
                // freed physical slot behind the live ones, so the new frame lands in the slot popped LAST (or, without a
                // pop, in the first unused one)
                const int keepn = sd_len < sd_size - 1 ? sd_len : (sd_size - 1 > 0 ? sd_size - 1 : 0);
                const int want = sd_len > keepn ? sd_len - keepn - 1 : sd_len;
#pragma unroll
                for (int k = 1; k < MMW_RING_MAX; k++) if (k == want) phys = sd_rs[k];
            } else {
#pragma unroll
                for (int k = 1; k < MMW_RING_MAX; k++) if (sd_len < cfg.ring && k == sd_len) phys = sd_rs[k];
            }
            L.seg_dst[tid] = tid < T ? (((long long)s * cfg.t_cap + L.slot[tid]) * cfg.ring + phys) * (long long)cfg.ring_rows * 8
                                     : ((long long)s * cfg.ring + phys) * (long long)NP * 8;
        }
        lds_barrier();
        PROBE(30);
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = q * kThreads + tid;
            if (i < n) {
                // park the point in the LDS tile (SoA, 6 columns) at its CLASS-SORTED position: every cloud is
                // then a contiguous run in input order, and the per-cluster sums below read it with
                // consecutive addresses instead of chasing perm[].  (The gate area the tile overlays is dead.)
                const int cls = bestj[q] + 1, blk = q * kWaves + wave;
                const int local = L.cnt[blk * CLS + cls] + __popcll(mybal[q] & lanemask_lt());  // rank inside its cloud
                const int pos = L.cls_off[cls] + local;
                L.p6[0 * NPs + pos] = pr[q][0].x; L.p6[1 * NPs + pos] = pr[q][0].y;
                L.p6[2 * NPs + pos] = pr[q][1].x; L.p6[3 * NPs + pos] = pr[q][1].y;
                L.p6[4 * NPs + pos] = pr[q][2].x; L.p6[5 * NPs + pos] = pr[q][2].y;
                // ... and its full row (8 columns) goes to the ring it belongs to: track j keeps the first
                // ring_rows rows of its cloud (Tracking.py:341 via BatchedData), the global ring all unassigned rows
                if (cls == 0 || local < cfg.ring_rows) {
                    double *dst = (cls == 0 ? st.g_ring + L.seg_dst[T] : st.trk_ring + L.seg_dst[cls - 1]) + (size_t)local * 8;
                    double2 *d2 = reinterpret_cast<double2 *>(dst);
                    d2[0] = pr[q][0]; d2[1] = pr[q][1]; d2[2] = pr[q][2]; d2[3] = pr[q][3];
                }
                // a NaN / an infinite value in a row that enters the global ring (all 8 columns: the gate only looked at six,
                // and never takes a point whose innovation is not finite): sklearn will refuse the cloud (see the trigger)
                if (cls == 0) {
                    const int nfb = row_nonfinite_bits(pr[q]);
                    if (nfb) atomicOr(&L.misc[15], nfb);
                }
            }
        }
        PROBE(31);
        lds_barrier();
    }
    STAMP(3);  // class split
    // ---- batch.add_frame(unassigned) on the global ring (Tracking.py:689-691), early: the DBSCAN screen at the
    //      end of the kernel wants the older frames' rows, and their loads can be in flight during the phases
    //      in between.  (The split above was the last reader of the ring state before this frame.) ----
    if (tid == kThreads - 1) {
        // (registers and unrolled selects throughout: no dependent LDS or global read but the one of this frame's count)
        int len = gp_len;
        const int gsz = (INNER && hdr_ring_size(gp_skw)) ? hdr_ring_size(gp_skw) : cfg.ring;
#pragma unroll
        for (int it = 0; it < MMW_RING_MAX; it++) {   // pop_frame while len >= size (a ring of fixed size pops at most once)
            if (len >= gsz && len > 0) {
                const int first = gp_s[0];
#pragma unroll
                for (int k = 1; k < MMW_RING_MAX; k++) if (k < len) { gp_s[k - 1] = gp_s[k]; gp_n[k - 1] = gp_n[k]; }
#pragma unroll
                for (int k = 0; k < MMW_RING_MAX; k++) if (k == len - 1) gp_s[k] = first;
                len--;
            }
        }
        const int nun = L.cls_n[0];
        int phys = gp_s[0];
#pragma unroll
        for (int k = 1; k < MMW_RING_MAX; k++) if (k == len) phys = gp_s[k];
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) if (k == len) gp_n[k] = nun;
        len++;
        int U = 0;
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) { if (k >= len) gp_n[k] = 0; U += gp_n[k]; }
        int *gs = L.misc + 4, *gn = L.misc + 8;   // (what the later phases read: slots and counts of the live frames, oldest first)
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) { gs[k] = gp_s[k]; gn[k] = gp_n[k]; hdr->g_slot[k] = gp_s[k]; hdr->g_n[k] = gp_n[k]; }
        hdr->g_len = len;
        hdr->db_u = U;
        L.misc[1] = phys;
        L.misc[3] = len;
        L.misc[13] = U;
        // the ring's non-finite flags (two bits per physical slot, SceneHdr.skipped): this frame's replace those of the slot it
        // was written to; [15] = the new flags | what the live frames hold together << 8, for the trigger below
        const int nff = nf_flags_with((gp_skw >> kSkipNfShift) & kSkipNfMask, phys, L.misc[15]);
        int live = 0;
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) if (k < len) live |= (nff >> (2 * gp_s[k])) & 3;
        L.misc[15] = nff | (live << 8);
    }
    PROBE(3);

    // ---- associate_pointcloud (Tracking.py:314-341): PointCluster stats, N_est, spread_est.
    //      The column sums are sequential in row order (np.mean(axis=0)), i.e. serial chains, so the chain
    //      carries nothing but the add: every wave takes two tracks per round, lanes 0..11 walk the twelve
    //      (track, column) sums while lanes 16..63 form min/max over four contiguous quarters of each run
    //      (combined in order with the sequential rule "a later value wins only if strictly smaller/larger",
    //      which is what a single left-to-right pass yields) ----
    // The (track, column) epilogue: centroid, spread estimate, N_est, lifetime, the leaves of a large cloud's pairwise sums
    auto finish = [&](int j, int m, int nj, double sum, double mn, double mx, double old, double ne_old, TrackRec *rec) {
        if (nj == 0) {
            if (m == 0) rec->lifetime += dt;  // update_lifetime(dt) Tracking.py:400-407
            return;
        }
        const double cen = sum / (double)nj;
        L.cen[j * 6 + m] = cen;
        rec->centroid[m] = cen;
        rec->minv[m] = mn;
        rec->maxv[m] = mx;
        // _estimate_measurement_spread Tracking.py:246-268
        double spread = mx - mn;
        const double lim = cfg.kf_spread_lim[m], lim2 = 2 * lim;
        if (nj != 1) spread = spread * (double)(nj + 1) / (double)(nj - 1);
        spread = spread < lim2 ? spread : lim2;
        spread = spread > lim ? spread : lim;
        rec->spread[m] = spread > old ? spread : (1.0 - cfg.kf_a_spr) * old + cfg.kf_a_spr * spread;
        if (m == 0) {
            if (nj > 128) {  // leaves of this cloud's pairwise sums, for the dispersion phase below
                int cnt = 0;
                pw_for_each_leaf<kPwDepth>(0, nj, [&](int, int) { cnt++; });
                const int first = atomicAdd(&L.ml[0], cnt), c = atomicAdd(&L.ml[1], 1);
                int *lf = L.ml + 2 + first * 3, *cl = L.ml + 2 + 3 * pw_max_leaves(NP);
                cl[c * 2] = j; cl[c * 2 + 1] = first;
                int k = 0;
                pw_for_each_leaf<kPwDepth>(0, nj, [&](int o, int len) { lf[k * 3] = j; lf[k * 3 + 1] = o; lf[k * 3 + 2] = len; k++; });
            }
            rec->lifetime = 0.0;
            rec->point_num = nj;
            // _estimate_point_num Tracking.py:232-244
            double ne = ne_old;
            if (cfg.kf_enable_est) ne = ((double)nj > ne) ? (double)nj : (1 - cfg.kf_a_n) * ne + cfg.kf_a_n * (double)nj;
            else ne = cfg.kf_est_pointnum > (double)nj ? cfg.kf_est_pointnum : (double)nj;
            rec->n_est = ne;
            L.nest[j] = ne;
        }
    };
    if (T <= 10) {  // uniform
        // Up to ten tracks: ALL column sums on ONE wave (six lanes per track: the loop body is issued once for the scene, not
        // once per pair of tracks), the min / max passes on the other three (192 lanes: S slices per (track, column), S the
        // largest power of two that fits); the lane that ends up with a column's min / max finishes that column behind a
        // barrier, with the sum the other wave left in LDS.
        double mn = __longlong_as_double(0x7ff0000000000000LL), mx = -mn, old = 0.0, ne_old = 0.0;
        int fg = -1;   // the (track, column) = fg / 6, fg % 6 this lane finishes behind the barrier
        if (role == 0) {
            const int grp = lane / 6, m = lane - grp * 6, j = grp;
            const bool valid = lane < 60 && j < T;
            const int nj = valid ? L.cls_n[j + 1] : 0, off = valid ? L.cls_off[j + 1] : 0;
            if (nj > 0) {
                const double *col = L.p6 + m * NPs + off;
                double sum = 0.0;
                int r = 0;
                double v[8], w[8];
                if (nj >= 8) {  // eight rows in flight, the next eight requested before these are added
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = col[u];
                    for (r = 8; r + 8 <= nj; r += 8) {
#pragma unroll
                        for (int u = 0; u < 8; u++) w[u] = col[r + u];
#pragma unroll
                        for (int u = 0; u < 8; u++) sum += v[u];
#pragma unroll
                        for (int u = 0; u < 8; u++) v[u] = w[u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) sum += v[u];
                }
                {   // up to seven left
                    const int left = nj - r;
#pragma unroll
                    for (int u = 0; u < 7; u++) v[u] = (u < left) ? col[r + u] : 0.0;
#pragma unroll
                    for (int u = 0; u < 7; u++) if (u < left) sum += v[u];
                }
                L.cen[j * 6 + m] = sum;   // (the column's owner replaces it by the centroid below)
            }
        } else {
            int lg = 5;   // S = 2^lg slices: 6 T S <= 192
            while (lg > 0 && 6 * T * (1 << lg) > 192) lg--;
            const int task = (role - 1) * 64 + lane, g = task >> lg, slice = task & ((1 << lg) - 1);
            const int j = g / 6, m = g - j * 6;
            const bool valid = g < 6 * T;
            const int nj = valid ? L.cls_n[j + 1] : 0, off = valid ? L.cls_off[j + 1] : 0;
            TrackRec *rec = trk + L.slot[valid ? j : 0];
            if (valid && slice == 0) {
                fg = g;
                if (nj > 0) {  // issued now, consumed behind the barrier
                    old = rec->spread[m];
                    if (m == 0) ne_old = rec->n_est;
                }
            }
            if (nj > 0) {
                const double *col = L.p6 + m * NPs + off;
                const int r0 = (nj * slice) >> lg, r1 = (nj * (slice + 1)) >> lg;
                if (r0 < r1) { mn = col[r0]; mx = mn; }
                int r = r0 + 1;
                for (; r + 8 <= r1; r += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = col[r + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) { mn = v[u] < mn ? v[u] : mn; mx = v[u] > mx ? v[u] : mx; }
                }
                {
                    const int left = r1 - r;
                    double v[7];
#pragma unroll
                    for (int u = 0; u < 7; u++) v[u] = (u < left) ? col[r + u] : mn;  // mn itself never wins a strict compare
#pragma unroll
                    for (int u = 0; u < 7; u++) if (u < left) { mn = v[u] < mn ? v[u] : mn; mx = v[u] > mx ? v[u] : mx; }
                }
            }
            for (int d = 1; d < (1 << lg); d <<= 1) {  // neighbouring slices, then pairs of them, ...: the partner holds the LATER rows
                const double tn = __shfl_down(mn, d), tx = __shfl_down(mx, d);
                mn = tn < mn ? tn : mn;
                mx = tx > mx ? tx : mx;
            }
        }
        lds_barrier();
        if (fg >= 0) {
            const int fj = fg / 6, fn = L.cls_n[fj + 1];
            finish(fj, fg - fj * 6, fn, fn > 0 ? L.cen[fg] : 0.0, mn, mx, old, ne_old, trk + L.slot[fj]);
        }
    } else {
        for (int jb = 0; jb < T; jb += 2 * kWaves) {
            const bool grpA = lane < 12, grpB = lane >= 16;
            const int pi = grpA ? lane : (grpB ? (lane - 16) >> 2 : 0), slice = (lane - 16) & 3;
            const int jj = pi / 6, m = pi - jj * 6, j = jb + wave * 2 + jj;
            const bool valid = j < T && (grpA || grpB);
            const int nj = valid ? L.cls_n[j + 1] : 0, off = valid ? L.cls_off[j + 1] : 0;
            TrackRec *rec = trk + L.slot[valid ? j : 0];
            const double *col = L.p6 + m * NPs + off;
            double old = 0.0, ne_old = 0.0, sum = 0.0;
            PROBE(20);
            if (grpA && nj > 0) {
                old = rec->spread[m];  // issued now, consumed after the chain
                if (m == 0) ne_old = rec->n_est;
                // LDS latency is what this chain waits for, not the adds: eight loads in flight, the next
                // batch requested before the current one is summed
                int r = 0;
                double v[8], w[8];
                if (nj >= 8) {
#pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = col[u];
                    for (r = 8; r + 8 <= nj; r += 8) {
#pragma unroll
                        for (int u = 0; u < 8; u++) w[u] = col[r + u];
#pragma unroll
                        for (int u = 0; u < 8; u++) sum += v[u];
#pragma unroll
                        for (int u = 0; u < 8; u++) v[u] = w[u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) sum += v[u];
                }
                {   // up to seven left
                    const int left = nj - r;
    #pragma unroll
                    for (int u = 0; u < 7; u++) v[u] = (u < left) ? col[r + u] : 0.0;
    #pragma unroll
                    for (int u = 0; u < 7; u++) if (u < left) sum += v[u];
                }
            }
            PROBE(21);
            double mn = __longlong_as_double(0x7ff0000000000000LL), mx = -mn;  // empty quarter: never wins
            if (grpB && nj > 0) {
                const int r0 = (nj * slice) >> 2, r1 = (nj * (slice + 1)) >> 2;
                if (r0 < r1) { mn = col[r0]; mx = mn; }
                int r = r0 + 1;
                for (; r + 8 <= r1; r += 8) {
                    double v[8];
    #pragma unroll
                    for (int u = 0; u < 8; u++) v[u] = col[r + u];
    #pragma unroll
                    for (int u = 0; u < 8; u++) { mn = v[u] < mn ? v[u] : mn; mx = v[u] > mx ? v[u] : mx; }
                }
                {
                    const int left = r1 - r;
                    double v[7];
    #pragma unroll
                    for (int u = 0; u < 7; u++) v[u] = (u < left) ? col[r + u] : mn;  // mn itself never wins a strict compare
    #pragma unroll
                    for (int u = 0; u < 7; u++) if (u < left) { mn = v[u] < mn ? v[u] : mn; mx = v[u] > mx ? v[u] : mx; }
                }
            }
            PROBE(22);
    #pragma unroll
            for (int d = 1; d <= 2; d <<= 1) {  // quarters (0,1),(2,3) then halves: the partner holds the LATER rows
                const double tn = __shfl_down(mn, d), tx = __shfl_down(mx, d);
                mn = tn < mn ? tn : mn;
                mx = tx > mx ? tx : mx;
            }
            double *wmm = L.wmm + wave * 24;
            if (grpB && slice == 0) { wmm[pi * 2] = mn; wmm[pi * 2 + 1] = mx; }
            wave_sync();
            PROBE(23);
            if (grpA && valid) finish(j, m, nj, sum, nj > 0 ? wmm[pi * 2] : 0.0, nj > 0 ? wmm[pi * 2 + 1] : 0.0, old, ne_old, rec);
            wave_sync();
        }
    }
    PROBE(24);
    lds_barrier();
    STAMP(4);  // centroid/min/max/spread
    PROBE(4);
    // this thread's point of the DBSCAN cloud (the global ring, oldest frame first), for the screen at the end:
    // older frames from global memory (requested now), this frame's unassigned rows from the tile, where
    // they are the first run
    double sx = 0.0, sy = 0.0, sz = 0.0;
    {
        const int Ucl = L.misc[13], nun = L.cls_n[0];
        if (Ucl <= 256 && tid < Ucl) {
            const int old_n = Ucl - nun;  // rows of the older frames
            if (tid >= old_n) {
                const int k = tid - old_n;
                sx = L.p6[0 * NPs + k]; sy = L.p6[1 * NPs + k]; sz = L.p6[2 * NPs + k];
            } else {
                const int *gs = L.misc + 4, *gn = L.misc + 8;
                int f = 0, base = 0;  // frame of row `tid`
#pragma unroll
                for (int k = 0; k < MMW_RING_MAX - 1; k++) if (tid >= base + gn[f] && f < L.misc[3] - 2) { base += gn[f]; f++; }
                const double *r = st.g_ring + ((size_t)s * cfg.ring + gs[f]) * (size_t)NP * 8 + (size_t)(tid - base) * 8;
                const double2 a = *reinterpret_cast<const double2 *>(r);
                sx = a.x; sy = a.y; sz = r[2];
            }
        }
    }
    // status: sqrt(sum(centroid[3:6]^2)) < TR_VEL_THRES (Tracking.py:132-136)
    // (the LAST wave does it: the dispersion items below fill the waves from the front, and this is a chain of
    //  global round trips)
    for (int j = kThreads - 1 - tid; j < T; j += kThreads) {
        const int nj = L.cls_n[j + 1];
        if (nj == 0 && INNER) trk[L.slot[j]].inner &= 255;  // associate_pointcloud did not run on this track
        if (nj > 0) {
            TrackRec *rec = trk + L.slot[j];
            const int rsize = INNER ? (rec->inner & 255) : cfg.ring;
            const double v3 = L.cen[j * 6 + 3], v4 = L.cen[j * 6 + 4], v5 = L.cen[j * 6 + 5];
            rec->is_static = sqrt((v3 * v3 + v4 * v4) + v5 * v5) < cfg.tr_vel_thres ? 1 : 0;
            // BatchedData.add_frame on the track ring (Tracking.py:43-51); the rows were written above.
            // ring_len, ring_n[], ring_slot[] come in together and go back together.
            int len = rec->ring_len;
            int rn[MMW_RING_MAX], rs[MMW_RING_MAX];
#pragma unroll
            for (int k = 0; k < MMW_RING_MAX; k++) { rn[k] = rec->ring_n[k]; rs[k] = rec->ring_slot[k]; }
            while (len >= rsize && len > 0) {  // pop_frame: the freed physical slot becomes the first free entry
                const int first = rs[0];
#pragma unroll
                for (int k = 1; k < MMW_RING_MAX; k++) if (k < len) { rs[k - 1] = rs[k]; rn[k - 1] = rn[k]; }
#pragma unroll
                for (int k = 0; k < MMW_RING_MAX; k++) if (k == len - 1) rs[k] = first;
                len--;
            }
#pragma unroll
            for (int k = 0; k < MMW_RING_MAX; k++) if (k == len) rn[k] = nj;
#pragma unroll
            for (int k = 0; k < MMW_RING_MAX; k++) { rec->ring_n[k] = rn[k]; rec->ring_slot[k] = rs[k]; }
            rec->ring_len = len + 1;
            if (INNER) rec->inner = rsize | kInnerTouched;  // k_inner runs seek_inner_clusters on it
        }
    }
    // _estimate_group_disp_matrix + _get_D (Tracking.py:270-297): 21 symmetric entries per track, each the
    // 1-D np.mean of n products in numpy's pairwise order (pw_* above).  Work items: (track, entry) for
    // clouds of one leaf (n <= 128), (leaf, entry) for the leaves of larger clouds, whose sums meet in LDS
    // and are combined by one lane per (track, entry) after a barrier.
    {
        PROBE(10);
        double *leafsum = L.work + 6 * NPs;  // [leaf][21]
        const int nleaf = L.ml[0], ncloud = L.ml[1];
        const int *lf = L.ml + 2, *cl = L.ml + 2 + 3 * pw_max_leaves(NP);
        auto entry = [](int e, int &a, int &b) { a = 0; while (e >= 6 - a) { e -= 6 - a; a++; } b = a + e; };
        auto blend = [&](TrackRec *rec, int a, int b, double res, int nj, double g_ab, double g_ba, double ne) {
            const double D = res / (double)nj;
            if (ne == 0.0) { err |= ERR_DIVZERO; return; }
            const double al = (double)nj / ne;
            rec->gd[a * 6 + b] = (1 - al) * g_ab + al * D;
            if (a != b) rec->gd[b * 6 + a] = (1 - al) * g_ba + al * D;
        };
        for (int it = tid; it < (T + nleaf) * 21; it += kThreads) {
            const int u = it / 21;
            int a, b;
            entry(it - u * 21, a, b);
            const bool direct = u < T;
            const int j = direct ? u : lf[(u - T) * 3];
            const int nj = L.cls_n[j + 1];
            if (direct && (nj == 0 || nj > 128)) continue;
            const int off = L.cls_off[j + 1] + (direct ? 0 : lf[(u - T) * 3 + 1]), len = direct ? nj : lf[(u - T) * 3 + 2];
            TrackRec *rec = trk + L.slot[j];
            PROBE(11);
            double g_ab = 0.0, g_ba = 0.0, ne = 1.0;
            if (direct) { g_ab = rec->gd[a * 6 + b]; g_ba = rec->gd[b * 6 + a]; ne = L.nest[j]; }  // (loads in flight during the sum)
            const double res = pw_leaf(L.p6 + a * NPs + off, L.p6 + b * NPs + off, L.cen[j * 6 + a], L.cen[j * 6 + b], len);
            PROBE(12);
            if (direct) blend(rec, a, b, res, nj, g_ab, g_ba, ne);
            else leafsum[(u - T) * 21 + (it - u * 21)] = res;
        }
        PROBE(13);
        if (ncloud > 0) {  // uniform
            lds_barrier();
            PROBE(14);
            for (int it = tid; it < ncloud * 21; it += kThreads) {
                const int c = it / 21, e = it - c * 21;
                int a, b;
                entry(e, a, b);
                const int j = cl[c * 2], nj = L.cls_n[j + 1];
                TrackRec *rec = trk + L.slot[j];
                const double g_ab = rec->gd[a * 6 + b], g_ba = rec->gd[b * 6 + a], ne = L.nest[j];
                int idx = 0;
                const double res = pw_combine<kPwDepth>(nj, leafsum + cl[c * 2 + 1] * 21 + e, 21, idx);
                blend(rec, a, b, res, nj, g_ab, g_ba, ne);
            }
        }
        PROBE(15);
    }
    STAMP(5);  // dispersion matrices (wave 0's share)
    __syncthreads();  // full fence: maintenance reads is_static / lifetime other threads stored to the records
    {   // the point tile is dead from here on: clear the cell grid of the DBSCAN screen at the end of the kernel
        int *grid = reinterpret_cast<int *>(reinterpret_cast<char *>(L.work) + 4096);
        for (int i = tid; i < kCloudGrid; i += kThreads) grid[i] = 0;
        if (tid == 0) { grid[kCloudGrid] = 0; grid[kCloudGrid + 1] = 0; L.misc[12] = 0; }  // (mm[0] behind the grid)
    }
    STAMP(6);  // track ring rows + barrier (includes waiting for the other waves' dispersion work)

    // ---- _maintain_tracks (Tracking.py:513-528) ----
    if (role == 0) {
        const bool have = lane < T;
        bool keep = false;
        int sl = 0;
        if (have) {
            sl = L.slot[lane];
            const TrackRec *rec = trk + sl;
            const double lim = rec->is_static ? cfg.tr_lifetime_static : cfg.tr_lifetime_dynamic;
            keep = !(rec->lifetime > lim);
        }
        const unsigned long long kb = __ballot(have && keep), db = __ballot(have && !keep);
        const int nk = __popcll(kb);
        if (have) {
            const int pos = keep ? __popcll(kb & lanemask_lt()) : nk + __popcll(db & lanemask_lt());
            L.slot2[pos] = sl;
        }
        if (lane == 0) L.misc[0] = nk;
    }
    lds_barrier();
    {
        const int Told = T;
        T = L.misc[0];
        for (int j = tid; j < Told; j += kThreads) { L.slot[j] = L.slot2[j]; order[j] = L.slot2[j]; }
    }
    lds_barrier();
    PROBE(6);
    STAMP(7);  // maintenance

    // (_update_all, Tracking.py:598-603, is batched over all tracks in the next launch: k_post)
    // (track-wise layout only: this scene's tracks join the update list of its shard, mmw_device.hpp)
    const int upd_units = cfg.n_scenes * kalman_waves_per_scene(cfg.tr_max_tracks);
    const bool upd_on = tracks_dense(cfg, kalman_waves_per_scene(cfg.tr_max_tracks));
    const int upd_sh = (int)blockIdx.x % upd_shards(upd_units);
    int upd_pos = -1;
    STAMP(8);  // update
    // ---- DBSCAN trigger (Tracking.py:693-697) ----
    if (tid == 0) {
        const int U = L.misc[13];
        hdr->n_tracks = T;
        hdr->n_upd = T;
        const int nfw = L.misc[15];
        hdr->skipped = (INNER ? (hdr->skipped & (kSkipRingMask << kSkipRingShift)) : 0) | ((nfw & kSkipNfMask) << kSkipNfShift);
        // this scene's tracks join the update list of this workgroup's shard (k_post lays _update_all out over tracks): the
        // place is requested here and written at the very end of the kernel -- the atomic's round trip (a microsecond
        // under 4096 workgroups) must not sit in front of the screen below
        if (T > 0 && upd_on) upd_pos = atomicAdd(&st.upd_count[parity * kUpdWords + upd_sh], T);
        bool need = U > 0 && T < cfg.tr_max_tracks;
        const int nfe = nf_error_of(nfw >> 8);
        if (need && nfe) {
            // apply_DBscan is reached with a NaN / an infinite value in the ring: sklearn's input validation raises ValueError
            // (Utils.py:272-278) and track() ends here -- frame in the ring, nothing clustered, nothing cleared -- on every frame
            // the row is in the ring while the trigger holds.  With seek_inner the trigger is re-evaluated behind the inner
            // clusters (k_inner: they may fill the track list): the verdict is left to it (need_db = 2 | error bits << 2).
            if (INNER && cfg.seek_inner) hdr->need_db = 2 | (nfe << 2);
            else {
                err |= nfe;
                hdr->need_db = 0;
                if (db_n_out) db_n_out[s] = kDbRaised;
            }
            need = false;
        } else
            hdr->need_db = need ? 1 : 0;
        L.misc[2] = need ? U : 0;
    }
    if (err) atomicOr(&hdr->err, err);
    STAMP(9);  // global ring append
    // ---- apply_DBscan, first stage (Tracking.py:697, Utils.py:250-291).  The steady state is a ring of
    //      clutter in which no point can be a core point; for most scenes an O(U) cell count proves it
    //      (cloud_cells_prove_no_core, mmw_cloud.hpp) on the rows this workgroup has just appended, and the scene is
    //      finished with all labels -1.  The others go to the work lists of k_post / k_dbscan_big. ----
    PROBE(8);
    lds_barrier();
    const int Udb = L.misc[2];
    if (Udb > 0) {  // uniform
        bool listed = true;
        int *grid = reinterpret_cast<int *>(reinterpret_cast<char *>(L.work) + 4096);  // (cleared after the fence above)
        unsigned long long *mm = reinterpret_cast<unsigned long long *>(grid + kCloudGrid);
        if (Udb <= 256) {
            listed = !cloud_cells_prove_no_core(cfg, Udb, sx, sy, sz, mm, &L.misc[12], grid);
            if (listed) {  // (uniform)
                // second stage, the exact pair count, while this thread still holds its point of the cloud: 1-2 us here, and
                // only the handful of clouds per step that can hold a core point travel to the DBSCAN workers.  (Left to the
                // workers, a step's ~10^3 undecided clouds were a backlog the 8 side-stream workgroups could not clear beside
                // this launch: k_post's 256 worker blocks spent their first 19 us on it, in workgroup slots the Kalman update
                // was waiting for.)
                float4 *P4 = reinterpret_cast<float4 *>(L.work);                                   // [256]
                int *pcnt = reinterpret_cast<int *>(reinterpret_cast<char *>(L.work) + 12288);     // [256], behind the grid
                unsigned long long *mm2 = reinterpret_cast<unsigned long long *>(pcnt + 256);      // [3]
                listed = !cloud_pairs_prove_no_core_xyz<kThreads>(cfg, Udb, sx, sy, sz, P4, pcnt, mm2, &L.misc[14]);
            }
        } else {  // large clouds (no tracks yet, or lost): rows from the global ring (the fence above made this frame's visible)
            const int *gs = L.misc + 4, *gn = L.misc + 8;
            const int nfr = L.misc[3], big = 0x7fffffff;
            RowSrc src;
            src.gb = st.g_ring + (size_t)s * cfg.ring * (size_t)NP * 8;
            src.stride = (size_t)NP * 8;
            src.slots = (unsigned)gs[0] | ((unsigned)gs[1] << 8) | ((unsigned)gs[2] << 16) | ((unsigned)gs[3] << 24);
            src.c1 = nfr > 1 ? gn[0] : big;
            src.c2 = nfr > 2 ? gn[0] + gn[1] : big;
            src.c3 = nfr > 3 ? gn[0] + gn[1] + gn[2] : big;
            listed = !cloud_cells_prove_no_core_rows(cfg, src, Udb, mm, &L.misc[12], grid);
        }
        if (listed) {
            // Work list 3 = clouds <= 256 points, read by k_post after this launch.  The other two are QUEUES consumers claim
            // from, also WHILE this launch is running (chain workers on side streams; the kernels that follow on this stream take
            // what is left): queue 0 = the small clouds that can hold a new cluster (more points than a ring of clutter: kEarlyU),
            // i.e. the ones whose DBSCAN may be a ~60 us BallTree chain; queue 1 = the clouds of more than 256 points
            // (100-250 us each).  Everything this workgroup has stored for the scene must be visible device-wide before the
            // queue entry is -- one workgroup barrier + a release store, paid by those scenes only.
            // (work list 2 = the clouds the LDS cannot hold, k_dbscan_huge: only in contexts with ring * max_pts > kBigCloudMax)
            const int cls = Udb <= 256 ? (Udb >= kEarlyU && cfg.side_worker ? 0 : 3) : (Udb <= kBigCloudMax ? 1 : 2);  // uniform
            // (a device-scope release writes this XCD's L2 back: in the start-up frames, when EVERY scene pushes a large cloud
            //  and the side workers could take a handful, the entries are stored plainly for the kernels behind this one --
            //  cfg.big_live; frame 0 of 4096 scenes: k_track 386 -> 120 us)
            const bool live = cls == 0 || (cls == 1 && cfg.big_live);
            if (live) __syncthreads();
            if (tid == 0) {
                // (queues: the RELEASE of the entry's store is the only fence: it follows the workgroup barrier, so it covers
                //  what the other waves stored; a full __threadfence() here would also invalidate this CU's caches)
                int32_t *cnt = cls >= 2 ? st.db_count + parity * 4 + cls : st.q + cls * kQBig + parity * 8 + kQCount;
                const int pos = atomicAdd(cnt, 1);  // (< n_scenes: one push per scene and step)
                int32_t *e = st.db_list + (size_t)cls * cfg.n_scenes + pos;
                if (cls >= 2) *e = s;
                else if (live) __hip_atomic_store(e, s + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(e, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            cloud_finish_empty(st, hdr, s, Udb, UM_out, db_labels_out, db_n_out);
        }
    }
    PROBE(9);
    STAMP(10);  // DBSCAN screens + push
    if (tid == 0 && st.stats) {
        // algorithmic bytes of this scene-frame: points in, assoc out, per track the gate record
        // and the record fields this kernel reads and writes, unassigned rows appended to the global ring, rows
        // appended to track rings
        int ring_rows = 0;
        for (int j = 0; j < Tin; j++) ring_rows += min(L.cls_n[j + 1], cfg.ring_rows);
        unsigned long long *sl = stats_slot(st, s);
        atomicAdd(&sl[0], (unsigned long long)((F32 ? 32 : 64) * n + 4 * n + Tin * kTrackBytesPerTrack + 64 * L.cls_n[0] + 64 * ring_rows));
        atomicAdd(&sl[2], 1ULL);
        atomicAdd(&sl[5], (unsigned long long)Tin);
        atomicAdd(&sl[6], (unsigned long long)n * (unsigned long long)Tin);
    }
    if (tid < 64) {   // (wave 0: its lane 0 holds the place; T <= 63 entries, one store instruction)
        const int pos = __builtin_amdgcn_readfirstlane(upd_pos);
        if (pos >= 0 && tid < T) st.upd_list[((size_t)parity * kUpdShards + upd_sh) * upd_region(cfg.n_scenes, cfg.t_cap) + pos + tid] = upd_pack(s, tid, L.slot[tid]);
    }
    WGTIME(1);
}

template <int PPT, bool F32>
static void launch_track_t(const DevCfg &cfg, const DevState &st, const void *pts, const int32_t *n_pts, const double *dt,
                           int32_t *assoc, int32_t *db_n, int32_t *db_labels, int UM, int parity, hipStream_t stream)
{
    if (pred_in_track(cfg)) {  // (never with seek_inner)
        if (cfg.var_ring)
            mmw_launch(k_track<PPT, true, true, F32>, dim3(cfg.n_scenes), dim3(kThreads), track_lds_bytes(cfg), stream, cfg, st, pts, n_pts, dt, assoc,
                       db_n, db_labels, UM, parity);
        else
            mmw_launch(k_track<PPT, false, true, F32>, dim3(cfg.n_scenes), dim3(kThreads), track_lds_bytes(cfg), stream, cfg, st, pts, n_pts, dt, assoc,
                       db_n, db_labels, UM, parity);
        return;
    }
    if (cfg.seek_inner || cfg.var_ring)
        mmw_launch(k_track<PPT, true, false, F32>, dim3(cfg.n_scenes), dim3(kThreads), track_lds_bytes(cfg), stream, cfg, st, pts, n_pts, dt, assoc, db_n,
                   db_labels, UM, parity);
    else
        mmw_launch(k_track<PPT, false, false, F32>, dim3(cfg.n_scenes), dim3(kThreads), track_lds_bytes(cfg), stream, cfg, st, pts, n_pts, dt, assoc, db_n,
                   db_labels, UM, parity);
}

void launch_track(const DevCfg &cfg, const DevState &st, const void *pts, bool f32, const int32_t *n_pts, const double *dt,
                  int32_t *assoc, int32_t *db_n, int32_t *db_labels, int UM, int parity, hipStream_t stream)
{
    const int ppt = (cfg.max_pts + kThreads - 1) / kThreads;
    if (f32) {
        if (ppt <= 1) launch_track_t<1, true>(cfg, st, pts, n_pts, dt, assoc, db_n, db_labels, UM, parity, stream);
        else if (ppt == 2) launch_track_t<2, true>(cfg, st, pts, n_pts, dt, assoc, db_n, db_labels, UM, parity, stream);
        else launch_track_t<4, true>(cfg, st, pts, n_pts, dt, assoc, db_n, db_labels, UM, parity, stream);
        return;
    }
    if (ppt <= 1) launch_track_t<1, false>(cfg, st, pts, n_pts, dt, assoc, db_n, db_labels, UM, parity, stream);
    else if (ppt == 2) launch_track_t<2, false>(cfg, st, pts, n_pts, dt, assoc, db_n, db_labels, UM, parity, stream);
    else launch_track_t<4, false>(cfg, st, pts, n_pts, dt, assoc, db_n, db_labels, UM, parity, stream);
}

template <int PPT, bool F32>
static hipError_t prepare_track_t(int lds)
{
    const void *fns[4] = {(const void *)k_track<PPT, false, false, F32>, (const void *)k_track<PPT, true, false, F32>,
                          (const void *)k_track<PPT, false, true, F32>, (const void *)k_track<PPT, true, true, F32>};
    for (const void *f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t prepare_track(const DevCfg &cfg)
{
    const int lds = (int)track_lds_bytes(cfg);
    hipError_t e = prepare_track_t<1, false>(lds);
    if (e == hipSuccess) e = prepare_track_t<2, false>(lds);
    if (e == hipSuccess) e = prepare_track_t<4, false>(lds);
    if (e == hipSuccess) e = prepare_track_t<1, true>(lds);
    if (e == hipSuccess) e = prepare_track_t<2, true>(lds);
    if (e == hipSuccess) e = prepare_track_t<4, true>(lds);
    return e;
}

}  // namespace mmw
