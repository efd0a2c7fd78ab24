// k_scene.hip -- TrackBuffer.track (Tracking.py:664-703) of ONE scene in ONE workgroup, start to finish: _predict_all,
// gating / association, _get_gated_clouds, associate_pointcloud, _maintain_tracks, _update_all, batch.add_frame and the first
// stage of the apply_DBscan screen.  It is the step kernel of contexts SMALL enough that every scene is resident at once
// (two 256-thread workgroups per CU: <= 512 scenes -- one GPU's shard of BASELINE configs[2]/[4] on eight GPUs).  There the
// step is not throughput but ONE scene's chain of dependent work, and the bulk kernels (k_predict / k_track / k_post) pay
// for their generality in exactly that currency: three launches, a schedule indirection, track records that travel through
// global memory between the stages, 16 lanes per Kalman filter, barriers and global round trips between short phases.
// What this kernel does about it:
//   * the records of the scene's first kRes tracks are staged into LDS once (`rec`), predicted there, carry the frame's
//     statistics there, are updated from there; x and P go back to global memory once, after the update;
//   * 32 or 64 lanes per Kalman filter (the products are laid out over matrix elements), so a scene of <= 8 tracks runs
//     all its predictions / updates at once on the four waves;
//   * every first-level load of the frame (header, count, track order, the record slots of the first Kalman round) leaves
//     in one batch, the records follow, and only then the frame's rows (the bulk of the traffic): the chain starts with
//     the Kalman prediction, which must not queue behind 32 KB of points per scene;
//   * the class split is one barrier: block counts by ballot, class offsets and block prefixes by a wave scan;
//   * the sequential column sums of PointCluster (np.mean(axis=0): one dependent fp64 add per row) and the min / max passes
//     of a cloud run on DIFFERENT waves;
//   * no global-memory fence between phases: what a later phase needs of a record is in LDS.
// The arithmetic per number is that of the bulk kernels and of the oracle (mmw_math.hpp, mmw_kalman.hpp): the parity
// tests run every scenario through both.  Not for seek_inner / resized rings (k_track's INNER instantiations) or
// t_cap > 63.  Tracks past the first kRes (a scene may exceed TR_MAX_TRACKS for a frame, Tracking.py:576-589) take
// a slow path through global memory: correct, not fast.
#include "mmw_device.hpp"
#include "mmw_math.hpp"
#include "mmw_cloud.hpp"
#include "mmw_kalman.hpp"
#include "mmw_launch.hpp"

namespace mmw {

// Diagnostic build only (make STAMPS=1): raw clock of lane 0 of every wave of ONE workgroup (block kProbeBlock), and the
// start / end of every workgroup -- scripts/probe_timeline.py, scripts/wg_times.py.  Never compiled into the product library.
#ifdef MMW_STAMPS
#ifndef MMW_PROBE_BLOCK
#define MMW_PROBE_BLOCK 7
#endif
#define PROBE(id)                                                                             \
    do {                                                                                      \
        if (blockIdx.x == MMW_PROBE_BLOCK && (threadIdx.x & 63) == 0)                         \
            st.stats[kStatSlots * kStatWords + (threadIdx.x >> 6) * 64 + (id)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define WGTIME(k)                                                                             \
    do {                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < 2048) {                                          \
            st.stats[kStatSlots * kStatWords + 256 + blockIdx.x * 4 + (k) * 2] = __builtin_amdgcn_s_memrealtime(); \
            st.stats[kStatSlots * kStatWords + 256 + blockIdx.x * 4 + (k) * 2 + 1] = __builtin_amdgcn_s_memtime(); \
        }                                                                                     \
    } while (0)
#else
#define PROBE(id)
#define WGTIME(k)
#endif

namespace scene {

constexpr int kRes = 16;       // tracks whose record prefix lives in LDS for the whole step
constexpr int kTilePad = 2;    // columns of the point tile are NP + 2 doubles apart (k_track.hip)
constexpr int kTrackBytesPerTrack = 352 + 392 + 540;  // as k_track counts them (bench.py prices the Kalman stages itself)

__host__ __device__ inline int split8(int n) { const int h = n / 2; return h - h % 8; }
__host__ __device__ inline int max_leaves(int np) { return np > 128 ? np / 57 + 1 : 0; }
__host__ __device__ inline size_t al16(size_t v) { return (v + 15) & ~(size_t)15; }

struct Lds {
    double *rec;         // [kRes][kRecStage] record prefix of tracks 0..kRes-1, by effective_tracks position at the start of the frame
    double *tmp;         // [4][kRecStage] staging of the tracks past kRes (slow path: wave 0, four at a time)
    double *work;        // three lives: Kalman scratch (predict) | point tile + leaf sums | Kalman scratch (update) | screen grid
    double *cen;         // [t_cap][6]
    double *colsum;      // [t_cap][6]  column sums (sum lanes -> epilogue)
    double *mnmx;        // [t_cap][12] min / max per column (min/max lanes -> epilogue)
    double *nest;        // [t_cap]
    double *life;        // [t_cap] lifetime after associate_pointcloud / update_lifetime
    long long *seg_dst;  // [CLS + 1] where this frame's rows go (in doubles)
    unsigned short *cnt; // [NB][CLS] class counts per 64-point block
    int *cls_n;          // [CLS]
    int *cls_off;        // [CLS + 1]
    int *ml;             // leaves of clouds of more than 128 rows: [0] leaves, [1] clouds, per leaf (track, off, len), per cloud (track, first leaf)
    int *slot;           // [t_cap] effective_tracks position -> physical record, as the frame found it
    int *slot2;          // [t_cap] ... after _maintain_tracks
    int *stat;           // [t_cap] cluster.status after this frame
    int *keep;           // [t_cap] survives _maintain_tracks
    int *misc;           // [16]
};

template <bool WRITE>
__host__ __device__ __forceinline__ size_t lds_layout(const DevCfg &c, char *base, Lds *L)
{
    const int NP = c.max_pts, NB = (NP + 63) / 64, CLS = c.t_cap + 1;
    size_t off = 0;
#define CARVE(field, type, count)                            \
    if constexpr (WRITE) L->field = (type *)(base + off);    \
    off = al16(off + sizeof(type) * (size_t)(count));
    CARVE(rec, double, kRes *kRecStage)
    CARVE(tmp, double, 4 * kRecStage)
    {
        const size_t tile = (size_t)6 * (NP + kTilePad) + (size_t)max_leaves(NP) * 21;
        const size_t kal = (size_t)16 * (kUpdW > kPredW ? kUpdW : kPredW);
        const size_t scr = (4096 + kCloudGrid * 4 + 64) / 8;
        size_t w = tile > kal ? tile : kal;
        w = w > scr ? w : scr;
        CARVE(work, double, w)
    }
    CARVE(cen, double, c.t_cap * 6)
    CARVE(colsum, double, c.t_cap * 6)
    CARVE(mnmx, double, c.t_cap * 12)
    CARVE(nest, double, c.t_cap)
    CARVE(life, double, c.t_cap)
    CARVE(seg_dst, long long, CLS + 1)
    CARVE(cnt, unsigned short, NB *CLS)
    CARVE(cls_n, int, CLS)
    CARVE(cls_off, int, CLS + 1)
    CARVE(ml, int, 2 + 5 * (max_leaves(NP) + 1))
    CARVE(slot, int, c.t_cap)
    CARVE(slot2, int, c.t_cap)
    CARVE(stat, int, c.t_cap)
    CARVE(keep, int, c.t_cap)
    CARVE(misc, int, 16)
#undef CARVE
    return off;
}

// numpy pairwise_sum_DOUBLE (the 1-D np.mean of ClusterTrack._get_D, Tracking.py:286) -- see k_track.hip for the order.
template <int D, typename F>
__device__ __forceinline__ void for_each_leaf(int off, int n, F f)
{
    if constexpr (D == 0) f(off, n);
    else {
        if (n <= 128) f(off, n);
        else { const int n2 = split8(n); for_each_leaf<D - 1>(off, n2, f); for_each_leaf<D - 1>(off + n2, n - n2, f); }
    }
}
template <int D>
__device__ __forceinline__ double combine_leaves(int n, const double *leafsum, int stride, int &idx)
{
    if constexpr (D == 0) { const double v = leafsum[idx * stride]; idx++; return v; }
    else {
        if (n <= 128) { const double v = leafsum[idx * stride]; idx++; return v; }
        const int n2 = split8(n);
        const double l = combine_leaves<D - 1>(n2, leafsum, stride, idx);
        const double r = combine_leaves<D - 1>(n - n2, leafsum, stride, idx);
        return l + r;
    }
}
constexpr int kPwDepth = 4;

// one leaf (n <= 128) of sum_r (pa[r]-ca)*(pb[r]-cb): numpy's eight interleaved accumulators, then the n%8 leftovers
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

// a word of track j's record prefix: from the LDS copy (j < kRes) or from global memory
__device__ __forceinline__ double rec_ld(const Lds &L, const TrackRec *g, int j, int w)
{
    return j < kRes ? L.rec[j * kRecStage + w] : reinterpret_cast<const double *>(g)[w];
}
// ... stored to global memory always (later kernels, the host) and to the LDS copy (this kernel's later phases)
__device__ __forceinline__ void rec_st(const Lds &L, TrackRec *g, int j, int w, double v)
{
    reinterpret_cast<double *>(g)[w] = v;
    if (j < kRes) L.rec[j * kRecStage + w] = v;
}

// _predict_all for the tracks [j0, j0 + 256 / LP) of the scene, one per LP-lane group, on the LDS copies (j < kRes only).
// `between` runs once the record loads have been requested (the caller's point loads: behind the records, not in front).
template <int DX, int LP, typename F>
__device__ __forceinline__ void predict_round(const DevCfg &cfg, const DevState &st, const Lds &L, TrackRec *trk, int s, int j0, int Tres, int slot_reg,
                                              bool use_reg, double dt, int &perr, F between)
{
    const int tid = threadIdx.x, lane = tid & 63, grp = tid / LP, c = tid % LP;
    const int j = j0 + grp;
    const bool live = j < Tres;
    const int sl = live ? (use_reg ? slot_reg : L.slot[j]) : 0;
    double *R = live ? L.rec + (size_t)j * kRecStage : L.tmp + (size_t)(grp & 3) * kRecStage;  // (idle groups: a dead copy)
    double *W = L.work + (size_t)grp * kPredW;
    StageRegs<LP> S;
    stage_record_load<LP>(trk + sl, c, S);
    between();
    stage_record_store<LP>(R, c, S);
    wave_sync();
    predict_math<DX, LP, false>(cfg, nullptr, st.gate_buf + ((size_t)s * cfg.t_cap + (live ? j : 0)) * kGateRec, live, dt, R, W, lane, c, perr);
}
// _update_all for the same groups: tracks that survive _maintain_tracks (L.keep)
template <int DX, int LP>
__device__ __forceinline__ void update_round(const Lds &L, TrackRec *trk, int j0, int Tres, int &uerr)
{
    const int tid = threadIdx.x, lane = tid & 63, grp = tid / LP, c = tid % LP;
    const int j = j0 + grp;
    const bool live = j < Tres && L.keep[j] != 0;
    const double *R = L.rec + (size_t)(j < Tres ? j : 0) * kRecStage;
    double *W = L.work + (size_t)grp * kUpdW;
    update_math<DX, LP>(trk + (live ? L.slot[j] : 0), live, R, W, lane, c, uerr);
}

}  // namespace scene

using namespace scene;

template <int NT, int PPT, int DX, bool F32 = false>
// Register budget: two workgroups per CU (256 VGPRs per wave) for up to 512 points per frame; four points per thread (513..1024
// points) need ~265 and would spill 8-10 of them into scratch memory there: those instantiations take ONE workgroup per CU --
// the unified register file then gives a wave 512 registers and the surplus sits in AGPRs (mmw_api.hip sizes "all scenes
// resident" accordingly: 256 scenes).
__global__ __launch_bounds__(NT, PPT >= 4 ? NT / 256 : NT / 128) void k_scene(DevCfg cfg, DevState st, const void *__restrict__ pts_all, const int32_t *__restrict__ n_pts,
                                                       const double *__restrict__ dt_all, int32_t *__restrict__ assoc_out,
                                                       int32_t *__restrict__ db_n_out, int32_t *__restrict__ db_labels_out, int UM_out, int parity)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    Lds L;
    lds_layout<true>(cfg, lds_raw, &L);
    const int s = blockIdx.x;  // (every scene is resident: the dispatch order is no schedule here, st.perm is not read)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NP = cfg.max_pts, CLS = cfg.t_cap + 1, NPs = NP + kTilePad;
    PROBE(0);
    WGTIME(0);
    SceneHdr *hdr = st.hdr + s;
    int32_t *order = st.order + (size_t)s * cfg.t_cap;
    TrackRec *trk = st.trk + (size_t)s * cfg.t_cap;
    // ---- every first-level load of the frame in one batch: the header, the count, dt, the track order (position `tid` for
    //      the tables, and the position this thread's Kalman group takes in the first round under each of the three
    //      groupings, so that the records can be requested without a barrier in between) ----
    const int n_raw = n_pts[s];
    const SceneHdr hv = *hdr;
    const double dt = dt_all[s];
    const int my_slot = tid < cfg.t_cap ? order[tid] : 0;
    // (the lanes per Kalman filter depend on the track count, which is in flight: the slot for each grouping)
    const int slot64 = (tid >> 6) < cfg.t_cap ? order[tid >> 6] : 0, slot32 = (tid >> 5) < cfg.t_cap ? order[tid >> 5] : 0,
              slot16 = (tid >> 4) < cfg.t_cap ? order[tid >> 4] : 0;
    double2 pr[PPT][4];
    // (the frame's rows: 64 B per thread and point, the bulk of the step's traffic.  They are requested BEHIND the track
    //  records -- the Kalman prediction is the head of the chain and the gate needs the rows only after it --, without
    //  waiting for the point count: rows past it are allocated memory, loaded speculatively and ignored)
    auto load_points = [&]() {
        asm volatile("" : : : "memory");  // (compiler only: keeps these loads behind the ones requested so far)
        const void *frame = frame_of(pts_all, s, NP, F32);   // (fp32 rows, mmw_step_f32: promoted as they are loaded)
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = q * NT + tid;
            load_point_row<F32>(frame, i, i < NP, pr[q]);
        }
    };
    const int n = n_raw < 0 ? 0 : n_raw;  // MMW_EMPTY_FRAME: track() on an empty cloud
    if (tid == 0 && db_n_out) db_n_out[s] = -1;
    if (s == 0 && tid < 4) st.db_count[(parity ^ 1) * 4 + tid] = 0;  // next step's work-list lengths and queue counters (k_post's workers)
    if (s == 0 && tid >= 4 && tid < 7) st.q[(parity ^ 1) * 8 + (tid - 4)] = tid - 4 == kQHead ? q_tag(cfg.epoch + 1) : 0;  // ... and its queue counters (kQCount, kQHead -- tagged with the step it will serve --, kQDone)
    if (s == 0 && tid >= 8 && tid < 11) st.q[kQBig + (parity ^ 1) * 8 + (tid - 8)] = tid - 8 == kQHead ? q_tag(cfg.epoch + 1) : 0;  // ... and those of the large clouds' queue
    if (s == 0 && tid < kUpdWords) st.upd_count[(parity ^ 1) * kUpdWords + tid] = 0;
    if (s == 0 && tid == NT - 1) st.spc_count[parity ^ 1] = 0;
    if (!frame_reaches_track(n_raw, NP)) {  // offline_main.py:56: empty frames never reach track()
        if (tid == 0) {
            hdr->need_db = 0;
            hdr->skipped = (hv.skipped & ~255) | 1;   // (the ring's non-finite flags stay)
            if (n_raw != 0) atomicOr(&hdr->err, ERR_BADCOUNT);
        }
        return;
    }
    int T = hv.n_tracks;
    const int Tin = T, Tres = T < kRes ? T : kRes;
    // lanes per Kalman filter (uniform): every track of the scene in ONE round of the workgroup's waves when it holds <= 16
    const int lp = T <= NT / 64 ? 64 : (T <= NT / 32 ? 32 : 16);
    const int per_round = NT / lp;
    const int grp_slot = lp == 64 ? slot64 : (lp == 32 ? slot32 : slot16);
    int err = 0;
    if (tid < cfg.t_cap) L.slot[tid] = my_slot;
    if (tid == 0) { L.ml[0] = 0; L.ml[1] = 0; L.misc[15] = 0; }
    PROBE(1);

    // ---- _predict_all (Tracking.py:591-596) + the gate matrices, on the LDS copies ----
    if (T > 0) {  // uniform
        int perr = 0;
        for (int j0 = 0; j0 < Tres; j0 += per_round) {
            if (j0 > 0 && j0 == per_round) lds_barrier();  // L.slot for the later rounds
            auto between = [&]() { if (j0 == 0) load_points(); };
            if (lp == 64) predict_round<DX, 64>(cfg, st, L, trk, s, j0, Tres, grp_slot, j0 == 0, dt, perr, between);
            else if (lp == 32) predict_round<DX, 32>(cfg, st, L, trk, s, j0, Tres, grp_slot, j0 == 0, dt, perr, between);
            else predict_round<DX, 16>(cfg, st, L, trk, s, j0, Tres, grp_slot, j0 == 0, dt, perr, between);
        }
        if (T > kRes) {  // the slow path: x and P through global memory, wave 0, four tracks at a time
            lds_barrier();
            if (wave == 0) {
                const int g = lane >> 4, c = lane & 15;
                for (int j0 = kRes; j0 < T; j0 += 4) {
                    const int j = j0 + g;
                    const bool live = j < T;
                    TrackRec *rec = trk + (live ? L.slot[j] : 0);
                    double *R = L.tmp + (size_t)g * kRecStage;
                    stage_record<16>(rec, R, c);
                    wave_sync();
                    predict_math<DX, 16, true>(cfg, rec, st.gate_buf + ((size_t)s * cfg.t_cap + (live ? j : 0)) * kGateRec, live, dt, R,
                                               L.work + (size_t)g * kPredW, lane, c, perr);
                }
            }
        }
        PROBE(2);
        if (perr) atomicOr(&hdr->err, perr);
        // The gate records went to global memory (gate_buf) and come back through the SCALAR cache, which has not seen these
        // lines in this launch but may hold a neighbour scene's share of one: stores acknowledged (vmcnt(0): the vector L1
        // writes through), workgroup fence + barrier, scalar-cache invalidate.  (Tried instead: the records in LDS, read as
        // broadcasts into VGPR operands -- no round trip, but 21 k cycles for eight tracks against 14 k + 5 k this way.)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_s_dcache_inv();
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    } else {
        load_points();
        lds_barrier();
    }

    PROBE(3);
    // ---- gate every point against the scene's tracks (Tracking.py:553-572): see k_track.hip ----
    double bestd[PPT];
    int bestj[PPT];
#pragma unroll
    for (int q = 0; q < PPT; q++) { bestd[q] = 0.0; bestj[q] = -1; }
    {
        // One track's record (C^-1, log det, predicted position: 43 doubles) is the same for every point: read through the
        // scalar cache (constant address space, uniform address -> s_load) it enters the fp64 VALU ops as their SGPR operand.
        // y' C^-1 y as k-ordered FUSED chains, the arithmetic definition the oracle shares (k_track.hip).
#ifdef MMW_DIAG_VGATE   // (diagnostic build, scripts/dual_run.py: the records by VECTOR loads -- volatile global -- instead of through the scalar cache)
        typedef const volatile double *gate_ptr;
#else
        typedef const double __attribute__((address_space(4))) *gate_ptr;
#endif
        const int su = __builtin_amdgcn_readfirstlane(s), Tu = __builtin_amdgcn_readfirstlane(T);
        gate_ptr gb = (gate_ptr)(st.gate_buf + (size_t)su * cfg.t_cap * kGateRec);
        // (the constant address space promises the compiler memory that does not change: the pointer is made opaque HERE,
        //  behind the invalidate, so that no load through it can be moved above this statement)
        asm volatile("; mmw: gate pointer opaque from here" : "+s"(gb) : : "memory");
#ifndef MMW_DIAG_VGATE
        {   // warm the scalar cache: one dword of every 64-byte line of the records, all requests in flight together
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
        }
#endif
        PROBE(4);
        for (int j = 0; j < Tu; j++) {
            gate_ptr G = gb + j * kGateRec;
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = q * NT + tid;
                if (q * NT < n) {  // wave-uniform
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

    PROBE(5);
    // ---- _get_gated_clouds: order-preserving split by class (Tracking.py:605-629), ONE barrier: class counts of every
    //      64-point block by ballot (lane c keeps class c's), one LDS exchange, then every wave forms the class totals, the
    //      class offsets (a scan over lanes) and the prefixes of its own blocks by itself ----
    const int NB = (n + 63) / 64;
    unsigned long long mybal[PPT];
    int cls[PPT];
#pragma unroll
    for (int q = 0; q < PPT; q++) {
        const int i = q * NT + tid, blk = q * (NT / 64) + wave;
        mybal[q] = 0;
        cls[q] = (i < n) ? bestj[q] + 1 : -1;
        if (blk < NB) {  // wave-uniform
            if (i < n && assoc_out) assoc_out[(size_t)s * NP + i] = bestj[q];
            int mine = 0;
            for (int c = 0; c <= T; c++) {
                const unsigned long long b = __ballot(cls[q] == c);
                if (cls[q] == c) mybal[q] = b;
                if (lane == c) mine = __popcll(b);
            }
            if (lane <= T) L.cnt[blk * CLS + lane] = (unsigned short)mine;
        }
    }
    // where this frame's rows go, from the ring state BEFORE this frame's push: a full ring recycles its oldest slot
    // (BatchedData.add_frame, Tracking.py:43-51).  Track `tid`'s ring from its record, the global ring from the header.
    if (tid <= T) {
        int len, rs[MMW_RING_MAX];
        if (tid < T) {
            if (tid < kRes) {
                const int32_t *ri = reinterpret_cast<const int32_t *>(L.rec + (size_t)tid * kRecStage + rInts);
                len = ri[2];
#pragma unroll
                for (int k = 0; k < MMW_RING_MAX; k++) rs[k] = ri[8 + k];
            } else {
                const TrackRec *rec = trk + L.slot[tid];
                len = rec->ring_len;
#pragma unroll
                for (int k = 0; k < MMW_RING_MAX; k++) rs[k] = rec->ring_slot[k];
            }
        } else {
            len = hv.g_len;
#pragma unroll
            for (int k = 0; k < MMW_RING_MAX; k++) rs[k] = hv.g_slot[k];
        }
        int phys = rs[0];
#pragma unroll
        for (int k = 1; k < MMW_RING_MAX; k++) if (len < cfg.ring && k == len) phys = rs[k];
        L.seg_dst[tid] = tid < T ? (((long long)s * cfg.t_cap + L.slot[tid]) * cfg.ring + phys) * (long long)cfg.ring_rows * 8
                                 : ((long long)s * cfg.ring + phys) * (long long)NP * 8;
    }
    PROBE(6);
    lds_barrier();
    PROBE(7);
    int nun = 0;  // unassigned rows of this frame (class 0), uniform
    {
        int tot = 0, pre[PPT];
#pragma unroll
        for (int q = 0; q < PPT; q++) pre[q] = 0;
        if (lane <= T) {
            int v[PPT * (NT / 64)];
#pragma unroll
            for (int b = 0; b < PPT * (NT / 64); b++) v[b] = b < NB ? L.cnt[b * CLS + lane] : 0;
#pragma unroll
            for (int b = 0; b < PPT * (NT / 64); b++) {
#pragma unroll
                for (int q = 0; q < PPT; q++) if (b < q * (NT / 64) + wave) pre[q] += v[b];
                tot += v[b];
            }
        }
        // class offsets and this thread's tile position: a uniform walk over the classes, totals and block prefixes taken
        // from their lanes by v_readlane (no LDS crossbar, no scan)
        int run = 0, local[PPT], pos[PPT];
#pragma unroll
        for (int q = 0; q < PPT; q++) { local[q] = 0; pos[q] = 0; }
        for (int c = 0; c <= T; c++) {
            const int tc = __builtin_amdgcn_readlane(tot, c);
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int pc = __builtin_amdgcn_readlane(pre[q], c);
                if (cls[q] == c) { local[q] = pc; pos[q] = run + pc; }
            }
            if (tid == 0) { L.cls_n[c] = tc; L.cls_off[c] = run; }
            if (c == 0) nun = tc;
            run += tc;
        }
        if (tid == 0) L.cls_off[T + 1] = run;
#pragma unroll
        for (int q = 0; q < PPT; q++) {
            const int i = q * NT + tid;
            const int cc = cls[q] < 0 ? 0 : cls[q];
            const int rank = __popcll(mybal[q] & lanemask_lt());  // rank inside its cloud's share of the block
            const int local_q = local[q] + rank, pos_q = pos[q] + rank;
            if (i < n) {
                // park the point in the LDS tile (SoA, 6 columns) at its CLASS-SORTED position ...
                L.work[0 * NPs + pos_q] = pr[q][0].x; L.work[1 * NPs + pos_q] = pr[q][0].y;
                L.work[2 * NPs + pos_q] = pr[q][1].x; L.work[3 * NPs + pos_q] = pr[q][1].y;
                L.work[4 * NPs + pos_q] = pr[q][2].x; L.work[5 * NPs + pos_q] = pr[q][2].y;
                // ... and its full row (8 columns) goes to the ring it belongs to: track j keeps the first ring_rows rows of
                // its cloud (Tracking.py:341 via BatchedData), the global ring all unassigned rows
                if (cc == 0 || local_q < cfg.ring_rows) {
                    double *dst = (cc == 0 ? st.g_ring + L.seg_dst[T] : st.trk_ring + L.seg_dst[cc - 1]) + (size_t)local_q * 8;
                    double2 *d2 = reinterpret_cast<double2 *>(dst);
                    d2[0] = pr[q][0]; d2[1] = pr[q][1]; d2[2] = pr[q][2]; d2[3] = pr[q][3];
                }
                // a NaN / an infinite value in a row that enters the global ring (all 8 columns): see the trigger (k_track.hip)
                if (cc == 0) {
                    const int nfb = row_nonfinite_bits(pr[q]);
                    if (nfb) atomicOr(&L.misc[15], nfb);
                }
            }
        }
    }
    PROBE(8);
    const double *p6 = L.work;
    // ---- batch.add_frame(unassigned) on the global ring (Tracking.py:689-691), from the header as it was loaded: uniform
    //      arithmetic, every thread does it (no LDS), thread 0 stores ----
    int g_len = hv.g_len, gs[MMW_RING_MAX], gn[MMW_RING_MAX], U = 0;
    {
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) { gs[k] = hv.g_slot[k]; gn[k] = hv.g_n[k]; }
        if (g_len >= cfg.ring && g_len > 0) {  // pop_frame (a ring of fixed size pops once): the freed physical slot becomes the first free entry
            const int first = gs[0];
#pragma unroll
            for (int k = 1; k < MMW_RING_MAX; k++) if (k < g_len) { gs[k - 1] = gs[k]; gn[k - 1] = gn[k]; }
#pragma unroll
            for (int k = 0; k < MMW_RING_MAX; k++) if (k == g_len - 1) gs[k] = first;
            g_len--;
        }
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) if (k == g_len) gn[k] = nun;
        g_len++;
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) { if (k >= g_len) gn[k] = 0; U += gn[k]; }
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < MMW_RING_MAX; k++) { hdr->g_slot[k] = gs[k]; hdr->g_n[k] = gn[k]; }
            hdr->g_len = g_len;
            hdr->db_u = U;
        }
    }
    lds_barrier();  // the tile, cls_n, cls_off
    PROBE(9);

    // ---- associate_pointcloud (Tracking.py:314-341): PointCluster stats.  The column sums are sequential in row order
    //      (np.mean(axis=0)): one dependent add per row, six lanes per track; tracks j = wave (mod 4) on this wave.  The
    //      min / max passes (four contiguous slices per column, combined in order with the sequential rule "a later value
    //      wins only if strictly smaller / larger") of tracks j = wave + 2 (mod 4): a lone large cloud has its sums on one
    //      wave and its min / max on another ----
    {
        // (eight waves: the sums on waves 0..3, the min / max passes on waves 4..7; four waves: both on every wave)
        const bool sum_wave = NT == 256 || wave < 4, mm_wave = NT == 256 || wave >= 4;
        const int grp = lane / 6, m = lane - grp * 6;
        for (int jb = 0; sum_wave && jb < T; jb += 40) {
            const int j = jb + grp * 4 + (wave & 3);
            const bool valid = lane < 60 && j < T;
            const int nj = valid ? L.cls_n[j + 1] : 0, off = valid ? L.cls_off[j + 1] : 0;
            if (nj > 0) {
                const double *col = p6 + m * NPs + off;
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
                L.colsum[j * 6 + m] = sum;
            }
        }
        PROBE(10);
        // min / max: 24 lanes per track (six columns x four slices), two tracks per pass: tracks j = wave + 2 (mod 4)
        const int half = lane >= 32 ? 1 : 0, l5 = lane & 31, pi = l5 >> 2, slice = l5 & 3;  // (lanes 24..31 of a half idle)
        for (int jb = NT == 256 ? (wave + 2) & 3 : wave - 4; mm_wave && jb < T; jb += 8) {
            const int j = jb + 4 * half;
            const bool valid = j < T && pi < 6;
            const int nj = valid ? L.cls_n[j + 1] : 0, off = valid ? L.cls_off[j + 1] : 0;
            double mn = __longlong_as_double(0x7ff0000000000000LL), mx = -mn;  // empty slice: never wins
            if (nj > 0) {
                const double *col = p6 + pi * NPs + off;
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
#pragma unroll
            for (int d = 1; d <= 2; d <<= 1) {  // slices (0,1),(2,3), then halves: the partner holds the LATER rows
                const double tn = __shfl_down(mn, d), tx = __shfl_down(mx, d);
                mn = tn < mn ? tn : mn;
                mx = tx > mx ? tx : mx;
            }
            if (nj > 0 && slice == 0) { L.mnmx[j * 12 + pi * 2] = mn; L.mnmx[j * 12 + pi * 2 + 1] = mx; }
        }
    }
    PROBE(11);
    // this thread's point of the DBSCAN cloud (the global ring, oldest frame first), for the screen at the end: older
    // frames from global memory (requested now), this frame's unassigned rows from the tile, where they are the first run
    double sx = 0.0, sy = 0.0, sz = 0.0;
    if (U <= 256 && tid < U) {
        const int old_n = U - nun;  // rows of the older frames
        if (tid >= old_n) {
            const int k = tid - old_n;
            sx = p6[0 * NPs + k]; sy = p6[1 * NPs + k]; sz = p6[2 * NPs + k];
        } else {
            int f = 0, base = 0;  // frame of row `tid`
#pragma unroll
            for (int k = 0; k < MMW_RING_MAX - 1; k++) {
                int gnf = gn[0];
#pragma unroll
                for (int k2 = 1; k2 < MMW_RING_MAX; k2++) if (k2 == f) gnf = gn[k2];
                if (tid >= base + gnf && f < g_len - 2) { base += gnf; f++; }
            }
            int gsf = gs[0];
#pragma unroll
            for (int k2 = 1; k2 < MMW_RING_MAX; k2++) if (k2 == f) gsf = gs[k2];
            const double *r = st.g_ring + ((size_t)s * cfg.ring + gsf) * (size_t)NP * 8 + (size_t)(tid - base) * 8;
            const double2 a = *reinterpret_cast<const double2 *>(r);
            sx = a.x; sy = a.y; sz = r[2];
        }
    }
    lds_barrier();
    PROBE(12);
    // ---- ... N_est, spread_est (Tracking.py:232-268), update_lifetime (400-407): a thread per (track, column) ----
    for (int it = tid; it < T * 6; it += NT) {
        const int j = it / 6, m = it - j * 6;
        const int nj = L.cls_n[j + 1];
        TrackRec *rec = trk + L.slot[j];
        if (nj == 0) {
            if (m == 0) {
                const double lf = rec_ld(L, rec, j, rLife) + dt;  // update_lifetime(dt)
                rec_st(L, rec, j, rLife, lf);
                L.life[j] = lf;
                L.stat[j] = j < kRes ? reinterpret_cast<const int32_t *>(L.rec + (size_t)j * kRecStage + rInts)[1] : rec->is_static;
            }
            continue;
        }
        const double old = rec_ld(L, rec, j, rSpr + m);
        const double mn = L.mnmx[j * 12 + m * 2], mx = L.mnmx[j * 12 + m * 2 + 1];
        const double cen = L.colsum[j * 6 + m] / (double)nj;
        L.cen[j * 6 + m] = cen;
        rec_st(L, rec, j, rCen + m, cen);
        rec->minv[m] = mn;
        rec->maxv[m] = mx;
        // _estimate_measurement_spread Tracking.py:246-268
        double spread = mx - mn;
        const double lim = cfg.kf_spread_lim[m], lim2 = 2 * lim;
        if (nj != 1) spread = spread * (double)(nj + 1) / (double)(nj - 1);
        spread = spread < lim2 ? spread : lim2;
        spread = spread > lim ? spread : lim;
        rec_st(L, rec, j, rSpr + m, spread > old ? spread : (1.0 - cfg.kf_a_spr) * old + cfg.kf_a_spr * spread);
        if (m == 0) {
            if (nj > 128) {  // leaves of this cloud's pairwise sums, for the dispersion phase below
                int cnt = 0;
                for_each_leaf<kPwDepth>(0, nj, [&](int, int) { cnt++; });
                const int first = atomicAdd(&L.ml[0], cnt), c = atomicAdd(&L.ml[1], 1);
                int *lf = L.ml + 2 + first * 3, *cl = L.ml + 2 + 3 * max_leaves(NP);
                cl[c * 2] = j; cl[c * 2 + 1] = first;
                int k = 0;
                for_each_leaf<kPwDepth>(0, nj, [&](int o, int len) { lf[k * 3] = j; lf[k * 3 + 1] = o; lf[k * 3 + 2] = len; k++; });
            }
            rec_st(L, rec, j, rLife, 0.0);
            L.life[j] = 0.0;
            rec->point_num = nj;
            if (j < kRes) reinterpret_cast<int32_t *>(L.rec + (size_t)j * kRecStage + rInts)[0] = nj;
            // _estimate_point_num Tracking.py:232-244
            double ne = rec_ld(L, rec, j, rNest);
            if (cfg.kf_enable_est) ne = ((double)nj > ne) ? (double)nj : (1 - cfg.kf_a_n) * ne + cfg.kf_a_n * (double)nj;
            else ne = cfg.kf_est_pointnum > (double)nj ? cfg.kf_est_pointnum : (double)nj;
            rec_st(L, rec, j, rNest, ne);
            L.nest[j] = ne;
        }
    }
    PROBE(13);
    lds_barrier();
    PROBE(14);
    // status: sqrt(sum(centroid[3:6]^2)) < TR_VEL_THRES (Tracking.py:132-136) and BatchedData.add_frame on the track ring
    // (Tracking.py:43-51; the rows were written above): the LAST wave's threads, a track each -- the dispersion items below
    // fill the waves from the front
    for (int j = NT - 1 - tid; j < T; j += NT) {
        const int nj = L.cls_n[j + 1];
        if (nj > 0) {
            TrackRec *rec = trk + L.slot[j];
            const double v3 = L.cen[j * 6 + 3], v4 = L.cen[j * 6 + 4], v5 = L.cen[j * 6 + 5];
            const int stc = sqrt((v3 * v3 + v4 * v4) + v5 * v5) < cfg.tr_vel_thres ? 1 : 0;
            rec->is_static = stc;
            L.stat[j] = stc;
            int len, rn[MMW_RING_MAX], rs[MMW_RING_MAX];
            if (j < kRes) {
                const int32_t *ri = reinterpret_cast<const int32_t *>(L.rec + (size_t)j * kRecStage + rInts);
                len = ri[2];
#pragma unroll
                for (int k = 0; k < MMW_RING_MAX; k++) { rn[k] = ri[4 + k]; rs[k] = ri[8 + k]; }
            } else {
                len = rec->ring_len;
#pragma unroll
                for (int k = 0; k < MMW_RING_MAX; k++) { rn[k] = rec->ring_n[k]; rs[k] = rec->ring_slot[k]; }
            }
            while (len >= cfg.ring && len > 0) {  // pop_frame: the freed physical slot becomes the first free entry
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
        }
    }
    PROBE(15);
    // _estimate_group_disp_matrix + _get_D (Tracking.py:270-297): 21 symmetric entries per track, each the 1-D np.mean of
    // n products in numpy's pairwise order.  Work items: (track, entry) for clouds of one leaf (n <= 128), (leaf, entry)
    // for the leaves of larger clouds, whose sums meet in LDS and are combined by one lane per (track, entry) after a
    // barrier.  (Eight lanes per item -- one per interleaved accumulator -- was tried: with 168 items of 57 rows the
    // per-item overhead outweighs the shorter loops, 19 k against 7 k cycles.)
    {
        double *leafsum = L.work + 6 * NPs;  // [leaf][21]
        const int nleaf = L.ml[0], ncloud = L.ml[1];
        const int *lf = L.ml + 2, *cl = L.ml + 2 + 3 * max_leaves(NP);
        auto entry = [](int e, int &a, int &b) { a = 0; while (e >= 6 - a) { e -= 6 - a; a++; } b = a + e; };
        auto blend = [&](TrackRec *rec, int j, int a, int b, double res, int nj, double g_ab, double g_ba, double ne) {
            const double D = res / (double)nj;
            if (ne == 0.0) { err |= ERR_DIVZERO; return; }
            const double al = (double)nj / ne;
            rec_st(L, rec, j, rGd + a * 6 + b, (1 - al) * g_ab + al * D);
            if (a != b) rec_st(L, rec, j, rGd + b * 6 + a, (1 - al) * g_ba + al * D);
        };
        for (int it = tid; it < (T + nleaf) * 21; it += NT) {
            const int u = it / 21;
            int a, b;
            entry(it - u * 21, a, b);
            const bool direct = u < T;
            const int j = direct ? u : lf[(u - T) * 3];
            const int nj = L.cls_n[j + 1];
            if (direct && (nj == 0 || nj > 128)) continue;
            const int off = L.cls_off[j + 1] + (direct ? 0 : lf[(u - T) * 3 + 1]), len = direct ? nj : lf[(u - T) * 3 + 2];
            TrackRec *rec = trk + L.slot[j];
            double g_ab = 0.0, g_ba = 0.0, ne = 1.0;
            if (direct) { g_ab = rec_ld(L, rec, j, rGd + a * 6 + b); g_ba = rec_ld(L, rec, j, rGd + b * 6 + a); ne = L.nest[j]; }
            const double res = pw_leaf(p6 + a * NPs + off, p6 + b * NPs + off, L.cen[j * 6 + a], L.cen[j * 6 + b], len);
            if (direct) blend(rec, j, a, b, res, nj, g_ab, g_ba, ne);
            else leafsum[(u - T) * 21 + (it - u * 21)] = res;
        }
        if (ncloud > 0) {  // uniform
            lds_barrier();
            for (int it = tid; it < ncloud * 21; it += NT) {
                const int c = it / 21, e = it - c * 21;
                int a, b;
                entry(e, a, b);
                const int j = cl[c * 2], nj = L.cls_n[j + 1];
                TrackRec *rec = trk + L.slot[j];
                const double g_ab = rec_ld(L, rec, j, rGd + a * 6 + b), g_ba = rec_ld(L, rec, j, rGd + b * 6 + a), ne = L.nest[j];
                int idx = 0;
                const double res = combine_leaves<kPwDepth>(nj, leafsum + cl[c * 2 + 1] * 21 + e, 21, idx);
                blend(rec, j, a, b, res, nj, g_ab, g_ba, ne);
            }
        }
    }
    PROBE(16);
    lds_barrier();  // status, lifetimes and the dispersion matrices are in LDS
    PROBE(17);

    // ---- _maintain_tracks (Tracking.py:513-528): wave 0, a lane per track (t_cap <= 63) ----
    if (wave == 0) {
        const bool have = lane < T;
        bool keep = false;
        int sl = 0;
        if (have) {
            sl = L.slot[lane];
            const double lim = L.stat[lane] ? cfg.tr_lifetime_static : cfg.tr_lifetime_dynamic;
            keep = !(L.life[lane] > lim);
            L.keep[lane] = keep ? 1 : 0;
        }
        const unsigned long long kb = __ballot(have && keep), db = __ballot(have && !keep);
        const int nk = __popcll(kb);
        if (have) {
            const int pos = keep ? __popcll(kb & lanemask_lt()) : nk + __popcll(db & lanemask_lt());
            L.slot2[pos] = sl;
            order[pos] = sl;
        }
        if (lane == 0) L.misc[0] = nk;
    }
    if (Tin > kRes) __syncthreads();  // slow path: the tracks past kRes are updated from global memory (other threads' stores)
    else lds_barrier();
    T = L.misc[0];
    PROBE(18);

    // ---- _update_all (Tracking.py:598-603) for the tracks that are left, from the LDS copies; x and P to global memory ----
    if (Tin > 0) {  // uniform
        int uerr = 0;
        for (int j0 = 0; j0 < Tres; j0 += per_round) {
            if (lp == 64) update_round<DX, 64>(L, trk, j0, Tres, uerr);
            else if (lp == 32) update_round<DX, 32>(L, trk, j0, Tres, uerr);
            else update_round<DX, 16>(L, trk, j0, Tres, uerr);
        }
        if (Tin > kRes && wave == 0) {
            const int g = lane >> 4, c = lane & 15;
            for (int j0 = kRes; j0 < Tin; j0 += 4) {
                const int j = j0 + g;
                const bool live = j < Tin && L.keep[j] != 0;
                TrackRec *rec = trk + (j < Tin ? L.slot[j] : 0);
                double *R = L.tmp + (size_t)g * kRecStage;
                stage_record<16>(rec, R, c);
                wave_sync();
                update_math<DX, 16>(rec, live, R, L.work + (size_t)g * kUpdW, lane, c, uerr);
            }
        }
        if (uerr) atomicOr(&hdr->err, uerr);
    }
    PROBE(19);
    // ---- DBSCAN trigger (Tracking.py:693-697) ----
    bool need = U > 0 && T < cfg.tr_max_tracks;
    {
        // the ring's non-finite flags (two bits per physical slot, SceneHdr.skipped): this frame's replace those of the slot it
        // was written to.  apply_DBscan reached with a NaN / an infinite value in the ring: sklearn's input validation raises
        // ValueError (Utils.py:272-278) and track() ends here -- frame in the ring, nothing clustered, nothing cleared (k_track.hip)
        int phys = gs[0], live = 0;
#pragma unroll
        for (int k = 1; k < MMW_RING_MAX; k++) if (k == g_len - 1) phys = gs[k];
        const int nff = nf_flags_with((hv.skipped >> kSkipNfShift) & kSkipNfMask, phys, L.misc[15]);
#pragma unroll
        for (int k = 0; k < MMW_RING_MAX; k++) if (k < g_len) live |= (nff >> (2 * gs[k])) & 3;
        const int nfe = nf_error_of(live);
        const bool raised = need && nfe != 0;
        if (raised) need = false;
        if (tid == 0) {
            hdr->n_tracks = T;
            hdr->n_upd = T;
            hdr->skipped = nff << kSkipNfShift;
            hdr->need_db = need ? 1 : 0;
            if (raised) {
                err |= nfe;
                if (db_n_out) db_n_out[s] = kDbRaised;
            }
        }
    }
    if (err) atomicOr(&hdr->err, err);
    lds_barrier();  // the Kalman scratch is dead: the screen's grid takes its place
    PROBE(20);
    int *grid = reinterpret_cast<int *>(reinterpret_cast<char *>(L.work) + 4096);
    unsigned long long *mm = reinterpret_cast<unsigned long long *>(grid + kCloudGrid);
    if (need) {  // uniform
        for (int i = tid; i < kCloudGrid; i += NT) grid[i] = 0;
        if (tid == 0) { grid[kCloudGrid] = 0; grid[kCloudGrid + 1] = 0; L.misc[12] = 0; }  // (mm[0] behind the grid)
        lds_barrier();
        // ---- apply_DBscan, first stage (Tracking.py:697, Utils.py:250-291): see k_track.hip ----
        bool listed = true;
        if (U <= 256) {
            listed = !cloud_cells_prove_no_core(cfg, U, sx, sy, sz, mm, &L.misc[12], grid);
        } else {  // large clouds (no tracks yet, or lost): rows from the global ring -- this frame's were stored by other threads
            __syncthreads();
            const int big = 0x7fffffff;
            RowSrc src;
            src.gb = st.g_ring + (size_t)s * cfg.ring * (size_t)NP * 8;
            src.stride = (size_t)NP * 8;
            src.slots = (unsigned)gs[0] | ((unsigned)gs[1] << 8) | ((unsigned)gs[2] << 16) | ((unsigned)gs[3] << 24);
            src.c1 = g_len > 1 ? gn[0] : big;
            src.c2 = g_len > 2 ? gn[0] + gn[1] : big;
            src.c3 = g_len > 3 ? gn[0] + gn[1] + gn[2] : big;
            listed = !cloud_cells_prove_no_core_rows<NT>(cfg, src, U, mm, &L.misc[12], grid);
        }
        if (listed && U <= 256) {
            // second stage, the exact pair count (k_post does it for the bulk kernels): this thread still holds its point of
            // the cloud, and a scene that ends here needs no worker block behind this launch
            float4 *P4 = reinterpret_cast<float4 *>(L.work);                        // [256]
            int *pcnt = reinterpret_cast<int *>(reinterpret_cast<char *>(L.work) + 12288);  // [256], behind the grid
            unsigned long long *mm2 = reinterpret_cast<unsigned long long *>(pcnt + 256);   // [3]
            listed = !cloud_pairs_prove_no_core_xyz<NT>(cfg, U, sx, sy, sz, P4, pcnt, mm2, &L.misc[11]);
        }
        if (listed) {
            // work list 3 = clouds <= 256 points (BallTree + _add_tracks), queue 1 = the larger ones: k_post's worker blocks
            // take both after this launch (no side-stream workers beside a context this small)
            const int cl3 = U <= 256 ? 3 : (U <= kBigCloudMax ? 1 : 2);  // (list 2: k_dbscan_huge, contexts whose rings hold more than the LDS classes)
            if (tid == 0) {
                int32_t *cnt = cl3 >= 2 ? st.db_count + parity * 4 + cl3 : st.q + kQBig + parity * 8 + kQCount;
                const int pos = atomicAdd(cnt, 1);
                int32_t *e = st.db_list + (size_t)cl3 * cfg.n_scenes + pos;
                if (cl3 >= 2) *e = s;
                else __hip_atomic_store(e, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            cloud_finish_empty(st, hdr, s, U, UM_out, db_labels_out, db_n_out);
        }
    }
    PROBE(21);
    if (tid == 0 && st.stats) {
        int ring_rows = 0;
        for (int j = 0; j < Tin; j++) ring_rows += min(L.cls_n[j + 1], cfg.ring_rows);
        unsigned long long *sl = stats_slot(st, s);
        atomicAdd(&sl[0], (unsigned long long)((F32 ? 32 : 64) * n + 4 * n + Tin * kTrackBytesPerTrack + 64 * nun + 64 * ring_rows));
        atomicAdd(&sl[2], 1ULL);
        atomicAdd(&sl[5], (unsigned long long)Tin);
        atomicAdd(&sl[6], (unsigned long long)n * (unsigned long long)Tin);
    }
    PROBE(22);
    WGTIME(1);
}

size_t scene_lds_bytes(const DevCfg &c) { return lds_layout<false>(c, nullptr, nullptr); }

template <int NT, int PPT, bool F32>
static void launch_scene_t(const DevCfg &cfg, const DevState &st, const void *pts, const int32_t *n_pts, const double *dt, int32_t *assoc,
                           int32_t *db_n, int32_t *db_labels, int UM, int parity, hipStream_t stream)
{
    if (cfg.dx == 9)
        mmw_launch(k_scene<NT, PPT, 9, F32>, dim3(cfg.n_scenes), dim3(NT), scene_lds_bytes(cfg), stream, cfg, st, pts, n_pts, dt, assoc, db_n, db_labels,
                   UM, parity);
    else
        mmw_launch(k_scene<NT, PPT, 6, F32>, dim3(cfg.n_scenes), dim3(NT), scene_lds_bytes(cfg), stream, cfg, st, pts, n_pts, dt, assoc, db_n, db_labels,
                   UM, parity);
}

// (NT = 256: eight waves per scene were tried -- one point per thread, a wave per Kalman filter, sums and min / max on different
//  waves.  Two such workgroups per CU leave 128 VGPRs per lane: the kernel spills 80 of them and ran 57 us instead of 45.)
void launch_scene(const DevCfg &cfg, const DevState &st, const void *pts, bool f32, const int32_t *n_pts, const double *dt, int32_t *assoc,
                  int32_t *db_n, int32_t *db_labels, int UM, int parity, hipStream_t stream)
{
    const int ppt = (cfg.max_pts + 255) / 256;
    if (f32) {
        if (ppt <= 1) launch_scene_t<256, 1, true>(cfg, st, pts, n_pts, dt, assoc, db_n, db_labels, UM, parity, stream);
        else if (ppt == 2) launch_scene_t<256, 2, true>(cfg, st, pts, n_pts, dt, assoc, db_n, db_labels, UM, parity, stream);
        else launch_scene_t<256, 4, true>(cfg, st, pts, n_pts, dt, assoc, db_n, db_labels, UM, parity, stream);
        return;
    }
    if (ppt <= 1) launch_scene_t<256, 1, false>(cfg, st, pts, n_pts, dt, assoc, db_n, db_labels, UM, parity, stream);
    else if (ppt == 2) launch_scene_t<256, 2, false>(cfg, st, pts, n_pts, dt, assoc, db_n, db_labels, UM, parity, stream);
    else launch_scene_t<256, 4, false>(cfg, st, pts, n_pts, dt, assoc, db_n, db_labels, UM, parity, stream);
}

hipError_t prepare_scene(const DevCfg &cfg)
{
    const int lds = (int)scene_lds_bytes(cfg);
    const void *fns[12] = {(const void *)k_scene<256, 1, 9>, (const void *)k_scene<256, 2, 9>, (const void *)k_scene<256, 4, 9>,
                           (const void *)k_scene<256, 1, 6>, (const void *)k_scene<256, 2, 6>, (const void *)k_scene<256, 4, 6>,
                           (const void *)k_scene<256, 1, 9, true>, (const void *)k_scene<256, 2, 9, true>, (const void *)k_scene<256, 4, 9, true>,
                           (const void *)k_scene<256, 1, 6, true>, (const void *)k_scene<256, 2, 6, true>, (const void *)k_scene<256, 4, 6, true>};
    for (const void *f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace mmw
