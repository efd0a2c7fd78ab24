// k_kalman.hip -- the per-track Kalman work of TrackBuffer.track (Tracking.py:683-703), batched over
// every (scene, track) pair of the context instead of being walked scene by scene:
//
//   k_predict  _predict_all (Tracking.py:591-596; filterpy predict, motion model constants.py:195-215)
//              + the gate matrix of _calc_dist_fun (Tracking.py:549-560): inverse and log|det| of
//              P[:6,:6] + diag((spread/2)^2) + group_disp_est, handed to k_track through `gate_buf`.
//   update_tracks_wave (mmw_kalman.hpp; launched as part of k_post, k_dbscan.hip)
//              _update_all (Tracking.py:598-603): update_state 387-398, _get_Rc 299-312, filterpy's
//              Joseph-form update.
//
// Why separate kernels: inside the one-workgroup-per-scene kernel these were latency chains (a global
// round trip, three or four short matrix stages, a 6x6 LU of ~4000 cycles) during which most of the
// workgroup's waves -- and, at 4 workgroups per CU, most of the chip -- waited at barriers.  Here a
// 16-lane group owns one track, a 64-thread workgroup owns four, there are no block barriers, the LDS
// footprint is a few KiB per wave, and 20+ waves per CU hide each other's latencies.
//
// The arithmetic per matrix element is unchanged (same operations in the same order as the CPU oracle).
#include "mmw_kalman.hpp"
#include "mmw_launch.hpp"

namespace mmw {

// Which (scene, track) a 16-lane group of wave `unit` takes when the work is laid out over the lists "scenes by track
// count, most tracks first" (st.upd_list / st.upd_count of `parity`): entry k = 4 * unit + group is track k % t of
// the (k / t)-th scene of its bin.  Returns false for a group past the end.  Needs t_cap <= 63 (one lane per bin).
struct DenseBins {
    int incl, excl, total;
};
__device__ __forceinline__ DenseBins dense_bins(const DevCfg &cfg, const DevState &st, int parity, int lane)
{
    const int nb = cfg.t_cap, t_of_lane = nb - lane;
    const int32_t *cnt = st.upd_count + (size_t)parity * (cfg.t_cap + 1);
    DenseBins B;
    B.incl = (lane < nb) ? t_of_lane * cnt[t_of_lane] : 0;
    const int mine = B.incl;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(B.incl, o); if (lane >= o) B.incl += v; }
    B.total = __shfl(B.incl, 63);
    B.excl = B.incl - mine;
    return B;
}
__device__ __forceinline__ bool dense_pick(const DevCfg &cfg, const DevState &st, int parity, const DenseBins &B, int unit, int lane, int g,
                                           int &my_s, int &my_j)
{
    // branch-free up to ONE load per lane (four uniform loads behind four branches were four round trips in a row): every lane
    // ends up with a valid address -- idle groups entry 0 of the fullest bin, and scene 0 / track 0 as their answer
    const int nb = cfg.t_cap;
    int my_t = nb, my_r = 0;
    bool live = false;
    my_j = 0;
#pragma unroll
    for (int gg = 0; gg < 4; gg++) {
        const int k = unit * 4 + gg;  // uniform
        const unsigned long long hit = __ballot(lane < nb && B.incl > k);
        const bool valid = k < B.total && hit != 0;
        const int b = valid ? __ffsll((long long)hit) - 1 : 0;  // first bin whose inclusive count exceeds k
        const int base = __shfl(B.excl, b), t = nb - b;
        const int rel = valid ? k - base : 0, r = rel / t, j = rel - r * t;
        if (g == gg) { my_t = t; my_r = r; my_j = j; live = valid; }
    }
    const int sc = st.upd_list[((size_t)parity * (cfg.t_cap + 1) + my_t) * cfg.n_scenes + my_r];
    my_s = live ? sc : 0;
    return live;
}

// k_predict, 64-thread workgroups = one wave = four tracks.  The tracks of the context come from three places:
//   * units [0, n_dense): the scenes tracked in the PREVIOUS frame, by their update lists (tracks 0 .. n_upd-1),
//     four real tracks per wave whatever the scenes hold;
//   * the last kSpecialUnits units: the scenes that spawned tracks in the previous frame (st.spc_list: their new tracks
//     n_upd .. n_tracks-1) and the scenes whose previous frame was empty (hdr->skipped: all their tracks), one per wave;
//   * contexts with t_cap > 63 or too few tracks to fill the chip (tracks_dense, mmw_kalman.hpp): the per-scene layout, wave q of a scene's nq takes tracks 4q.., 4(q+nq)..
// An empty frame (n_pts <= 0) predicts nothing (offline_main.py:56: such frames never reach track()).
constexpr int kSpecialUnits = 64;
// Three waves per SIMD (134 VGPRs, nothing spilled).  Rounds 1-4 capped the kernel at 128 VGPRs for a fourth wave, at the price of
// 2 spilled VGPRs and a scratch-enabled dispatch of 8256 workgroups; same box, alternating (scripts/ab_libs.sh, NOTEBOOK round 5):
// 4096 scenes 20.4-21.0 us capped against 20.1-20.4, 1024 scenes 11.8 against 11.1.
#ifndef MMW_PRED_OCC   // (diagnostic builds: another register budget)
#define MMW_PRED_OCC 3
#endif
template <int DX>
__global__ __launch_bounds__(64, MMW_PRED_OCC) void k_predict(DevCfg cfg, DevState st, const int32_t *__restrict__ n_pts,
                                                const double *__restrict__ dt_all, int nq, int parity)
{
    __shared__ double lds[4 * kPredScratch];
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    double *Wj = lds + g * kPredScratch;
    int err = 0;
    if (tracks_dense(cfg, nq)) {
        const int prev = parity ^ 1, n_dense = cfg.n_scenes * nq;
        // (bin 0 of the counts = the total: the dense work takes the first ceil(total / 4) units, every other unit of the launch
        //  -- the kSpecialUnits behind the dense range always, and the idle rest of the dense range: after a frame in which
        //  EVERY scene spawned tracks, the first frames after a reset, that is all of them; 64 waves alone took 555 us for the
        //  4096 scenes of the start-up -- serves the two lists below)
        // (the dependent round trips of a wave -- total, bin counts, list entry, scene words, slot, record -- are what this launch
        //  lasts, at 20 waves per CU: the bin counts are requested with the total, the scene's four words together: 6 -> 4)
        const int total = st.upd_count[(size_t)prev * (cfg.t_cap + 1)];
        const DenseBins B = dense_bins(cfg, st, prev, lane);
        int nd = (total + 3) >> 2;
        nd = nd < n_dense ? nd : n_dense;
        if ((int)blockIdx.x < nd) {
            for (int unit = blockIdx.x; unit * 4 < B.total; unit += n_dense) {
                int s, j;
                bool live = dense_pick(cfg, st, prev, B, unit, lane, g, s, j);   // (idle groups: s = j = 0)
                const int n = n_pts[s], nt = st.hdr[s].n_tracks, slot = st.order[(size_t)s * cfg.t_cap + j];
                const double dt = dt_all[s];
                // (j < n_tracks: a scene that mmw_reset_scenes has emptied since the lists were built holds no track -- its
                //  stale records must not be predicted, nor their error bits come back on the fresh scene)
                live = live && frame_reaches_track(n, cfg.max_pts) && j < nt;
                if (!__any(live)) continue;
                TrackRec *rec = st.trk + (size_t)s * cfg.t_cap + (live ? slot : 0);
                int e1 = 0;
                predict_one_track<DX>(cfg, st, rec, live, s, j, dt, Wj, lane, c, e1);
                if (e1 && live) atomicOr(&st.hdr[s].err, e1);
            }
            return;
        }
        const int pool = (int)gridDim.x - nd, me = (int)blockIdx.x - nd;
        const int count = st.spc_count[prev];
        for (int i = me; i < count; i += pool) {
            const int s = st.spc_list[((size_t)prev * cfg.n_scenes + i) * 2], first = st.spc_list[((size_t)prev * cfg.n_scenes + i) * 2 + 1];
            const int n = n_pts[s];
            if (!frame_reaches_track(n, cfg.max_pts)) continue;
            const int T = st.hdr[s].n_tracks;
            const int32_t *order = st.order + (size_t)s * cfg.t_cap;
            TrackRec *trk = st.trk + (size_t)s * cfg.t_cap;
            for (int j0 = first; j0 < T; j0 += 4) {
                const int j = j0 + g;
                const bool live = j < T;
                predict_one_track<DX>(cfg, st, trk + (live ? order[j] : 0), live, s, j, dt_all[s], Wj, lane, c, err);
            }
            if (err) { atomicOr(&st.hdr[s].err, err); err = 0; }
        }
        // scenes whose last frame was empty (no update list holds them): all their tracks.  64 scenes per look (one
        // per lane), the rare hits one after the other
        for (int s0 = me * 64; s0 < cfg.n_scenes; s0 += pool * 64) {
            const int sl = s0 + lane;
            bool mine = false;
            if (sl < cfg.n_scenes) {
                const int n = n_pts[sl];
                mine = (st.hdr[sl].skipped & 1) != 0 && st.hdr[sl].n_tracks > 0 && frame_reaches_track(n, cfg.max_pts);
            }
            unsigned long long todo = __ballot(mine);
            while (todo) {
                const int s = s0 + __ffsll((long long)todo) - 1;
                todo &= todo - 1ULL;
                const int T = st.hdr[s].n_tracks;
                const int32_t *order = st.order + (size_t)s * cfg.t_cap;
                TrackRec *trk = st.trk + (size_t)s * cfg.t_cap;
                for (int j0 = 0; j0 < T; j0 += 4) {
                    const int j = j0 + g;
                    const bool live = j < T;
                    predict_one_track<DX>(cfg, st, trk + (live ? order[j] : 0), live, s, j, dt_all[s], Wj, lane, c, err);
                }
                if (err) { atomicOr(&st.hdr[s].err, err); err = 0; }
            }
        }
        return;
    }
    if ((int)blockIdx.x >= cfg.n_scenes * nq) return;
    const int us = blockIdx.x / nq, q = blockIdx.x - us * nq;
    const int s = st.perm[(size_t)parity * cfg.n_scenes + us];  // heaviest scenes first (see k_track)
    const int n = n_pts[s];
    if (!frame_reaches_track(n, cfg.max_pts)) return;  // offline_main.py:56: empty frames never reach track()
    const SceneHdr *hdr = st.hdr + s;
    const int T = hdr->n_tracks;
    if (q * 4 >= T) return;
    const double dt = dt_all[s];
    const int32_t *order = st.order + (size_t)s * cfg.t_cap;
    TrackRec *trk = st.trk + (size_t)s * cfg.t_cap;
    for (int j0 = q * 4; j0 < T; j0 += nq * 4) {
        const int j = j0 + g;
        const bool live = j < T;
        predict_one_track<DX>(cfg, st, trk + (live ? order[j] : 0), live, s, j, dt, Wj, lane, c, err);
    }
    if (err) atomicOr(&st.hdr[s].err, err);
}

static int waves_per_scene(const DevCfg &cfg)
{
    int nq = (cfg.tr_max_tracks + 3) / 4;
    return nq < 1 ? 1 : nq;
}

void launch_predict(const DevCfg &cfg, const DevState &st, const int32_t *n_pts, const double *dt, int parity, hipStream_t stream)
{
    if (pred_in_track(cfg) || cfg.fused) return;  // _predict_all runs at the head of k_track / inside k_scene there (mmw_kalman.hpp)
    const int nq = waves_per_scene(cfg);
    const int grid = cfg.n_scenes * nq + (tracks_dense(cfg, nq) ? kSpecialUnits : 0);
    if (cfg.dx == 9) mmw_launch(k_predict<9>, dim3(grid), dim3(64), 0, stream, cfg, st, n_pts, dt, nq, parity);
    else mmw_launch(k_predict<6>, dim3(grid), dim3(64), 0, stream, cfg, st, n_pts, dt, nq, parity);
}

}  // namespace mmw
