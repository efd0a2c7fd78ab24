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
// The track-wise launch is persistent: 12 one-wave workgroups per CU, each walking its shard of the lists with the launch's stride
// (same box, alternating, K = T: one workgroup per four tracks 26.1-26.3 us, 16 per CU 24.7-25.2, 12 per CU 23.7-24.2, 8 per CU
// 27.0-27.2; the mixed population 18.4-18.7 either way -- profiles/NOTEBOOK.md round 5).
#ifndef MMW_PRED_RESIDENT   // (diagnostic builds: dense one-wave workgroups per CU of the persistent launch)
#define MMW_PRED_RESIDENT 12
#endif
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
        const int prev = parity ^ 1, n_dense = (int)gridDim.x - kSpecialUnits, unit = blockIdx.x;   // (the dense units this launch runs: launch_predict)
        // (unit w < n_dense serves shard w mod shards of last frame's update lists; every unit without dense work -- the
        //  kSpecialUnits behind the dense range always, and the idle rest of every shard: after a frame in which EVERY scene
        //  spawned tracks, the first frames after a reset, that is all of them; 64 waves alone took 555 us for the 4096 scenes
        //  of the start-up -- serves the two lists below)
        // (the dependent round trips of a wave are what this launch lasts, at 12 waves per CU: the shard lengths and this group's
        //  list entry are requested together, then the scene's four words AND the record, whose address the entry holds: two)
        const int nsh = upd_shards(cfg.n_scenes * nq);
        int tot[kUpdShards];
#pragma unroll
        for (int i = 0; i < kUpdShards; i++) tot[i] = st.upd_count[prev * kUpdWords + i];
        if (unit < n_dense) {
            int e;
            UpdCursor C = upd_cursor(cfg, st, unit, n_dense, prev, g, e);
            if (C.k - g < C.tot) {
                do {
                    bool live = C.k < C.tot;
                    const int s = live ? e >> 12 : 0, j = live ? (e >> 6) & 63 : 0;
                    const int n = n_pts[s], nt = st.hdr[s].n_tracks;
                    const double dt = dt_all[s];
                    TrackRec *rec = st.trk + (size_t)s * cfg.t_cap + (live ? (e & 63) : 0);
                    stage_record<16>(rec, Wj, c);   // (requested with the scene's words, before they are looked at)
                    wave_sync();
                    // (j < n_tracks: a scene that mmw_reset_scenes has emptied since the lists were built holds no track -- its
                    //  stale records must not be predicted, nor their error bits come back on the fresh scene)
                    live = live && frame_reaches_track(n, cfg.max_pts) && j < nt;
                    if (__any(live)) {
                        int e1 = 0;
                        predict_staged_track<DX>(cfg, st, rec, live, s, j, dt, Wj, lane, c, e1);
                        if (e1 && live) atomicOr(&st.hdr[s].err, e1);
                    }
                    wave_sync();   // (the staging area is the next entry's)
                    C.k += C.stride4;
                    if (C.k - g < C.tot) e = C.list[C.k < C.last ? C.k : C.last];
                } while (C.k - g < C.tot);
                return;
            }
        }
        // this unit's place among the launch's idle ones: the units of shard i are w = i, i + shards, ...; the first
        // ceil(tot[i] / 4) of them are busy
        int pool = (int)gridDim.x - n_dense, me = unit < n_dense ? 0 : unit - n_dense;
#pragma unroll
        for (int i = 0; i < kUpdShards; i++) {
            if (i < nsh) {
                const int units_i = (n_dense - i + nsh - 1) / nsh;
                int busy = (tot[i] + 3) >> 2;
                busy = busy < units_i ? busy : units_i;
                const int before = unit < n_dense ? (unit > i ? (unit - i + nsh - 1) / nsh : 0) : units_i;
                pool += units_i - busy;
                me += before > busy ? before - busy : 0;
            }
        }
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

static int waves_per_scene(const DevCfg &cfg) { return kalman_waves_per_scene(cfg.tr_max_tracks); }

void launch_predict(const DevCfg &cfg, const DevState &st, const int32_t *n_pts, const double *dt, int parity, hipStream_t stream)
{
    if (pred_in_track(cfg) || cfg.fused) return;  // _predict_all runs at the head of k_track / inside k_scene there (mmw_kalman.hpp)
    const int nq = waves_per_scene(cfg);
    int grid = cfg.n_scenes * nq;
    if (tracks_dense(cfg, nq)) {
        // the track-wise launch is PERSISTENT: as many one-wave workgroups as the chip holds at once (MMW_PRED_RESIDENT per CU), each
        // walking its shard of the lists with that stride
        static const int resident = []() {
            int dev = 0, n_cu = 256;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
            return n_cu * MMW_PRED_RESIDENT;
        }();
        if (resident >= kUpdShards && resident < grid) grid = resident;
        grid += kSpecialUnits;
    }
    if (cfg.dx == 9) mmw_launch(k_predict<9>, dim3(grid), dim3(64), 0, stream, cfg, st, n_pts, dt, nq, parity);
    else mmw_launch(k_predict<6>, dim3(grid), dim3(64), 0, stream, cfg, st, n_pts, dt, nq, parity);
}

}  // namespace mmw
