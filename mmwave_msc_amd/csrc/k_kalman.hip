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

// One wave = four tracks of one scene (16-lane groups).  Wave q of a scene's `nq` waves takes tracks
// 4q.., 4(q+nq).., so scenes with more than 4*nq tracks just loop.  DX = dim_x (6 or 9): a template so
// that every dot product below is fully unrolled with constant LDS offsets.
template <int DX>
__global__ __launch_bounds__(64) void k_predict(DevCfg cfg, DevState st, const int32_t *__restrict__ n_pts,
                                                const double *__restrict__ dt_all, int nq, int parity)
{
    __shared__ double lds[4 * kPredScratch];
    const int us = blockIdx.x / nq, q = blockIdx.x - us * nq;
    const int s = st.perm[(size_t)parity * cfg.n_scenes + us];  // heaviest scenes first (see k_track)
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int n = n_pts[s];
    if (n <= 0 || n > cfg.max_pts) return;  // offline_main.py:56: empty frames never reach track()
    const SceneHdr *hdr = st.hdr + s;
    const int T = hdr->n_tracks;
    if (q * 4 >= T) return;
    const double dt = dt_all[s];
    const int32_t *order = st.order + (size_t)s * cfg.t_cap;
    TrackRec *trk = st.trk + (size_t)s * cfg.t_cap;
    double *Wj = lds + g * kPredScratch;
    int err = 0;
    for (int j0 = q * 4; j0 < T; j0 += nq * 4) {
        const int j = j0 + g;
        const bool live = j < T;
        TrackRec *rec = trk + (live ? order[j] : 0);
        stage_record(rec, Wj, c);
        wave_sync();
        const double dtm = Wj[rLife] + dt;
        const double h = 0.5 * (dtm * dtm);
        if (live) {
            for (int k = c; k < 81; k += 16) {
                // A = F P.  F has ones on the diagonal, dt at (i,i+3), h at (i,i+6): the k-ordered dense
                // dot product reduces to these terms (the others are exact zeros).
                const int i = k / 9, cc = k - i * 9;
                if (i < DX && cc < DX) {
                    double a = Wj[rP + k];
                    if (i + 3 < DX) a += dtm * Wj[rP + (i + 3) * 9 + cc];
                    if (i + 6 < DX) a += h * Wj[rP + (i + 6) * 9 + cc];
                    Wj[pA + k] = a;
                }
            }
            if (c < DX) {
                double xn = Wj[rX + c];
                if (c + 3 < DX) xn += dtm * Wj[rX + c + 3];
                if (c + 6 < DX) xn += h * Wj[rX + c + 6];
                Wj[pXn + c] = xn;
            }
        }
        wave_sync();
        if (live) {
            const double dt2 = dtm * dtm, dt3 = dt2 * dtm, dt4 = dt2 * dt2;
            for (int k = c; k < 81; k += 16) {
                const int i = k / 9, cc = k - i * 9;
                if (i < DX && cc < DX) {
                    double b = Wj[pA + k];  // B = A F^T
                    if (cc + 3 < DX) b += Wj[pA + i * 9 + cc + 3] * dtm;
                    if (cc + 6 < DX) b += Wj[pA + i * 9 + cc + 6] * h;
                    double qn = 0.0;
                    if (i / 3 == cc / 3) {  // block_diag of Q_discrete_white_noise(dim=3) (constants.py:210-215)
                        const int qi = i % 3, qc = cc % 3, sdeg = qi + qc;
                        const double base = sdeg == 0 ? 0.25 * dt4 : sdeg == 1 ? 0.5 * dt3 : sdeg == 2 ? ((qi == 1) ? dt2 : 0.5 * dt2)
                                          : sdeg == 3 ? dtm : 1.0;
                        qn = base * cfg.kf_q_std;
                    }
                    const double pn = b + qn;
                    rec->P[k] = pn;
                    Wj[rP + k] = pn;
                }
            }
            if (c < DX) { const double xn = Wj[pXn + c]; rec->x[c] = xn; Wj[rX + c] = xn; }
        }
        wave_sync();
        // gate matrix: lane c < 6 of the group holds column c of C = P[:6,:6] + diag((spread/2)^2) + group_disp_est
        {
            const bool valid = live && c < 6;
            double v[6], det;
#pragma unroll
            for (int i = 0; i < 6; i++) v[i] = (c == i) ? 1.0 : 0.0;  // idle groups: identity
            if (valid) {
                const double hh = Wj[rSpr + c] / 2;
#pragma unroll
                for (int i = 0; i < 6; i++) v[i] = (Wj[rP + i * 9 + c] + ((i == c) ? hh * hh : 0.0)) + Wj[rGd + i * 6 + c];
            }
            const bool ok = lu6_inverse_cols(v, lane, det);
            if (live) {
                if (!ok) err |= ERR_SINGULAR;
                double *G = st.gate_buf + ((size_t)s * cfg.t_cap + j) * kGateRec;  // by effective_tracks position
                if (c >= 6 && c < 12) {
#pragma unroll
                    for (int r = 0; r < 6; r++) G[r * 6 + c - 6] = v[r];
                }
                if (c == 0) G[36] = dlog(fabs(det));
                if (c < 6) G[37 + c] = Wj[rX + c];
            }
        }
        wave_sync();
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
    const int nq = waves_per_scene(cfg);
    if (cfg.dx == 9) mmw_launch(k_predict<9>, dim3(cfg.n_scenes * nq), dim3(64), 0, stream, cfg, st, n_pts, dt, nq, parity);
    else mmw_launch(k_predict<6>, dim3(cfg.n_scenes * nq), dim3(64), 0, stream, cfg, st, n_pts, dt, nq, parity);
}

}  // namespace mmw
