// mmw_device.hpp -- device-side data layout shared by the HIP kernels and the C-ABI glue.
// gfx950 only (wave64, 160 KiB LDS/CU).  fp64 everywhere a decision is taken.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmw.h"

namespace mmw {

constexpr int kThreads = 256;       // 4 waves per workgroup, one workgroup per scene
constexpr int kWaves = kThreads / 64;
constexpr int kGateChunk = 16;      // tracks whose gate matrices sit in LDS at once
constexpr int kGateRec = 44;        // doubles per gate record: Ci[36] | log|det| | predicted position hx[6] | pad
constexpr int kLeafSize = 30;       // sklearn BallTree default leaf_size (DBSCAN passes it through)
constexpr int kSkNeighbors = 5;     // NearestNeighbors' default n_neighbors, left alone by DBSCAN.fit (sklearn/cluster/_dbscan.py:410-418):
                                    // clouds of n // 2 <= 5 points are answered by brute force, not by the tree (neighbors/_base.py:622-633)
constexpr int kMaxLeaves = 32;      // n <= 1920 -> <= 63 nodes -> <= 32 leaves
constexpr int kBigCloudMax = 30 * 64;  // = 1920: the largest cloud whose BallTree carve-up fits the LDS; larger ones (work list 2) run on slabs in global memory
constexpr int kMaxNodes = 63;

// per-scene error bits (sticky until mmw_reset)
// ERR_NONFINITE_*: apply_DBscan was reached with a NaN / an infinite value in its cloud -- sklearn's input validation raises
// ValueError there (Utils.py:272-278): "contains NaN" when any value is NaN, else "contains infinity" (oracle/c: orc_check_finite)
enum : int { ERR_SINGULAR = 1, ERR_DIVZERO = 2, ERR_CAPACITY = 4, ERR_BADCOUNT = 8, ERR_NONFINITE_NAN = 16, ERR_NONFINITE_INF = 32 };
constexpr int kDbRaised = -2;   // db_n of a frame whose apply_DBscan call "raised" (MMW_DB_RAISED; -1 = not called)

struct DevCfg {
    int32_t ring;            // FB_FRAMES_BATCH + 1
    int32_t db_min_samples;
    int32_t tr_max_tracks;
    int32_t kf_enable_est;
    int32_t model_min_input;
    int32_t dx;              // 9 / 6
    int32_t ring_rows;
    int32_t t_cap;
    int32_t max_pts;
    int32_t n_scenes;
    int32_t dense_min_units;  // Kalman kernels laid out over tracks when the context has more 4-track waves than this (mmw_kalman.hpp)
    int32_t seek_inner;       // Tracking.py:656 active: seek_inner_clusters after every associate_pointcloud (k_inner)
    int32_t db_points_thres, fb_frames_batch_static;
    int32_t big_live;        // this step's large clouds may be claimed WHILE k_track runs (side workers on, start-up frames over): they are pushed with a release; otherwise plainly, for the kernels behind k_track
    int32_t fused;           // the step is k_scene (one workgroup runs a scene's whole track(): k_scene.hip) + the worker blocks of k_post: contexts whose scenes are all resident at once
    int32_t epoch;           // number of this step (mmw_api.hip: counts committed steps of the context): tags the claim words of the DBSCAN queues
    int32_t var_ring, side_worker;   // side_worker: k_chain runs beside k_track on a second stream (mmw_api.hip); var_ring: a global ring size was changed (mmw_set_batch_size): k_track reads ring sizes from the headers
    double db_spread_thres, db_inner_eps;
    double db_z_weight, db_range_weight, db_eps;
    double tr_lifetime_dynamic, tr_lifetime_static, tr_vel_thres, tr_gate;
    double kf_q_std, kf_p_init, kf_group_disp_est_init, kf_a_n, kf_est_pointnum;
    double kf_spread_lim[6];
    double kf_a_spr;
    double intensity_mu, intensity_std;
    double s_height, tilt_cos, tilt_sin;
    double m_x, m_y, m_z, fade_max, fade_min, fade_weight;   // output step (k_table)
};

// 64 B header per scene
struct SceneHdr {
    int32_t n_tracks;
    int32_t g_len;                 // frames in the global BatchedData ring
    int32_t g_n[MMW_RING_MAX];     // rows per frame, oldest first
    int32_t g_slot[MMW_RING_MAX];  // permutation: physical slot of the k-th oldest frame
    int32_t need_db;               // set by the track kernel: run apply_DBscan this frame
    int32_t err;
    int32_t db_u;
    int32_t next_uid;              // TrackBuffer.next_track_id (Tracking.py:509,588)
    int32_t n_upd;                 // tracks after _maintain_tracks of this frame = what _update_all covers (k_track -> k_post)
    int32_t skipped;               // bit 0: the last frame was empty for this scene (k_track returned at once): its tracks are in no update list;
                                   // bits 8..15: size of the global ring after BatchedData.change_buffer_size (0 = FB_FRAMES_BATCH + 1);
                                   // bits 16..23: two per PHYSICAL slot p of the global ring -- bit 16 + 2p: the frame stored there holds a
                                   // NaN, bit 17 + 2p: an infinite value (any of the 8 columns; set by whoever writes a frame into the slot)
};
static_assert(sizeof(SceneHdr) == 64, "SceneHdr");
constexpr int kSkipRingShift = 8, kSkipRingMask = 255, kSkipNfShift = 16, kSkipNfMask = 255;
__host__ __device__ inline int hdr_ring_size(int skipped) { return (skipped >> kSkipRingShift) & kSkipRingMask; }
// bit 0: a NaN, bit 1: an infinite value among the 8 columns of one row (x, y, z, vx, vy, vz, doppler, peakVal).  One
// v_cmp_class_f64 per value; the second pass only for the (rare) row that has one.
__device__ __forceinline__ int row_nonfinite_bits(const double2 (&r)[4])
{
    constexpr int kNanInf = 0x3 | 0x4 | 0x200;   // sNaN | qNaN | -inf | +inf
    bool any = false;
#pragma unroll
    for (int u = 0; u < 4; u++) any = any || __builtin_amdgcn_class(r[u].x, kNanInf) || __builtin_amdgcn_class(r[u].y, kNanInf);
    if (!any) return 0;
    bool nan = false, inf = false;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        nan = nan || __builtin_amdgcn_class(r[u].x, 0x3) || __builtin_amdgcn_class(r[u].y, 0x3);
        inf = inf || __builtin_amdgcn_class(r[u].x, 0x204) || __builtin_amdgcn_class(r[u].y, 0x204);
    }
    return (nan ? 1 : 0) | (inf ? 2 : 0);
}
// The flags word after a frame with row bits `frame_bits` has been written into physical slot `phys`, and what the live
// frames gs[0 .. len) hold together (bit 0 NaN, bit 1 inf).
__host__ __device__ inline int nf_flags_with(int flags, int phys, int frame_bits) { return (flags & ~(3 << (2 * phys))) | ((frame_bits & 3) << (2 * phys)); }
// which ValueError sklearn raises for a cloud with these bits (NaN takes precedence)
__host__ __device__ inline int nf_error_of(int bits) { return (bits & 1) ? ERR_NONFINITE_NAN : ((bits & 2) ? ERR_NONFINITE_INF : 0); }

// One ClusterTrack.  187 doubles = 1496 B.
struct TrackRec {
    double x[9];
    double P[81];
    double centroid[6];
    double minv[6];
    double maxv[6];
    double spread[6];
    double gd[36];
    double n_est;
    double lifetime;
    int32_t point_num;
    int32_t is_static;
    int32_t ring_len;
    int32_t uid;                   // creation ordinal inside the scene (stable identity for views)
    int32_t ring_n[MMW_RING_MAX];
    int32_t ring_slot[MMW_RING_MAX];
    float kp[MMW_NKP];
    int32_t inner;                 // seek_inner_clusters state (cfg.seek_inner only): bits 0..7 = track.batch.size (what
                                   // BatchedData.change_buffer_size set, Tracking.py:60-64), bit 8 = associate_pointcloud ran this frame
};
constexpr int kInnerTouched = 256;
static_assert(sizeof(TrackRec) == 187 * 8, "TrackRec");

struct DevState {
    SceneHdr *hdr;        // [S]
    int32_t *order;       // [S][t_cap]  effective_tracks position -> physical record (always a permutation)
    TrackRec *trk;        // [S][t_cap]
    double *trk_ring;     // [S][t_cap][ring][ring_rows][8]
    double *g_ring;       // [S][ring][max_pts][8]
    const float *default_posture;  // [57]
    unsigned long long *stats;     // [kStatSlots][32] algorithmic-byte / work counters (see mmw_stats_get), summed on read-out
    int32_t *db_list;              // [4][S] scenes whose apply_DBscan the cell count could not finish this step, filled by k_track: list 3 = clouds <= 256 points (k_post), list 2 = clouds of more than kBigCloudMax points (k_dbscan_huge); rows 0 and 1 are the rings of the two queues below
    int32_t *db_count;             // [2][4] list lengths, double-buffered by step parity (list 3)
    int32_t *q;                    // [kQWords] the early queue = list 0 (n_scenes entries, 0 = empty, else scene + 1), per step parity p at
                                   // q[8p + ...]: kQCount pushes, kQHead claims, kQDone finished items of the step (reset a step ahead by
                                   // k_track); shared: q[kQStop] = last step whose k_post has begun (monotonic), q[kQTimeout];
                                   // the same three words at q[kQBig + 8p + ...] for the queue of the larger clouds (ring = list 1)
    int32_t *perm;                 // [2][S] by step parity: scene handled by unit b of k_predict / k_track / k_post, heaviest scenes (most tracks) first; k_post builds the next step's
    double *gate_buf;              // [S][t_cap][kGateRec] gate matrices of this frame, by effective_tracks position (k_predict -> k_track)
    int32_t *upd_count;            // [2][kUpdWords] by step parity: tracks in each of the kUpdShards update lists of this frame (k_track -> k_post, next k_predict)
    int32_t *upd_list;             // [2][kUpdShards][upd_region] ... and the tracks: one packed word each (upd_pack: scene, position, record slot) --
                                   // _update_all / _predict_all are laid out over the TRACKS of the context, four per wave
    int32_t *spc_count;            // [2] by step parity: scenes the next k_predict cannot take from the update lists ...
    int32_t *spc_list;             // [2][S][2] ... (scene, first new track): scenes that spawned tracks this frame
    int32_t *inner_buf;            // [S][kInnerHdr + inner_cap] seek_inner_clusters calls of the last frame (cfg.seek_inner; mmw_get_inner)
    int32_t inner_cap;             // label words per scene
    char *huge_scratch;            // [workers][huge_stride] BallTree carve-ups of the clouds of more than kBigCloudMax points (k_dbscan_huge); null when no ring of the context can hold one
    size_t huge_stride;
};
// The update lists (track-wise Kalman kernels).  A scene's workgroup of k_track appends its T tracks to the list of its SHARD
// (workgroup index mod shards: eight counters instead of one word that every workgroup of the launch adds to) with one atomicAdd;
// the consumers' unit w serves shard w mod shards, entries 4 (w / shards) .. + 3, so the list entry and the list's length are ONE
// round trip (the entry is read speculatively, clamped into the region) and the record's address is in the entry: two dependent
// round trips in front of a track's arithmetic where the lists "scenes by track count" of rounds 2-4 had four (bin counts, list
// entry, order[], record).
constexpr int kUpdShards = 8, kUpdWords = 16;
__host__ __device__ inline int kalman_waves_per_scene(int tr_max_tracks) { const int nq = (tr_max_tracks + 3) / 4; return nq < 1 ? 1 : nq; }
__host__ __device__ inline int upd_shards(int n_units) { return n_units < kUpdShards ? (n_units < 1 ? 1 : n_units) : kUpdShards; }
__host__ __device__ inline size_t upd_region(int n_scenes, int t_cap) { return ((size_t)(n_scenes + kUpdShards - 1) / kUpdShards + 1) * (size_t)t_cap; }
__host__ __device__ inline int upd_pack(int s, int j, int slot) { return (s << 12) | (j << 6) | slot; }   // (t_cap <= 63, n_scenes < 2^19: tracks_dense)
constexpr int kUpdMaxScenes = 1 << 19;

// Contexts of at most this many scenes run a two-launch step (mmw_kalman.hpp: pred_in_track, k_dbscan.hip: k_post takes the large
// clouds): their step is launch latency.  768 = what is resident at once with the PRED build of k_track (three workgroups
// per CU); measured 0.0777 -> 0.0705 ms at 768 scenes, 0.0937 -> 0.1128 at 1024.
constexpr int kSmallContextScenes = 768;
constexpr int kQCount = 0, kQHead = 1, kQDone = 2, kQStop = 3, kQTimeout = 4;
// The head (claim) word of a queue carries the number of the step the queue serves in its upper bits (q_tag), written when the
// queue is reset a step ahead: a claim is a compare-and-swap on tag + index, so a side-stream worker of ANOTHER step -- one that
// checked the stop epoch, was held up (six processes time-sharing the GPU: milliseconds), and looks at the queue of its parity two
// steps later -- cannot take an entry that is not its step's (it did: scripts/dual_run.py, profiles/NOTEBOOK.md round 6; the
// cloud was clustered correctly but its labels / db_n went to the OLD step's output buffers).  Consumers on the context's own
// stream run inside their step and mask the tag.
constexpr int kQTagShift = 20, kQIdxMask = (1 << kQTagShift) - 1;   // (n_scenes < 2^19 entries per step)
__host__ __device__ inline int q_tag(int epoch) { return (epoch & 0x7ff) << kQTagShift; }
constexpr int kQBig = 16;   // q[kQBig + 8p + kQCount/kQHead/kQDone]: the queue of the clouds of more than 256 points
constexpr int kQWords = 32;
constexpr int kEarlyU = 0;     // clouds of at least this many points go to the early queue (k_track.hip); a threshold above the clutter level (180) measured slower
constexpr int kInnerHdr = 2 + 16;  // calls, labels stored, rows of the first 16 calls

// n_pts[s] of a step: 1..max_pts = track() on that many points; 0 = the frame never reaches track() (offline_main.py:56
// skips empty frames); MMW_EMPTY_FRAME (-1) = track() IS called, on an empty point cloud (the reference then predicts,
// ages and expires tracks, runs _update_all and pushes an empty frame into the ring); anything else is ERR_BADCOUNT.
__device__ __forceinline__ bool frame_reaches_track(int n, int max_pts) { return n != 0 && n >= -1 && n <= max_pts; }

// Counters are spread over kStatSlots copies (one 256-byte line each, picked by scene index): thousands of
// workgroups adding to ONE address serialise in the memory-side atomic unit and that tail was longer than
// the kernels themselves.
constexpr int kStatSlots = 256;
constexpr int kStatWords = 32;
constexpr int kStatFeatBytes = 30;  // k_features: algorithmic bytes (mmw_stats_get_ext)
constexpr int kStatFeatRows = 31;   // k_features: feature tensors written
__device__ inline unsigned long long *stats_slot(const DevState &st, int scene)
{
    return st.stats ? st.stats + (size_t)(scene & (kStatSlots - 1)) * kStatWords : nullptr;
}

// One row of a frame (8 columns: x, y, z, vx, vy, vz, doppler, peakVal) into registers.  fp64 rows (mmw_step) are four
// 16-byte loads; fp32 rows (mmw_step_f32: 32 bytes per point, what a radar front end or a CSV reader produces) two, each
// value promoted to fp64 in registers -- exactly; from there on both entries run the same instructions on the same bits.
// Rows past the frame's capacity load nothing.
template <bool F32>
__device__ __forceinline__ void load_point_row(const void *frame, int i, bool valid, double2 (&r)[4])
{
    if constexpr (F32) {
        const float4 *src = reinterpret_cast<const float4 *>(frame);
        const float4 a = valid ? src[i * 2] : float4{0.f, 0.f, 0.f, 0.f}, b = valid ? src[i * 2 + 1] : float4{0.f, 0.f, 0.f, 0.f};
        r[0] = double2{(double)a.x, (double)a.y};
        r[1] = double2{(double)a.z, (double)a.w};
        r[2] = double2{(double)b.x, (double)b.y};
        r[3] = double2{(double)b.z, (double)b.w};
    } else {
        const double2 *src = reinterpret_cast<const double2 *>(frame);
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = valid ? src[i * 4 + u] : double2{0.0, 0.0};
    }
}
// start of scene s's frame: rows of 8 values, fp64 or fp32
__device__ __forceinline__ const void *frame_of(const void *pts_all, int s, int NP, bool f32)
{
    return reinterpret_cast<const char *>(pts_all) + (size_t)s * NP * (f32 ? 32 : 64);
}

__host__ __device__ inline size_t trk_ring_stride_track(const DevCfg &c) { return (size_t)c.ring * c.ring_rows * 8; }

}  // namespace mmw
