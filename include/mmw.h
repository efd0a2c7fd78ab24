/*
 * mmw.h -- C-ABI of libmmw_hip.so: the MI355X (gfx950) implementation of the
 * per-frame point-cloud hot path of AsteriosPar/mmWave_MSc.
 *
 * The reference has no FFI for this path: it is plain Python
 * (`TrackBuffer.track`, `TrackBuffer.estimate_posture`, `Utils.normalize_data`,
 * `Utils.apply_DBscan`).  Each entry point below names the reference interface
 * it replaces (file:line under the reference's src/); INTEGRATION.md shows the
 * ctypes binding a reference maintainer would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no C++/torch types.
 *  - Every function returns 0 on success, <0 on error (MMW_E_*); never throws.
 *    `mmw_last_error(ctx)` returns a message (ctx may be NULL for create errors).
 *  - One context = S independent scenes (one reference TrackBuffer + global
 *    BatchedData each) resident on ONE device.  A context is used from one host
 *    thread at a time; calls are ordered on the context's HIP stream and are
 *    asynchronous unless stated ("sync").
 *  - "dev" pointers are device memory of the context's device (hipMalloc,
 *    torch.Tensor.data_ptr(), or mmw_dev_alloc); "host" pointers are host memory.
 *  - There is NO CPU fallback: creation fails if no gfx950-capable HIP device
 *    is usable.
 *  - Numerics: tracker state and every decision in fp64 (the reference's numpy
 *    dtype), features in fp32 (what Keras feeds the CNN).
 */
#ifndef MMW_H
#define MMW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMW_RING_MAX 4      /* FB_FRAMES_BATCH + 1 <= 4 */
#define MMW_NKP 57          /* 19 joints x (x,y,z)  (preprocessing.py:377) */
#define MMW_MAX_PTS_LIMIT 1024
#define MMW_TRACK_CAP_LIMIT 64
#define MMW_EMPTY_FRAME (-1) /* n_pts value: TrackBuffer.track() on an empty point cloud (0 = frame skipped) */

#define MMW_OK 0
#define MMW_E_ARG (-1)        /* bad argument / size */
#define MMW_E_SINGULAR (-2)   /* a 6x6 gate/innovation matrix was singular (numpy raises LinAlgError) */
#define MMW_E_DIVZERO (-3)    /* (N_est-1)*N == 0 in _get_Rc (Python raises ZeroDivisionError) */
#define MMW_E_CAPACITY (-4)   /* more tracks than track_cap */
#define MMW_E_HIP (-5)        /* HIP runtime error */
#define MMW_E_NODEVICE (-6)   /* no usable gfx950 device: the library has no CPU path */
#define MMW_E_NONFINITE (-7)  /* apply_DBscan was reached with a NaN or an infinite value in its cloud: sklearn's input validation raises
                                 ValueError there (Utils.py:272-278 -> DBSCAN.fit_predict -> check_array).  Any of the 8 columns of any row
                                 of the global ring counts; the gate never takes a point whose columns 0..5 are not finite
                                 (Tracking.py:559-563), so such rows always end up there.  The frame is in the ring, nothing was
                                 clustered or cleared -- exactly the state the exception leaves -- and the scene raises again on every
                                 frame the row is still in the ring while the trigger holds (Tracking.py:693-697) */
#define MMW_DB_RAISED (-2)    /* db_n of a scene whose apply_DBscan call of this frame raised (MMW_E_NONFINITE); -1 = not called */

/* Mirrors constants.py 1:1 (line numbers = /root/reference/src/constants.py). */
typedef struct mmw_config {
    int32_t fb_frames_batch;        /* FB_FRAMES_BATCH :66  (ring length = +1) */
    int32_t db_min_samples;         /* DB_MIN_SAMPLES_MIN :73 */
    int32_t tr_max_tracks;          /* TR_MAX_TRACKS :85 */
    int32_t kf_enable_est;          /* KF_ENABLE_EST :100 */
    int32_t model_min_input;        /* MODEL_MIN_INPUT :111 */
    int32_t dim_x;                  /* MOTION_MODEL :246 -> 9 CONST_ACC_MODEL (176-215), 6 CONST_VEL_MODEL (218-243) */
    int32_t ring_rows;              /* rows stored per frame of a per-track ring; >=64 (format_single_frame reads [:64]) */
    int32_t track_cap;              /* capacity of effective_tracks per scene; 0 = (TR_MAX_TRACKS-1)+ring*max_pts/min_samples+1, <=64 */
    double db_z_weight;             /* DB_Z_WEIGHT :70 */
    double db_range_weight;         /* DB_RANGE_WEIGHT :71 */
    double db_eps;                  /* DB_EPS :72 */
    double tr_lifetime_dynamic;     /* TR_LIFETIME_DYNAMIC :86 */
    double tr_lifetime_static;      /* TR_LIFETIME_STATIC :87 */
    double tr_vel_thres;            /* TR_VEL_THRES :88 */
    double tr_gate;                 /* TR_GATE :89 */
    double kf_q_std;                /* KF_Q_STD :93 (passed as var= to Q_discrete_white_noise :212) */
    double kf_p_init;               /* KF_P_INIT :96 */
    double kf_group_disp_est_init;  /* KF_GROUP_DISP_EST_INIT :97 */
    double kf_a_n;                  /* KF_A_N :101 */
    double kf_est_pointnum;         /* KF_EST_POINTNUM :102 */
    double kf_spread_lim[6];        /* KF_SPREAD_LIM :103 */
    double kf_a_spr;                /* KF_A_SPR :104 */
    double intensity_mu;            /* INTENSITY_MU :108 */
    double intensity_std;           /* INTENSITY_STD :109 */
    double s_height;                /* S_HEIGHT :41 */
    double tilt_cos;                /* cos(radians(S_TILT)) :42, Utils.py:315-323 */
    double tilt_sin;                /* sin(radians(S_TILT)) */
    float default_posture[MMW_NKP]; /* MODEL_DEFAULT_POSTURE :112-172 */
    int32_t kalman_dense_min_units; /* not a reference constant: layout of the Kalman kernels.  0 = automatic (laid out over tracks
                                       when the context holds more than 512 scenes and > 1024 four-track waves; smaller contexts run
                                       a two-launch step per scene), < 0 = always per scene, n > 0 = over tracks from n waves on
                                       (tests run both layouts) */
    int32_t seek_inner;             /* 0 = Tracking.py:656 stays commented out (the reference as shipped); 1 = run
                                       ClusterTrack.seek_inner_clusters (Tracking.py:409-448) after every associate_pointcloud */
    int32_t db_points_thres;        /* DB_POINTS_THRES :76   (seek_inner_clusters) */
    int32_t fb_frames_batch_static; /* FB_FRAMES_BATCH_STATIC :67 */
    int32_t chain_side_stream;      /* not a reference constant: where the small-cloud DBSCAN (pair-count screen, BallTree chain) of a
                                       frame runs.  0 = automatic (contexts of > 512 scenes: worker blocks on a second stream beside
                                       the association kernel, whatever they have not taken by its end in the post kernel), -1 = post
                                       kernel only, 1 = always with the side stream (tests run both), 2 = as 1 without the check that
                                       the side stream really runs beside the context's (profilers that serialise kernels fail it: the
                                       workers then start, find nothing to claim in time and leave -- correct, and visible as a launch),
                                       3 = as 1, and the workers of a step do not start before the step's first kernel does (an event
                                       recorded on the context's stream at the head of every step: ~7 us per step; for callers that
                                       queue their own work on the context's stream between steps -- by default the side stream paces
                                       itself by the steps' stop epochs and its workers may poll empty queues a little early) */
    double db_spread_thres;         /* DB_SPREAD_THRES :77 */
    double db_inner_eps;            /* DB_INNER_EPS :78 */
    double m_x, m_y, m_z;           /* M_X, M_Y, M_Z :31-33  monitoring point (calc_projection_points, Utils.py:180-219) */
    double v_screen_fade_size_max;  /* V_SCREEN_FADE_SIZE_MAX :48 */
    double v_screen_fade_size_min;  /* V_SCREEN_FADE_SIZE_MIN :49 */
    double v_screen_fade_weight;    /* V_SCREEN_FADE_WEIGHT :50 */
    int32_t fused_step;             /* not a reference constant: which kernels make a step.  0 = automatic: a context of <= 512 scenes
                                       whose scenes are all resident at once (two workgroups per CU up to 512 points per frame), runs
                                       TrackBuffer.track of a scene in ONE workgroup start to finish (k_scene: the step is one scene's
                                       latency there), others the bulk kernels; 1 = the one-workgroup step whenever the configuration
                                       allows it (not with seek_inner, resized rings, track_cap > 63 or the side-stream workers);
                                       -1 = never (tests run both) */
    int32_t reserved_;
} mmw_config;

/* One entry of TrackBuffer.effective_tracks (Tracking.py:139-230), flattened:
 * ClusterTrack.{state.x, state.P, cluster.*, spread_est, group_disp_est, N_est,
 * lifetime, batch (lengths), keypoints}.  P is stored 9x9 row-major (the
 * leading 6x6 block is used for CONST_VEL_MODEL). */
typedef struct mmw_track_record {
    double x[9];
    double P[81];
    double centroid[6];
    double min_vals[6];
    double max_vals[6];
    double spread_est[6];
    double group_disp_est[36];
    double n_est;
    double lifetime;
    int32_t point_num;
    int32_t is_static;              /* cluster.status: STATIC=True (Tracking.py:17,132-136) */
    int32_t ring_len;               /* len(track.batch.buffer) */
    int32_t ring_n[MMW_RING_MAX];   /* rows per frame, oldest first */
    int32_t uid;                    /* creation ordinal in its scene (TrackBuffer.next_track_id, Tracking.py:588) */
    float keypoints[MMW_NKP];
} mmw_track_record;

/* Fixed-size per-track summary exchanged between GPUs. */
typedef struct mmw_track_summary {
    int32_t scene;      /* global scene id (scene_base + local index) */
    int32_t slot;       /* position in effective_tracks */
    int32_t alive;      /* 1 if slot < n_tracks */
    int32_t is_static;
    int32_t point_num;
    float lifetime;
    float x[9];
    float centroid[6];
    float keypoints[MMW_NKP];
    /* the output step after the path, fused into the table kernel: Visualizer.calc_fade_square (Visualizer.py:14-29)
     * = Utils.calc_projection_points (Utils.py:180-219) of the head keypoint (x index 3, y index 41, z index 22)
     * relative to the track position onto the screen plane y = 0, and the side of the faded square (shrinks with
     * range, clamped to [V_SCREEN_FADE_SIZE_MIN, V_SCREEN_FADE_SIZE_MAX]).  fp64 arithmetic, rounded once. */
    float fade_x, fade_z, fade_size;
} mmw_track_summary;

typedef struct mmw_ctx mmw_ctx;

/* constants.py defaults. */
int mmw_config_default(mmw_config *cfg);

/* TrackBuffer() + BatchedData() for `n_scenes` scenes (Tracking.py:504-511, 38-41;
 * offline_main.py:32-34).  max_pts = largest point count of one frame (<= MMW_MAX_PTS_LIMIT).  apply_DBscan clusters
 * the unassigned part of the ring, up to (FB_FRAMES_BATCH + 1) * max_pts <= 4096 points: up to 1920 points its BallTree
 * lives in on-chip memory; a context whose ring can hold more additionally gets a slower global-memory path for those
 * clouds (one more launch per step, only in such contexts). */
int mmw_create(const mmw_config *cfg, int32_t n_scenes, int32_t max_pts, int32_t device, mmw_ctx **out);
int mmw_destroy(mmw_ctx *ctx);
const char *mmw_last_error(const mmw_ctx *ctx);
/* Fresh TrackBuffer/BatchedData for every scene. */
int mmw_reset(mmw_ctx *ctx);
/* ... for the scenes whose flag is non-zero only (host array of n_scenes words): how a batched caller recovers ONE scene --
 * e.g. after MMW_E_CAPACITY, which names the scene; the error bits are per scene and the other scenes' state stays valid. */
int mmw_reset_scenes(mmw_ctx *ctx, const int32_t *scene_flags);
/* The sticky error bits of every scene (host array of n_scenes words; 0 = none): 1 singular 6x6 matrix, 2 division by zero
 * in _get_Rc, 4 more tracks than track_cap (the tracks that did not fit were dropped: this scene differs from the reference
 * from then on), 8 a point count the context was not sized for, 16 / 32 apply_DBscan reached with a NaN / an infinite value
 * in its cloud (MMW_E_NONFINITE; 16 = sklearn's "contains NaN" message, which it prefers when both are present, 32 = "contains
 * infinity").  mmw_check reports the first one as its return code. */
int mmw_get_errors(mmw_ctx *ctx, int32_t *err_bits);
/* Clears the given error bits of the scenes whose flag is non-zero (host array of n_scenes words; NULL = every scene) and
 * nothing else: for errors that leave the scene's state valid -- a caller that catches the reference's ValueError
 * (MMW_E_NONFINITE) and carries on sees the same state the reference is in, and the error comes back on the next frame if
 * it still applies.  ONE flavour of MMW_E_NONFINITE does NOT leave the reference's state: with mmw_config.seek_inner, a
 * non-finite doppler / peakVal of an ASSIGNED point reaches the inner apply_DBscan (Tracking.py:440), whose ValueError leaves
 * the reference's track() in the middle of _associate_points_to_tracks -- no further track associated, no _maintain_tracks, no
 * _update_all, no add_frame --, while here only that track's inner clustering is skipped and the frame completes (db_n is then
 * NOT MMW_DB_RAISED: the frame's own call did not raise).  Like MMW_E_CAPACITY such a scene differs from the reference from
 * then on: reset it (mmw_reset_scenes), do not clear the bit and carry on. */
#define MMW_ERRBIT_SINGULAR 1
#define MMW_ERRBIT_DIVZERO 2
#define MMW_ERRBIT_CAPACITY 4
#define MMW_ERRBIT_BADCOUNT 8
#define MMW_ERRBIT_NONFINITE_NAN 16
#define MMW_ERRBIT_NONFINITE_INF 32
int mmw_clear_errors(mmw_ctx *ctx, const int32_t *scene_flags, int32_t bits);
/* BatchedData.pop_frame() (Tracking.py:66-71; its caller is preprocessing.py:264): drop the oldest frame of the global
 * ring of every scene whose flag is non-zero (host array of n_scenes words; NULL = every scene). */
int mmw_pop_frame(mmw_ctx *ctx, const int32_t *scene_flags);
/* BatchedData.change_buffer_size(new_size) (Tracking.py:60-64) on the global ring of the flagged scenes (host array of
 * n_scenes words; NULL = every scene): from the next add_frame on, frames are popped while len >= new_size.  Sizes
 * above FB_FRAMES_BATCH + 1 act like it (the reference's deque has that maxlen); new_size < 1 is MMW_E_ARG (the
 * reference's add_frame would never terminate).  mmw_reset restores the default. */
int mmw_set_batch_size(mmw_ctx *ctx, const int32_t *scene_flags, int32_t new_size);
/* BatchedData(init_data) (Tracking.py:38-41): the global ring of `scene` becomes ONE frame holding rows[n][8] (host,
 * n <= max_pts) instead of the empty frame a default BatchedData() starts with.  Sync. */
int mmw_set_batch_frame(mmw_ctx *ctx, int32_t scene, const double *rows, int32_t n);
/* mmw_config.chain_side_stream at run time: on != 0 -> the small-cloud DBSCAN workers run on a second stream beside the
 * association kernel from the next mmw_step on, 0 -> in the post kernel only.  A caller that runs its own kernels beside
 * the tracker (the CNN of the previous frame on another stream) may prefer them off. */
int mmw_set_chain_side_stream(mmw_ctx *ctx, int32_t on);
/* Run on a caller-owned hipStream_t; NULL = the context's own (non-blocking) stream.
 * Note for callers that share device buffers with another runtime: calls on DEVICE pointers are ordered with that
 * runtime's work only if both use the same stream.  torch reports the legacy default stream as
 * torch.cuda.current_stream().cuda_stream == 0, which is NULL here, i.e. NOT torch's stream: pass MMW_STREAM_LEGACY
 * (HIP's hipStreamLegacy handle) for it, or -- better -- a torch.cuda.Stream() both sides use. */
#define MMW_STREAM_LEGACY ((void *)1)
int mmw_set_stream(mmw_ctx *ctx, void *hip_stream);
int mmw_synchronize(mmw_ctx *ctx);                                   /* sync */
/* Hand-over WITHOUT a host wait: whatever is queued on `hip_stream` after this call starts only when everything queued on the
 * context's stream so far has finished (an event recorded on the context's stream, hipStreamWaitEvent on the other).  For a
 * consumer on another stream of what the context wrote into device memory -- the RCCL all-gather of the track table on the
 * communicator's stream, a torch kernel reading mmw_features' rows.  hip_stream: a raw hipStream_t; NULL =
 * the legacy default stream (what torch reports as current_stream().cuda_stream == 0), MMW_STREAM_LEGACY the same. */
int mmw_stream_wait(mmw_ctx *ctx, void *hip_stream);
/* The other direction: whatever the context queues on ITS stream after this call starts only when everything queued on
 * `hip_stream` so far has finished.  For device buffers the context is about to REWRITE while a consumer on another stream may
 * still be reading them (the track table of the previous all-gather: write-after-read). */
int mmw_wait_stream(mmw_ctx *ctx, void *hip_stream);
int mmw_get_dims(const mmw_ctx *ctx, int32_t *n_scenes, int32_t *max_pts, int32_t *track_cap, int32_t *ring, int32_t *ring_rows);

/* Thin device-memory helpers so a ctypes host needs no HIP binding. */
int mmw_dev_alloc(mmw_ctx *ctx, size_t bytes, void **dptr);
int mmw_dev_free(mmw_ctx *ctx, void *dptr);
int mmw_memcpy_h2d(mmw_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);   /* stream-ordered, sync on return */
int mmw_memcpy_d2h(mmw_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);   /* sync */

/* Utils.normalize_data + point_transform_to_standard_axis (Utils.py:294-434):
 * raw[S][max_pts][5] = (x,y,z,doppler,peakVal) -> pts[S][max_pts][8], kept rows
 * compacted in input order; n_out[S].  All dev pointers. */
int mmw_normalize(mmw_ctx *ctx, const double *raw, const int32_t *n_raw, double *pts, int32_t *n_out);
/* The same from fp32 raw rows (20 bytes per detected object: what crosses PCIe in a host-fed loop, offline_main.py:40-57): each
 * value is promoted to fp64 as it is loaded -- exact; the IWR1443's objects are int16 counts scaled by a power of two
 * (ReadDataIWR1443.py:150-170) -- and the arithmetic is mmw_normalize's.  pts stays fp64 (normalize_data's own dtype). */
int mmw_normalize_f32(mmw_ctx *ctx, const float *raw, const int32_t *n_raw, double *pts, int32_t *n_out);

/* TrackBuffer.track(pointcloud, batch) for every scene (Tracking.py:664-703):
 *   pts[S][max_pts][8] fp64 (x,y,z,vx,vy,vz,doppler,peakVal), n_pts[S], dt[S] (= trackbuffer.dt).
 *   A scene with n_pts[s] == 0 is skipped entirely (offline_main.py:56 never calls track() on an empty frame);
 *   n_pts[s] == MMW_EMPTY_FRAME is track() called ON an empty point cloud: tracks are predicted, aged and expired,
 *   _update_all runs, an empty frame enters the ring (what the reference does when a caller does call it).
 * Outputs (dev, each may be NULL):
 *   assoc[S][max_pts]      _calc_dist_fun result: -1 = None, else index into the
 *                          track list as it was BEFORE _maintain_tracks (Tracking.py:530-574)
 *   db_labels[S][ring*max_pts], db_n[S]   sklearn labels of apply_DBscan on the global
 *                          ring (Utils.py:272-278); db_n = -1 when it was not called, MMW_DB_RAISED when
 *                          sklearn's input validation refused the cloud (MMW_E_NONFINITE). */
int mmw_step(mmw_ctx *ctx, const double *pts, const int32_t *n_pts, const double *dt,
             int32_t *assoc, int32_t *db_labels, int32_t *db_n);
/* mmw_step on fp32 rows: pts[S][max_pts][8] float (32 bytes per point, 16-byte aligned), promoted to fp64 in registers as
 * the association kernel loads them -- exact, so for rows that are fp32-representable (a CSV of the reference's logs, the
 * synthetic scenes of bench.py) every output is bit-equal to mmw_step's on the promoted rows.  Half the bytes per frame
 * over PCIe and out of HBM, and no conversion pass in front of the step. */
int mmw_step_f32(mmw_ctx *ctx, const float *pts, const int32_t *n_pts, const double *dt,
                 int32_t *assoc, int32_t *db_labels, int32_t *db_n);
/* mmw_step with host pointers (H2D, step, D2H; sync): mmw_frame_host without the normalisation. */
int mmw_step_host(mmw_ctx *ctx, const double *pts, const int32_t *n_pts, const double *dt,
                  int32_t *assoc, int32_t *db_labels, int32_t *db_n);
/* One frame of every scene from HOST memory in ONE round trip -- the body of the reference's loop (offline_main.py:40-57):
 *   raw != NULL (pts NULL): raw[S][max_pts][5] = (x, y, z, doppler, peakVal), n[S] rows each -> Utils.normalize_data -> the kept
 *                           rows -> TrackBuffer.track; a scene none of whose rows pass the scene filter is skipped
 *                           (offline_main.py:56), as is a scene with n = 0
 *   pts != NULL (raw NULL): pts[S][max_pts][8] normalised rows, n[S] as for mmw_step (0 = skipped, MMW_EMPTY_FRAME = track() on an
 *                           empty cloud)
 * dt[S] = trackbuffer.dt.  Outputs (host, each may be NULL): pts_out[S][max_pts][8] / n_out[S] = normalize_data's rows and
 * their counts (raw form; n_out = n otherwise), assoc[S][max_pts], db_labels[S][ring*max_pts], db_n[S] as mmw_step,
 * n_tracks[S] = len(effective_tracks) after the frame.  Uploads, kernels and read-backs are queued behind one another through
 * pinned staging blocks and the stream is waited for ONCE; the return code is mmw_check's (per-scene errors, first one). */
int mmw_frame_host(mmw_ctx *ctx, const double *raw, const double *pts, const int32_t *n, const double *dt, double *pts_out,
                   int32_t *n_out, int32_t *assoc, int32_t *db_labels, int32_t *db_n, int32_t *n_tracks);

/* The loop body WITH its posture estimate (offline_main.py:53-60: normalize_data, track, estimate_posture) in one round trip,
 * for the one-scene context of the drop-in's TrackBuffer.  mmw_attach_posture hands the context the define_CNN_3D model
 * (train.py:71-106, BatchNormalization folded into the Dense layers; fp32 DEVICE pointers that stay valid and unchanged in place
 * until detached with NULL): the conv kernels in Keras layout (kd,kh,kw,in,out), dense1_w[1536][dense1_ld] = Dense-1 transposed
 * (K = 6144 contiguous, Keras' Flatten order), dense2_w[57][1536].  Needs n_scenes == 1, FB_FRAMES_BATCH == 2 (the 3-frame
 * model), track_cap <= 64.
 * mmw_frame_posture_host = mmw_frame_host, and behind the step on the same stream -- unless the frame was skipped --
 * TrackBuffer.estimate_posture (Tracking.py:705-734): the feature tensors of the tracks with more than MODEL_MIN_INPUT ring
 * points, the CNN in Keras' own fp32 arithmetic (mmw_mars_conv3d's kernel, mmw_mars_head_small's kernels, with a row count
 * only the device knows) and track.keypoints = its rows.  *posture_rows (host, may be NULL) = the tracks estimated.  Still ONE
 * wait for the stream. */
typedef struct mmw_posture_model {
    const float *conv1_w, *conv1_b, *conv2_w, *conv2_b;
    const float *dense1_w;
    int64_t dense1_ld;
    const float *dense1_b, *dense2_w, *dense2_b;
} mmw_posture_model;
int mmw_attach_posture(mmw_ctx *ctx, const mmw_posture_model *model);
int mmw_frame_posture_host(mmw_ctx *ctx, const double *raw, const double *pts, const int32_t *n, const double *dt, double *pts_out,
                           int32_t *n_out, int32_t *assoc, int32_t *db_labels, int32_t *db_n, int32_t *n_tracks, int32_t *posture_rows);

/* Utils.apply_DBscan (Utils.py:250-291) on arbitrary clouds: pts[S][max_n][8], n[S]
 * -> labels[S][max_n], n_clusters[S] (dev pointers; max_n <= ring*max_pts; max_n > 1920: the global-memory path).
 * A cloud that holds a NaN or an infinite value in any of its 8 columns is refused as sklearn's input validation refuses it
 * (ValueError): its labels are not written and n_clusters[s] = -MMW_ERRBIT_NONFINITE_NAN (-16) or -MMW_ERRBIT_NONFINITE_INF (-32). */
int mmw_dbscan(mmw_ctx *ctx, const double *pts, const int32_t *n, int32_t max_n, double eps,
               int32_t min_samples, int32_t *labels, int32_t *n_clusters);

/* Feature side of TrackBuffer.estimate_posture (Tracking.py:718-730) =
 * relative_coordinates + format_single_frame (Utils.py:437-520) for every track of
 * every scene with len(batch.effective_data) > MODEL_MIN_INPUT, compacted in
 * (scene, track) order:
 *   feat[cap_rows][ring][8][8][5] fp32 (dev; [cap_rows][8][8][5] when FB_FRAMES_BATCH == 0)
 *   owner[cap_rows][2] int32 (dev) = (scene, track index)
 *   *n_rows (host) = rows written (sync).  MMW_E_CAPACITY if cap_rows is too small. */
int mmw_features(mmw_ctx *ctx, float *feat, int32_t *owner, int32_t cap_rows, int32_t *n_rows);
/* The same without the host wait, for a caller that pipelines frames (the CNN of frame f on one stream while the
 * tracker of frame f+1 runs on the context's): the rows are written behind the preceding mmw_step on the context's
 * stream, the eligible-track total follows them into pinned host memory, and mmw_features_wait(ticket) waits for THAT
 * copy only (not for the stream).  ticket in [0,4): up to four calls may be outstanding (ticket 3 is the one
 * mmw_features itself uses).  uid[cap_rows] (dev, may be NULL) receives each row's track creation ordinal
 * (mmw_track_record.uid) for mmw_set_keypoints_uid. */
int mmw_features_async(mmw_ctx *ctx, float *feat, int32_t *owner, int32_t *uid, int32_t cap_rows, int32_t ticket);
int mmw_features_wait(mmw_ctx *ctx, int32_t ticket, int32_t *n_rows);   /* waits for that ticket's total only */
/* Utils.relative_coordinates + Utils.format_single_frame (Utils.py:437-520) on caller
 * frames: frames[B][ring][64][8] fp64 (only rows [:64] matter, Utils.py:505-510),
 * counts[B][ring] valid rows (<= 0 rows: frame stays zero, Utils.py:493), ref[B][2] = (x, y)
 * subtracted from columns 0,1 -> feat[B][ring][8][8][5] fp32.  All dev pointers. */
int mmw_format_frames(mmw_ctx *ctx, const double *frames, const int32_t *counts, const double *ref, float *feat, int32_t n_items);
/* track.keypoints = frame_keypoints[i] (Tracking.py:733-734): kp[n_rows][57] fp32 dev. */
int mmw_set_keypoints(mmw_ctx *ctx, const float *kp, const int32_t *owner, int32_t n_rows);

/* The same assignment when later frames have been tracked since mmw_features_async took the rows (list positions are
 * stale then): row i goes to the track of scene owner[i][0] whose creation ordinal is uid[i]; rows of tracks that
 * have expired meanwhile are dropped, as `track.keypoints = ...` on an object no list refers to any more would be. */
int mmw_set_keypoints_uid(mmw_ctx *ctx, const float *kp, const int32_t *owner, const int32_t *uid, int32_t n_rows);

/* Read-back (sync; host pointers).  Also surfaces per-scene errors recorded by
 * the kernels (returns the first one and sets the message). */
int mmw_check(mmw_ctx *ctx);
int mmw_get_num_tracks(mmw_ctx *ctx, int32_t *n_tracks /*[S]*/);
int mmw_get_tracks(mmw_ctx *ctx, mmw_track_record *out /*[S][cap]*/, int32_t cap);
int mmw_get_batch_ring(mmw_ctx *ctx, int32_t *ring_len /*[S]*/, int32_t *ring_n /*[S][MMW_RING_MAX]*/);
/* rows of frame k (oldest first) of track t of scene s: out[ring_rows][8]; *n_rows = stored rows */
int mmw_get_track_ring_frame(mmw_ctx *ctx, int32_t scene, int32_t track, int32_t k, double *out, int32_t *n_rows);
int mmw_get_batch_ring_frame(mmw_ctx *ctx, int32_t scene, int32_t k, double *out /*[max_pts][8]*/, int32_t *n_rows);

/* ClusterTrack.seek_inner_clusters (Tracking.py:409-448) runs inside mmw_step when mmw_config.seek_inner = 1, i.e. as if
 * its call site Tracking.py:656 were active: per-track ring sizes follow change_buffer_size (Tracking.py:60-64), the
 * cluster's cloud is added to the track's ring a second time, apply_DBscan(eps = DB_INNER_EPS) runs on the ring and a
 * track is appended for clusters[1] before _update_all.  Such a context stores ring frames whole (ring_rows is raised
 * to ring * max_pts); a cloud of more than min(1920, ring * ring_rows) rows, or a frame longer than ring_rows, is
 * MMW_E_CAPACITY.  This read-back (sync) returns the calls of the LAST mmw_step: n_calls[S]; rows[S][16] = points
 * clustered by each of the first 16 calls, in track-list order; labels[S][cap_labels] = their sklearn labels back to
 * back (as many as fit).  rows / labels may be NULL. */
int mmw_get_inner(mmw_ctx *ctx, int32_t *n_calls, int32_t *rows, int32_t *labels, int32_t cap_labels);

/* Per-scene track table for the multi-GPU all-gather:
 * table[S][slots] (dev), scene ids offset by scene_base.  Async. */
int mmw_track_table(mmw_ctx *ctx, mmw_track_summary *table, int32_t slots, int32_t scene_base);

/* Kernel timing with hipEvents on the context's stream (bench.py roofline).
 * ids: 0 k_track (association + DBSCAN cell-count screen), 1 k_dbscan_big (BallTree DBSCAN of large clouds), 2 features,
 * 3 normalize, 4 table, 5 k_predict, 6 k_post (Kalman update + BallTree DBSCAN of small clouds). */
#define MMW_K_TRACK 0
#define MMW_K_DBSCAN 1
#define MMW_K_FEATURES 2
#define MMW_K_NORMALIZE 3
#define MMW_K_TABLE 4
#define MMW_K_PREDICT 5
#define MMW_K_POST 6
#define MMW_K_COUNT 7
/* The two Conv3D(3x3x3, same, relu) layers of define_CNN_3D (train.py:73-82) fused on the fp32 matrix
 * cores, on the current device and the given hipStream_t (NULL = default stream).  All dev pointers:
 *   feat[n][3][8][8][5]   channels-last input (what mmw_features writes)
 *   w1[3][3][3][5][16], b1[16], w2[3][3][3][16][32], b2[32]   Keras kernel layout (kd,kh,kw,in,out)
 *   out[n][3][8][8][32]   = Keras Flatten order (d,h,w,c): feed Dense-1 with Keras' weight rows. */
int mmw_mars_conv3d(void *hip_stream, const float *feat, const float *w1, const float *b1, const float *w2, const float *b2,
                    float *out, int32_t n);

/* The two convolution layers on the fp16 matrix cores, fp32-exact by operand splitting: every fp32 value a (inputs,
 * weights, activations) is carried as hi = fp16(a), lo' = fp16((a - hi) * 2^11) and a product a.w accumulates in fp32 as
 * hi.hi + 2^-11 (hi.lo' + lo'.hi): three exact fp16 products per term, the dropped lo'.lo' term 2^-22 relative --
 * measured closer to the fp64 oracle than the fp32 kernel above.  frames = 3: define_CNN_3D's Conv3D pair
 * (train.py:73-82), feat[n][3][8][8][5], kernels (kd,kh,kw,in,out); frames = 1: define_CNN's Conv2D pair
 * (train.py:35-44), feat[n][8][8][5], kernels (kh,kw,in,out).  The activation leaves the kernel already split, for a
 * Dense-1 of the same form: out16[n][ld_out] fp16, ld_out >= 2 * frames * 2048; value j of Keras' Flatten order has its
 * hi half at (j / 32) * 64 + j % 32 and its lo' half 32 further -- runs of [hi 32 | lo' 32], one 128-byte line per
 * 32-deep step of Dense-1.  The split is exact inside fp16's range only: range_flag (device pointer, may be NULL) gets
 * bit 0 set when an input or an activation of magnitude >= 65 504 (or a non-finite input) was split -- that sample's
 * outputs are then meaningless, where Keras' fp32 would have been finite; nothing clears it but the caller. */
int mmw_mars_conv_split(void *hip_stream, int32_t frames, const float *feat, const float *w1, const float *b1, const float *w2,
                        const float *b2, void *out16, int64_t ld_out, int32_t n, int32_t *range_flag, int32_t *sample_flags);
/* ... per sample: sample_flags (device int32[2 + MMW_RANGE_FIXUP_CAP], may be NULL; zero it once) is the fix-up list: [0] counts
 * the samples whose input or activations left fp16's range (running, atomic), [2 ..] are the indices of the first
 * MMW_RANGE_FIXUP_CAP of them, in any order.
 * mmw_mars_range_fixup recomputes exactly those samples in Keras' own fp32 arithmetic and overwrites their rows of kp[n][57] --
 * all on the device and the given stream, no host wait: the listed samples' feature tensors are copied, run through the fp32
 * conv pair (mmw_mars_conv3d's kernel) and the thin Dense-1 / Dense-2 kernels of mmw_mars_head_small with a row count only the
 * device knows (workgroups past it leave at once: a frame without such samples pays four empty launches), and scattered back;
 * the list is emptied for the next call ([1] = the number taken).  More than MMW_RANGE_FIXUP_CAP flagged samples in one call:
 * bit 1 of *range_flag is raised and the surplus keeps its meaningless rows.
 * define_CNN_3D only (feat[n][3][8][8][5]); cw1 / cb1 / cw2 / cb2 = the conv kernels in Keras layout (fp32), w1[1536][ldw] /
 * bias1 / w2[57][1536] / bias2 as for mmw_mars_head_small; scratch = MMW_RANGE_FIXUP_SCRATCH bytes of device memory. */
#define MMW_RANGE_FIXUP_CAP 64
#define MMW_RANGE_FIXUP_SCRATCH (512 + 64 * (960 + 6144 + 1536 + 57) * 4)
int mmw_mars_range_fixup(void *hip_stream, const float *feat, int32_t *sample_flags, int32_t n, const float *cw1, const float *cb1,
                         const float *cw2, const float *cb2, const float *w1, int64_t ldw, const float *bias1, const float *w2, const float *bias2,
                         void *scratch, float *kp, int32_t *range_flag);
/* Dense-1 of the MARS CNN (train.py:49,87: Dense(512 k, relu); BatchNormalization folded in) on split-fp16 operands, one
 * kernel: out = relu(bias + hi . W_hi + 2^-11 (hi . W_lo' + lo' . W_hi)), fp32 accumulation and output (k_dense.hip).
 * a2 [rows_padded][lda] fp16 as mmw_mars_conv_split writes it; w2 [n][ldw] fp16 = the transposed weights (K contiguous)
 * split and interleaved the same way; bias [n], out [rows_padded][n] fp32.  rows_padded is a multiple of 256 (rows past
 * the batch may hold anything: a row only feeds its own output row), k of 32, n of 128; lda, ldw >= 2 k.  hip_stream as above. */
int mmw_mars_dense1_split(void *hip_stream, const void *a2, int64_t lda, const void *w2, int64_t ldw, const float *bias, float *out,
                          int32_t rows_padded, int32_t k, int32_t n);

/* The head of the MARS CNN for a SMALL batch (n_rows <= 64: one scene's tracks, TrackBuffer.estimate_posture of the offline
 * loop): Dense-1 + ReLU (train.py:49,87) and Dense-2 (train.py:54,92), both with their BatchNormalization folded in, in fp32
 * fused multiply-adds -- Keras' own arithmetic.  The weight matrix is cut along the features over the whole chip (a tile kernel
 * would stream it through one band of eight workgroups).  act[n_rows][lda] fp32 = the conv pair's output in Keras' Flatten
 * order (mmw_mars_conv3d); w1[n1][ldw] fp32 = Dense-1's weights transposed (K contiguous), bias1[n1]; w2[57][n1], bias2[57];
 * hidden[n_rows][n1] scratch; kp[n_rows][57].  All dev pointers, 16-byte aligned; k, lda, ldw multiples of 4. */
int mmw_mars_head_small(void *hip_stream, const float *act, int64_t lda, const float *w1, int64_t ldw, const float *bias1, const float *w2,
                        const float *bias2, float *hidden, float *kp, int32_t n_rows, int32_t k, int32_t n1);

/* ReadIWR14xx.read (ReadDataIWR1443.py:27-201) on a byte buffer, host only (no context, no GPU work): the input
 * step before mmw_normalize.  Looks for the LAST 8-byte magic word 02 01 04 03 06 05 08 07 in buf[0 .. len-8),
 * needs more than 16 bytes from there and the whole packet (little-endian u32 totalPacketLen at offset 12).
 * If the header announces objects and the first TLV is MMWDEMO_UART_MSG_DETECTED_POINTS (type 1), the objects
 * (u16 count, u16 Q format, then int16 rangeIdx, dopplerIdx, peakVal, x, y, z each) become raw[n][5] =
 * (x, y, z, doppler, peakVal) -- the row layout mmw_normalize takes -- with x,y,z / 2^Q, doppler =
 * dopplerIdx * doppler_resolution_mps after the reference's wrap of indices above num_doppler_bins/2 - 1
 * (it subtracts 65535, in int16), and range_out[n] = rangeIdx * range_idx_to_meters (may be NULL).
 * Returns 1 (points parsed), 0 (no complete packet, no objects or another TLV first) or MMW_E_ARG;
 * *packet_start / *packet_len (0 when no complete packet) tell the caller what to drop from its buffer. */
typedef struct mmw_uart_cfg {
    double range_idx_to_meters;
    double doppler_resolution_mps;
    int32_t num_doppler_bins;
    int32_t reserved;
} mmw_uart_cfg;
int mmw_parse_uart(const uint8_t *buf, size_t len, const mmw_uart_cfg *cfg, double *raw /*[max_obj][5]*/, double *range_out /*[max_obj]*/,
                   int32_t max_obj, int32_t *n_obj, uint32_t *frame_number, size_t *packet_start, size_t *packet_len);

/* The batched, device-side form of that input step: the host only FINDS the packet, the GPU decodes it.
 * mmw_find_tlv (host, no GPU work) = the packet part of mmw_parse_uart -- last magic word, whole packet present, objects
 * announced, first TLV = detected points, its objects inside the buffer -- without decoding an object: *body_offset = offset
 * from buf of the TLV BODY (u16 numObj, u16 xyzQFormat, numObj x six int16: rangeIdx, dopplerIdx, peakVal, x, y, z; 12 bytes per
 * object, ReadDataIWR1443.py:107-150), -1 if there is none; *n_obj = the count the body announces (the caller checks it
 * against max_pts); returns 1 / 0 / MMW_E_ARG and the packet position as mmw_parse_uart does.
 * mmw_normalize_tlv = ReadIWR14xx.read's decode (ReadDataIWR1443.py:153-171) + Utils.normalize_data (Utils.py:342-434) for
 * every scene in ONE kernel, fused ahead of mmw_step: packets (dev) = the bytes as they arrived, all scenes' packets in one
 * buffer; tlv_offset[S] (dev) = byte offset into `packets` of each scene's TLV body (2-byte aligned), < 0 = no detected-points
 * TLV this frame (n_out = 0: mmw_step skips the scene's frame, offline_main.py:56); cfg (host) as for mmw_parse_uart;
 * pts[S][max_pts][8] / n_out[S] (dev) as mmw_normalize writes them -- bit-equal to mmw_parse_uart + mmw_normalize on the same
 * bytes.  12 bytes per object cross PCIe instead of 20 (fp32 raw rows) or 40 (fp64).  packets_bytes = the size of `packets`:
 * nothing outside it is read.  A body that does not lie inside it on a 2-byte boundary with every object it announces, or that
 * announces more than max_pts objects (mmw_parse_uart returns MMW_E_ARG for those bytes), gives n_out = MMW_BAD_FRAME: the
 * mmw_step that follows raises the scene's bad-count bit (MMW_E_ARG), the other scenes are unaffected. */
#define MMW_BAD_FRAME (-3)
int mmw_find_tlv(const uint8_t *buf, size_t len, int64_t *body_offset, int32_t *n_obj, uint32_t *frame_number, size_t *packet_start,
                 size_t *packet_len);
int mmw_normalize_tlv(mmw_ctx *ctx, const uint8_t *packets, size_t packets_bytes, const int64_t *tlv_offset, const mmw_uart_cfg *cfg, double *pts,
                      int32_t *n_out);

/* Work counters accumulated by the kernels since the last reset (sync):
 * [0] k_track algorithmic bytes  [1] k_dbscan algorithmic bytes  [2] scene-frames stepped
 * [3] apply_DBscan calls  [4] sum of U over those calls  [5] sum of tracks entering track()
 * [6] gate evaluations (points x tracks)  [7] clusters found.  Definitions: DESIGN.md §4. */
int mmw_stats_get(mmw_ctx *ctx, uint64_t *out /*[8]*/);
int mmw_stats_reset(mmw_ctx *ctx);
/* Are the chain workers in use?  0 = no (not configured, seek_inner, or the side streams turned out to share a hardware
 * queue with the context's stream: checked by the first mmw_step after mmw_create / mmw_set_stream /
 * mmw_set_chain_side_stream), 1 = yes, 2 = configured, not checked yet (no step since). */
int mmw_side_workers(mmw_ctx *ctx);
/* Do kernels queued on stream_b run while a kernel on stream_a is still running?  1 = yes, 0 = no: the HIP runtime multiplexes
 * streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default, dealt round-robin at stream creation), and two streams
 * that share one execute in order.  Callers that overlap their own work with the context's (the CNN beside the tracker:
 * posture.PosturePipeline) pick their second stream with this.  Synchronises both streams; ~20 us when they are independent. */
int mmw_streams_concurrent(mmw_ctx *ctx, void *stream_a, void *stream_b);
/* Which kernels the next mmw_step launches for TrackBuffer.track (Tracking.py:683-703): 1 = the one-workgroup step (k_scene: a
 * scene's whole track() in one workgroup, then the DBSCAN worker blocks of k_post; contexts whose scenes are all resident at
 * once, mmw_config.fused_step), 2 = two launches (k_track with _predict_all at its head, k_post), 4 = the bulk kernels
 * (k_predict, k_track, k_post, k_dbscan_big).  The results do not depend on it. */
int mmw_step_kind(mmw_ctx *ctx);
/* How the batched Kalman kernels of the bulk step (k_predict, the update half of k_post) are laid out -- mmw_config.
 * kalman_dense_min_units: 1 = over the TRACKS of the context (four per wave, from the update lists k_track builds), 0 = per
 * scene (also: always with seek_inner, with track_cap > 63 and in the one-workgroup step).  The results do not depend on
 * it; the parity tests assert that the layout they asked for is the one that ran. */
int mmw_kalman_layout(mmw_ctx *ctx);
/* Diagnostic: the queue of scenes whose small-cloud DBSCAN k_track could not rule out (k_dbscan.hip), per step parity p:
 * [8p] pushed, [8p+1] claimed, [8p+2] finished this step; [3] last step whose k_post has begun, [4] waits given up (also
 * reported by mmw_check); [16 + 8p ...] the same three words for the queue of the clouds of more than 256 points.
 * Sync; does not wait for the context's stream. */
int mmw_diag_queue(mmw_ctx *ctx, int32_t *out /*[32]*/);
/* [0..7] as mmw_stats_get; [8..29] per-phase cycle sums, non-zero only in the diagnostic build
 * (make -C mmwave_msc_amd/csrc STAMPS=1), see scripts/phase_stamps.py; [30] k_features algorithmic bytes (ring rows
 * read + fp32 tensors written)  [31] feature tensors written. */
int mmw_stats_get_ext(mmw_ctx *ctx, uint64_t *out /*[32]*/);
/* on = 0: off; 1: every kernel id; otherwise a mask, bit (k + 1) selects kernel id k.  Does not synchronise. */
int mmw_profile_enable(mmw_ctx *ctx, int32_t on);
int mmw_profile_reset(mmw_ctx *ctx);
int mmw_profile_get(mmw_ctx *ctx, int32_t kernel_id, double *total_ms, int64_t *launches);   /* sync */
const char *mmw_kernel_name(int32_t kernel_id);
/* "mmw-hip <version> (gfx950) src:<hash>": <hash> = first 16 hex digits of the SHA-256 over csrc/ and
 * this header at build time (csrc/Makefile); the Python loader refuses a library built from other sources. */
const char *mmw_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MMW_H */
