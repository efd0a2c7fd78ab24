// mmw_api.hip -- the C-ABI (include/mmw.h) over the HIP kernels.  No CPU path:
// if HIP cannot give us a gfx950-class device, creation fails loudly.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mmw_device.hpp"
#include "mmw_launch.hpp"
#include "mmw_kalman.hpp"

namespace mmw {
thread_local LaunchProf g_launch_prof;
size_t track_lds_bytes(const DevCfg &c);
hipError_t prepare_track(const DevCfg &cfg);
void launch_predict(const DevCfg &cfg, const DevState &st, const int32_t *n_pts, const double *dt, int parity, hipStream_t stream);
void launch_post(const DevCfg &cfg, const DevState &st, const int32_t *n_pts, int UM, int u_bound, int parity, int epoch, int32_t *labels,
                 int32_t *db_n, hipStream_t stream);
void launch_chain(const DevCfg &cfg, const DevState &st, int UM, int u_bound, int parity, int epoch, int32_t *labels, int32_t *db_n, hipStream_t side);
void launch_track(const DevCfg &cfg, const DevState &st, const void *pts, bool f32, const int32_t *n_pts, const double *dt,
                  int32_t *assoc, int32_t *db_n, int32_t *db_labels, int UM, int parity, hipStream_t stream);
size_t scene_lds_bytes(const DevCfg &c);
hipError_t prepare_scene(const DevCfg &cfg);
void launch_scene(const DevCfg &cfg, const DevState &st, const void *pts, bool f32, const int32_t *n_pts, const double *dt,
                  int32_t *assoc, int32_t *db_n, int32_t *db_labels, int UM, int parity, hipStream_t stream);
size_t dbscan_lds_bytes(int cls, int UM, int t_cap, int min_samples);
size_t dbscan_only_lds_bytes(int UM);
hipError_t prepare_dbscan(int UM, int t_cap, int min_samples);
void launch_dbscan_big(const DevCfg &cfg, const DevState &st, int UM, int u_bound, int parity, int32_t *labels, int32_t *db_n, hipStream_t stream);
hipError_t prepare_inner(const DevCfg &cfg);
size_t inner_lds_demand(const DevCfg &cfg);
int inner_um(const DevCfg &cfg);
void launch_inner(const DevCfg &cfg, const DevState &st, const int32_t *n_pts, int32_t *db_n, hipStream_t stream);
int dbscan_huge_workers(int n_scenes);
size_t dbscan_huge_slab_bytes(int UM, int t_cap, int min_samples);
void launch_dbscan_huge(const DevCfg &cfg, const DevState &st, int UM, int u_bound, int parity, int32_t *labels, int32_t *db_n, hipStream_t stream);
void launch_dbscan_only(const DevCfg &cfg, const DevState &st, int UM, const double *pts, const int32_t *n, int max_n, double eps, int min_samples,
                        int32_t *labels, int32_t *ncl, hipStream_t stream);
void launch_normalize(const DevCfg &cfg, const void *raw, bool f32, const int32_t *n_raw, double *out, int32_t *n_out, hipStream_t st);
void launch_normalize_tlv(const DevCfg &cfg, const uint8_t *packets, long long packets_bytes, const long long *tlv_offset, double half_bins,
                          double doppler_res, double *out, int32_t *n_out, hipStream_t st);
void launch_feat_scan(const DevCfg &cfg, const DevState &s, int32_t *row_off, hipStream_t st);
void launch_features(const DevCfg &cfg, const DevState &s, const int32_t *row_off, float *feat, int32_t *owner, int32_t *uid, int cap,
                     hipStream_t st, const int32_t *n_in = nullptr, int32_t *total_out = nullptr);
void launch_set_kp_uid(const DevCfg &cfg, const DevState &s, const float *kp, const int32_t *owner, const int32_t *uid, int n_rows,
                       hipStream_t st);
void launch_format_frames(const DevCfg &cfg, const double *frames, const int32_t *counts, const double *ref, float *feat, int B, hipStream_t st);
void launch_set_kp(const DevCfg &cfg, const DevState &s, const float *kp, const int32_t *owner, int n_rows, hipStream_t st,
                   const int32_t *dev_rows = nullptr);
void launch_export(const DevCfg &cfg, const DevState &s, mmw_track_record *out, int cap, hipStream_t st);
void launch_table(const DevCfg &cfg, const DevState &s, mmw_track_summary *out, int slots, int base, hipStream_t st);
void launch_reset(const DevCfg &cfg, const DevState &s, const int32_t *flags, hipStream_t st);
void launch_probe_wait(int32_t *w, int slot, int polls, hipStream_t st);
void launch_probe_set(int32_t *w, hipStream_t st);
void launch_pop_frame(const DevCfg &cfg, const DevState &s, const int32_t *flags, hipStream_t st);
void launch_clear_errors(const DevCfg &cfg, const DevState &s, const int32_t *flags, int bits, hipStream_t st);
void launch_set_batch_size(const DevCfg &cfg, const DevState &s, const int32_t *flags, int new_size, hipStream_t st);
void launch_mars_conv(const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, float *out, int B,
                      hipStream_t stream, const int32_t *dev_rows = nullptr);
void launch_range_gather(const float *feat, int32_t *list, int n, int per, int cap, float *small, int32_t *range_flag, hipStream_t stream);
void launch_range_scatter(const float *kp_small, const int32_t *list, float *kp, int cap, int nout, int n, hipStream_t stream);
int launch_mars_dense1(const void *a2, long long lda, const void *w2, long long ldw, const float *bias, float *out, int rows_padded, int K, int N,
                       hipStream_t stream);
int launch_mars_head_small(const float *act, long long lda, const float *w1, long long ldw, const float *bias1, const float *w2, const float *bias2,
                           float *hidden, float *kp, int n_rows, int K, int N1, int NOUT, hipStream_t stream, const int32_t *dev_rows = nullptr);
int launch_mars_conv16(int nz, const float *feat, const float *w1, const float *b1, const float *w2, const float *b2, void *out16, long long ld_out,
                        int B, int32_t *range_flag, int32_t *sample_flags, hipStream_t stream);
}  // namespace mmw

using namespace mmw;

static thread_local std::string g_last_error;

struct EventPair { hipEvent_t a, b; int kid; };

struct mmw_ctx {
    mmw_config cfg;
    DevCfg dc;
    DevState st;
    int device;
    int UM;                      // ring * max_pts
    hipStream_t own_stream, stream;
    hipStream_t side_stream = nullptr;   // k_chain beside k_track (contexts with dc.side_worker)
    hipEvent_t side_gate = nullptr;      // (gate_side only) recorded on the context's stream at the head of a step: k_chain does not start before it
    int gate_side = 0;                   // mmw_config.chain_side_stream == 3
    int side_wanted = 0;                 // what the configuration / mmw_set_chain_side_stream asked for
    int fused_wanted = 0;                // the one-workgroup step (k_scene) is what this context runs unless a ring was resized or the side workers were asked for
    int side_trusted = 0;                // mmw_config.chain_side_stream == 2: the side stream is used without the concurrency check
    int side_probed = 0;                 // the side streams have been checked against the current context stream (probe_side_streams)
    int32_t *d_probe = nullptr;          // [4] flag + results of that check
    int epoch = 0;                       // step number (queue protocol of list 3, k_dbscan.hip)
    std::string err;
    // internal scratch
    int32_t *d_row_off = nullptr;     // [S+1]
    int32_t *h_rows = nullptr;        // pinned [kTickets]: eligible-track totals of the outstanding mmw_features_async calls
    hipEvent_t feat_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t handoff_ev = nullptr;     // mmw_stream_wait: recorded on the context's stream, waited for by the caller's
    hipEvent_t handback_ev = nullptr;    // mmw_wait_stream: recorded on the caller's stream, waited for by the context's
    int32_t feat_cap[4] = {0, 0, 0, 0};
    float *d_posture = nullptr;
    unsigned long long *d_stats = nullptr;
    int32_t *d_db_list = nullptr, *d_db_count = nullptr, *d_q = nullptr;
    int step_parity = 0;
    int ring_frames_bound = 0;   // no scene's global ring holds more frames than this (host-side knowledge: steps since the last reset)
    // host-convenience staging (lazy): one device block in [rows | dt | n], one out [assoc | db_n | n_out | labels], their pinned
    // host mirrors, and pinned copies of the scene headers and the queue words -- mmw_frame_host moves each with ONE copy
    char *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
    double *d_raw = nullptr;          // [S][max_pts][8]: normalize_data's rows in the raw form of mmw_frame_host (the raw rows arrive in d_in's row area)
    SceneHdr *h_hdr = nullptr;        // pinned [S]
    int32_t *h_q = nullptr;           // pinned [kQWords]
    double *d_pts = nullptr; int32_t *d_n = nullptr; double *d_dt = nullptr;      // (views into d_in / d_out)
    int32_t *d_assoc = nullptr, *d_labels = nullptr, *d_dbn = nullptr, *d_nout = nullptr, *d_prows = nullptr;
    mmw_track_record *d_export = nullptr; int export_cap = 0;
    // mmw_attach_posture: the model and the chain's buffers ([cap] rows: feature tensors, owners, conv output, hidden, keypoints)
    bool has_model = false;
    mmw_posture_model model = {};
    char *d_pchain = nullptr;
    float *pc_feat = nullptr, *pc_act = nullptr, *pc_hidden = nullptr, *pc_kp = nullptr;
    int32_t *pc_owner = nullptr;
    // profiling
    unsigned prof_mask = 0;           // bit k: time kernel id k (mmw_profile_enable)
    std::vector<EventPair> pending;
    std::vector<EventPair> pool;
    double tot_ms[MMW_K_COUNT] = {0};
    int64_t launches[MMW_K_COUNT] = {0};
};

constexpr int kTickets = 4;
// Contexts of more than kPerSceneMaxScenes scenes run the Kalman kernels laid out over tracks and the DBSCAN chain workers on a side
// stream unless told otherwise (mmw_config.kalman_dense_min_units = 0, chain_side_stream = 0).  Round 3 had the workers from 1536
// scenes (every step recorded an event for them: 1024 / 1280 scenes 91 -> 96 us with them) and the per-scene, two-launch step up
// to 768.  Without that event (scripts/side_threshold.sh, scripts/layout_ab.py; ms per step, same box):
//   scenes                              576     640     768     896     1024    1280
//   round-3 choice, frames 20..120      -       0.0822  0.0899  0.0985  0.1108  0.1259
//   track-wise + side stream            -       0.0720  0.0809  0.0878  0.0969  0.1099
//   round-3 choice, frames 10..50       0.0646  0.0646  0.0676  0.0737  -       -
//   track-wise + side stream            0.0581  0.0599  0.0653  0.0740  -       -
// At 512 and below the one-workgroup step / the two-launch step stay ahead in the early window (0.0505 vs 0.0557 at 512).
constexpr int kPerSceneMaxScenes = 512;
constexpr int kSideWorkerMinScenes = kPerSceneMaxScenes + 1;

static int fail(mmw_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_last_error = buf;
    return code;
}

#define HIPCHK(ctx, expr)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(ctx, MMW_E_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static void prof_begin(mmw_ctx *c, int kid, EventPair &ep)
{
    ep.kid = -1;
    if (!((c->prof_mask >> kid) & 1u)) return;
    if (!c->pool.empty()) { ep = c->pool.back(); c->pool.pop_back(); }
    else { hipEventCreate(&ep.a); hipEventCreate(&ep.b); }
    ep.kid = kid;
    hipEventRecord(ep.a, c->stream);
}
// for an id that is exactly ONE launch: the events ride on the kernel's own packet (mmw_launch.hpp)
static void prof_arm(mmw_ctx *c, int kid, EventPair &ep)
{
    ep.kid = -1;
    if (!((c->prof_mask >> kid) & 1u)) return;
    if (!c->pool.empty()) { ep = c->pool.back(); c->pool.pop_back(); }
    else { hipEventCreate(&ep.a); hipEventCreate(&ep.b); }
    ep.kid = kid;
    g_launch_prof.a = ep.a;
    g_launch_prof.b = ep.b;
}
static void prof_armed_done(mmw_ctx *c, EventPair &ep)
{
    if (ep.kid < 0) return;
    if (g_launch_prof.a) {  // nothing was launched (e.g. no large-cloud class exists): nothing to time
        g_launch_prof = LaunchProf{};
        c->pool.push_back(ep);
        return;
    }
    c->pending.push_back(ep);  // (folded by mmw_profile_get / the event-pair path)
}
static void prof_fold(mmw_ctx *c)
{
    for (auto &ep : c->pending) {
        hipEventSynchronize(ep.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) { c->tot_ms[ep.kid] += ms; c->launches[ep.kid]++; }
        c->pool.push_back(ep);
    }
    c->pending.clear();
}
static void prof_end(mmw_ctx *c, EventPair &ep)
{
    if (ep.kid < 0) return;
    hipEventRecord(ep.b, c->stream);
    c->pending.push_back(ep);
    if (c->pending.size() >= 2048) prof_fold(c);
}

static int read_headers(mmw_ctx *c, std::vector<SceneHdr> &h);

// Which step a context runs is decided in ONE place: the one-workgroup step (k_scene) when it was chosen at creation and
// neither the side-stream workers are asked for (they claim scenes while the association kernel runs) nor a global ring has
// been resized (k_track's INNER instantiations read the sizes per ring).  What was ASKED for counts, not what the stream
// probe left of it: a context whose probe turned the workers off keeps the bulk kernels, as mmw_create's `can` has it.
static void refresh_step_kind(mmw_ctx *c) { c->dc.fused = (c->fused_wanted && !c->side_wanted && !c->dc.var_ring) ? 1 : 0; }

extern "C" {

// MMW_SRC_HASH: the first 16 hex digits of the SHA-256 of csrc/*.hip, csrc/*.hpp and include/mmw.h (csrc/Makefile passes it
// when this file is compiled, and this object depends on all of them): mmwave_msc_amd/_lib.py refuses a library whose
// hash is not that of the sources beside it.
#ifndef MMW_SRC_HASH
#define MMW_SRC_HASH "unhashed"
#endif
const char *mmw_version(void) { return "mmw-hip 0.4 (gfx950) src:" MMW_SRC_HASH; }

const char *mmw_kernel_name(int32_t k)
{
    static const char *names[MMW_K_COUNT] = {"k_track", "k_dbscan_big", "k_features", "k_normalize", "k_table", "k_predict", "k_post"};
    return (k >= 0 && k < MMW_K_COUNT) ? names[k] : "?";
}

int mmw_config_default(mmw_config *c)
{
    if (!c) return MMW_E_ARG;
    static const double lim[6] = {0.2, 0.2, 2, 1.2, 1.2, 0.2};
    static const float posture[MMW_NKP] = {
        0.0000f, -0.0007f, -0.0006f, -0.0038f, -0.1820f, -0.2540f, -0.2579f, 0.1830f, 0.2957f, 0.2940f,
        -0.0805f, -0.1141f, -0.1232f, -0.1358f, 0.0796f, 0.1436f, 0.1558f, 0.1720f, -0.0007f, 0.7699f,
        1.0906f, 1.4020f, 1.5513f, 1.2893f, 1.0360f, 0.7994f, 1.2865f, 1.0483f, 0.8117f, 0.7670f,
        0.3428f, 0.0000f, -0.0746f, 0.7713f, 0.3706f, -0.0128f, -0.0796f, 1.3255f, 0.0752f, 0.0533f,
        0.0203f, 0.0000f, 0.0496f, 0.1350f, 0.1303f, 0.0345f, 0.1277f, 0.1050f, 0.0392f, 0.0533f,
        0.0786f, -0.0056f, 0.0346f, -0.0007f, 0.0683f, -0.0082f, 0.0312f};
    memset(c, 0, sizeof(*c));
    c->fb_frames_batch = 2; c->db_min_samples = 35; c->tr_max_tracks = 4; c->kf_enable_est = 0;
    c->model_min_input = 0; c->dim_x = 9; c->ring_rows = 64; c->track_cap = 0; c->kalman_dense_min_units = 0;
    c->seek_inner = 0; c->chain_side_stream = 0; c->fused_step = 0; c->db_points_thres = 40; c->fb_frames_batch_static = 2; c->db_spread_thres = 0.7; c->db_inner_eps = 0.1;
    c->m_x = 0.32; c->m_y = -0.6; c->m_z = 1.3;
    c->v_screen_fade_size_max = 0.3; c->v_screen_fade_size_min = 0.2; c->v_screen_fade_weight = 0.08;
    c->db_z_weight = 0.4; c->db_range_weight = 0.03; c->db_eps = 0.3;
    c->tr_lifetime_dynamic = 3; c->tr_lifetime_static = 7; c->tr_vel_thres = 0.12; c->tr_gate = 4.5;
    c->kf_q_std = 1; c->kf_p_init = 0.1; c->kf_group_disp_est_init = 0.1; c->kf_a_n = 0.9; c->kf_est_pointnum = 10;
    memcpy(c->kf_spread_lim, lim, sizeof(lim));
    c->kf_a_spr = 0.9; c->intensity_mu = 27.0187; c->intensity_std = 70.351; c->s_height = 1.8;
    c->tilt_cos = 0.99619469809174555;   /* cos(radians(-5)) */
    c->tilt_sin = -0.087155742747658166; /* sin(radians(-5)) */
    memcpy(c->default_posture, posture, sizeof(posture));
    return MMW_OK;
}

const char *mmw_last_error(const mmw_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

// The chain workers' stream.  It must not share a hardware queue with the context's stream (the HIP runtime multiplexes
// streams onto a few queues -- GPU_MAX_HW_QUEUES, 4 by default -- round-robin at creation): probe_side_streams checks that
// and re-creates a stream that does.  (A highest-priority stream gets a queue of another pool, but the context's own
// launches then start ~8 us later per step: measured, not used.)
static hipError_t create_side_streams(mmw_ctx *c)
{
    hipError_t e = hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->side_gate, hipEventDisableTiming);
    return e;
}

// 1 = a kernel on the context's stream runs while a kernel on `side` spins, 0 = it does not (shared hardware queue: a worker
// polling for k_track's pushes would keep k_track from starting until its bounded wait runs out), -1 = HIP error.
static int probe_one(mmw_ctx *c, hipStream_t side, hipStream_t other)
{
    const int polls = 1 << 12;   // a few ms at most; ~20 us when the streams are independent
    if (hipMemsetAsync(c->d_probe, 0, 4 * sizeof(int32_t), c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    launch_probe_wait(c->d_probe, 0, polls, side);
    launch_probe_set(c->d_probe, other);
    if (hipStreamSynchronize(side) != hipSuccess || hipStreamSynchronize(other) != hipSuccess) return -1;
    int32_t w[4] = {0, 0, 0, 0};
    if (hipMemcpy(w, c->d_probe, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return w[1] ? 1 : 0;
}
static int probe_side_streams(mmw_ctx *c)
{
    int ok = probe_one(c, c->side_stream, c->stream);
    for (int attempt = 0; ok == 0 && attempt < 6; attempt++) {  // the next stream lands on the next hardware queue
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) return -1;
        hipStreamSynchronize(c->side_stream);
        hipStreamDestroy(c->side_stream);
        c->side_stream = fresh;
        ok = probe_one(c, fresh, c->stream);
    }
    return ok;
}

int mmw_create(const mmw_config *cfg, int32_t n_scenes, int32_t max_pts, int32_t device, mmw_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, MMW_E_ARG, "mmw_create: null argument");
    *out = nullptr;
    if (n_scenes < 1 || max_pts < 1 || max_pts > MMW_MAX_PTS_LIMIT) return fail(nullptr, MMW_E_ARG, "mmw_create: n_scenes=%d max_pts=%d out of range (max_pts <= %d)", n_scenes, max_pts, MMW_MAX_PTS_LIMIT);
    if (cfg->fb_frames_batch < 0 || cfg->fb_frames_batch + 1 > MMW_RING_MAX) return fail(nullptr, MMW_E_ARG, "FB_FRAMES_BATCH must be in [0,%d]", MMW_RING_MAX - 1);
    if (cfg->dim_x != 9 && cfg->dim_x != 6) return fail(nullptr, MMW_E_ARG, "dim_x must be 9 (CONST_ACC_MODEL) or 6 (CONST_VEL_MODEL)");
    // sklearn's parameter validation (DBSCAN._parameter_constraints, cluster/_dbscan.py:330-342: eps in (0, inf), min_samples an
    // integer >= 1): with anything else EVERY apply_DBscan call of the reference raises InvalidParameterError -- refused here
    if (!(cfg->db_eps > 0.0) || cfg->db_min_samples < 1 || (cfg->seek_inner && !(cfg->db_inner_eps > 0.0)))
        return fail(nullptr, MMW_E_ARG, "DB_EPS%s must be > 0 and DB_MIN_SAMPLES_MIN >= 1 (sklearn's DBSCAN refuses anything else)", cfg->seek_inner ? " / DB_INNER_EPS" : "");
    const int ring = cfg->fb_frames_batch + 1;
    // (apply_DBscan has no size limit, Utils.py:250-291; here a cloud is at most the ring: MMW_RING_MAX frames of MMW_MAX_PTS_LIMIT
    //  points.  Up to 1920 points its BallTree lives in the LDS; larger ones -- only contexts with ring * max_pts > 1920 can
    //  see them -- run on slabs in global memory, k_dbscan_huge)
    if (ring * max_pts > MMW_RING_MAX * MMW_MAX_PTS_LIMIT) return fail(nullptr, MMW_E_ARG, "ring*max_pts = %d exceeds %d", ring * max_pts, MMW_RING_MAX * MMW_MAX_PTS_LIMIT);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, MMW_E_NODEVICE, "no HIP device visible: libmmw_hip has no CPU path");
    if (device < 0 || device >= ndev) return fail(nullptr, MMW_E_NODEVICE, "device %d not available (%d visible)", device, ndev);
    mmw_ctx *c = new (std::nothrow) mmw_ctx();
    if (!c) return fail(nullptr, MMW_E_ARG, "out of host memory");
    c->cfg = *cfg;
    c->device = device;
    hipDeviceProp_t prop;
    {
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
        if (e != hipSuccess) { delete c; return fail(nullptr, MMW_E_HIP, "hipSetDevice / hipGetDeviceProperties(%d) -> %s", device, hipGetErrorString(e)); }
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete c; return fail(nullptr, MMW_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName); }
    DevCfg &d = c->dc;
    memset(&d, 0, sizeof(d));
    d.ring = ring; d.db_min_samples = cfg->db_min_samples; d.tr_max_tracks = cfg->tr_max_tracks;
    d.kf_enable_est = cfg->kf_enable_est; d.model_min_input = cfg->model_min_input; d.dx = cfg->dim_x;
    d.ring_rows = cfg->ring_rows < 64 ? 64 : cfg->ring_rows;
    d.seek_inner = cfg->seek_inner ? 1 : 0;
    d.db_points_thres = cfg->db_points_thres; d.fb_frames_batch_static = cfg->fb_frames_batch_static;
    d.db_spread_thres = cfg->db_spread_thres; d.db_inner_eps = cfg->db_inner_eps;
    d.var_ring = 0;
    // the BallTree chain workers beside k_track on a second stream: for contexts large enough that k_track is a long launch
    // (a small context's whole step is shorter than a chain), and not with seek_inner (k_inner may cancel queued scenes)
    d.side_worker = (!d.seek_inner && (cfg->chain_side_stream > 0 || (cfg->chain_side_stream == 0 && n_scenes >= kSideWorkerMinScenes))) ? 1 : 0;
    d.fused = 0;   // (decided below, once the track capacity is known)
    if (d.seek_inner) {
        // seek_inner_clusters clusters whole ring frames, and the first frame of a track it spawns is a cluster of up to
        // ring*max_pts rows: frames are stored whole.  A ring of size 0 would never leave add_frame's loop (Tracking.py:47-48).
        if (cfg->fb_frames_batch < 1 || cfg->fb_frames_batch_static < 1 || cfg->fb_frames_batch_static > ring) {
            delete c;
            return fail(nullptr, MMW_E_ARG, "seek_inner: FB_FRAMES_BATCH and FB_FRAMES_BATCH_STATIC must be in [1, FB_FRAMES_BATCH + 1 = %d]", ring);
        }
        if (d.ring_rows < ring * max_pts) d.ring_rows = ring * max_pts;
    }
    int cap = cfg->track_cap;
    if (cap <= 0) {
        const int ms = cfg->db_min_samples > 0 ? cfg->db_min_samples : 1;
        cap = (cfg->tr_max_tracks > 0 ? cfg->tr_max_tracks - 1 : 0) + (ring * max_pts) / ms + 1;
    }
    if (cap > MMW_TRACK_CAP_LIMIT) cap = MMW_TRACK_CAP_LIMIT;
    if (cap < 1) cap = 1;
    d.t_cap = cap; d.max_pts = max_pts; d.n_scenes = n_scenes;
    // layout of the Kalman kernels (mmw_kalman.hpp: tracks_dense): laid out over tracks when the context holds more
    // four-track waves than this; 0 = the default threshold (one wave per CU x 4), < 0 = always per scene
    d.dense_min_units = cfg->kalman_dense_min_units == 0 ? (n_scenes <= kPerSceneMaxScenes ? 0x7fffffff : 1024)   // (small contexts: per scene, two-launch step)
                                                         : (cfg->kalman_dense_min_units < 0 ? 0x7fffffff : cfg->kalman_dense_min_units - 1);
    if (d.seek_inner) d.dense_min_units = 0x7fffffff;  // k_inner changes a scene's track count between k_track and k_post: per-scene layout
    d.db_z_weight = cfg->db_z_weight; d.db_range_weight = cfg->db_range_weight; d.db_eps = cfg->db_eps;
    d.tr_lifetime_dynamic = cfg->tr_lifetime_dynamic; d.tr_lifetime_static = cfg->tr_lifetime_static;
    d.tr_vel_thres = cfg->tr_vel_thres; d.tr_gate = cfg->tr_gate; d.kf_q_std = cfg->kf_q_std; d.kf_p_init = cfg->kf_p_init;
    d.kf_group_disp_est_init = cfg->kf_group_disp_est_init; d.kf_a_n = cfg->kf_a_n; d.kf_est_pointnum = cfg->kf_est_pointnum;
    for (int i = 0; i < 6; i++) d.kf_spread_lim[i] = cfg->kf_spread_lim[i];
    d.kf_a_spr = cfg->kf_a_spr; d.intensity_mu = cfg->intensity_mu; d.intensity_std = cfg->intensity_std;
    d.s_height = cfg->s_height; d.tilt_cos = cfg->tilt_cos; d.tilt_sin = cfg->tilt_sin;
    d.m_x = cfg->m_x; d.m_y = cfg->m_y; d.m_z = cfg->m_z;
    d.fade_max = cfg->v_screen_fade_size_max; d.fade_min = cfg->v_screen_fade_size_min; d.fade_weight = cfg->v_screen_fade_weight;
    c->UM = ring * max_pts;
    {
        // The one-workgroup step (k_scene.hip): a scene's whole track() in one workgroup, for contexts whose scenes are all
        // resident at once, two workgroups per CU -- there a step is one scene's latency, and one launch boundary less is
        // what pays: measured on one box, 512 scenes x 512 points x 8 tracks 0.0617 -> 0.0588 ms per step.  (Round 3 kept the
        // two-launch step up to 256 scenes -- 0.0448 against 0.0477 ms then; round 4, scripts/fused_small.sh: 64 / 128 / 256
        // scenes x 256 points x 4 tracks 0.0355 / 0.0364 / 0.0414 two-launch against 0.0305 / 0.0313 / 0.0366 ms, 128 / 256
        // scenes x 512 x 8 equal.)  Hence "automatic" = all scenes resident at once, up to kPerSceneMaxScenes.  Not with
        // seek_inner (k_inner sits between association and update), the side-stream workers (they claim scenes while the
        // association kernel runs) or more than 63 tracks per scene (a lane per track in its maintenance step).
        const size_t sl = scene_lds_bytes(d);
        // (more than 512 points per frame: four points per thread, a register budget of one workgroup per CU -- k_scene.hip)
        const int per_cu = (sl <= 80 * 1024 && max_pts <= 512) ? 2 : (sl <= 160 * 1024 ? 1 : 0);
        const bool can = per_cu > 0 && !d.seek_inner && d.t_cap <= 63 && cfg->chain_side_stream <= 0;
        const bool want = cfg->fused_step > 0 || (cfg->fused_step == 0 && cfg->kalman_dense_min_units == 0 && n_scenes <= 256 * per_cu && n_scenes <= kPerSceneMaxScenes);
        c->fused_wanted = d.fused = (can && want) ? 1 : 0;
        if (d.fused) { d.dense_min_units = 0x7fffffff; d.side_worker = 0; }
    }

#ifdef MMW_DIAG_POISON   // (diagnostic build, mmw_launch.hpp: what mmw_create does not initialise reads as NaN / huge, not as whatever was there)
#define MMW_POISON_FRESH(ptr, bytes) (void)hipMemsetAsync((void *)(ptr), 0xFF, (bytes), c->own_stream)
#else
#define MMW_POISON_FRESH(ptr, bytes) ((void)0)
#endif
    // The context's stream FIRST: everything that initialises device memory below is queued on it and waited for before this
    // function returns.  (Up to round 5 the zero fills were plain hipMemset calls -- work on the NULL stream, asynchronous to the
    // host for device memory -- while the context's stream is non-blocking, i.e. not ordered with the null stream: with six
    // processes on the GPU a fill could still be pending when mmw_create returned, and landed on the track records / queue words
    // AFTER the first steps had written them, or on memory mmw_destroy had already freed.  scripts/dual_run.py caught it: about
    // one fresh context in 10^4 under that load; profiles/NOTEBOOK.md round 6.)
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { mmw_destroy(c); return fail(nullptr, MMW_E_HIP, "hipStreamCreate failed"); }
    c->stream = c->own_stream;
#define ALLOC(ptr, bytes)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes));                                                \
        if (e_ != hipSuccess) { int rc = fail(nullptr, MMW_E_HIP, "hipMalloc(%zu) -> %s", (size_t)(bytes), hipGetErrorString(e_)); mmw_destroy(c); return rc; } \
        MMW_POISON_FRESH(ptr, bytes);                                                                       \
    } while (0)
    const size_t S = (size_t)n_scenes;
    ALLOC(c->st.hdr, S * sizeof(SceneHdr));
    ALLOC(c->st.order, S * cap * sizeof(int32_t));
    ALLOC(c->st.trk, S * cap * sizeof(TrackRec));
    ALLOC(c->st.trk_ring, S * cap * (size_t)ring * d.ring_rows * 8 * sizeof(double));
    ALLOC(c->st.g_ring, S * (size_t)ring * max_pts * 8 * sizeof(double));
    ALLOC(c->d_posture, MMW_NKP * sizeof(float));
    ALLOC(c->d_row_off, (S + 1) * sizeof(int32_t));
    ALLOC(c->d_stats, ((size_t)kStatSlots * kStatWords + 256 + 8192) * sizeof(unsigned long long));  // + probe words, workgroup times of the diagnostic build
    ALLOC(c->d_db_list, 4 * S * sizeof(int32_t));
    ALLOC(c->d_db_count, 8 * sizeof(int32_t));
    ALLOC(c->d_q, kQWords * sizeof(int32_t));
    ALLOC(c->d_probe, 4 * sizeof(int32_t));
    ALLOC(c->st.gate_buf, S * cap * kGateRec * sizeof(double));
    ALLOC(c->st.perm, 2 * S * sizeof(int32_t));
    ALLOC(c->st.upd_count, 2 * (size_t)kUpdWords * sizeof(int32_t));
    ALLOC(c->st.upd_list, 2 * (size_t)kUpdShards * upd_region((int)S, cap) * sizeof(int32_t));
    ALLOC(c->st.spc_count, 2 * sizeof(int32_t));
    ALLOC(c->st.spc_list, 4 * S * sizeof(int32_t));
    c->st.inner_buf = nullptr;
    c->st.inner_cap = 0;
    if (d.seek_inner) {
        c->st.inner_cap = 2 * inner_um(d);
        ALLOC(c->st.inner_buf, S * (size_t)(kInnerHdr + c->st.inner_cap) * sizeof(int32_t));
    }
#undef ALLOC
    c->st.default_posture = c->d_posture;
    c->st.stats = c->d_stats;
    c->st.db_list = c->d_db_list;
    c->st.db_count = c->d_db_count;
    c->st.q = c->d_q;
#ifdef MMW_MUTANT_NULL_STREAM_INIT
    // (diagnostic build `make DIAG=nullinit DIAGFLAGS=-DMMW_MUTANT_NULL_STREAM_INIT`, never the product: the round-5 initialisation,
    //  which scripts/dual_run.py must catch under load)
#define MMW_FILL0(ptr, bytes) hipMemset((ptr), 0, (bytes))
#else
#define MMW_FILL0(ptr, bytes) hipMemsetAsync((ptr), 0, (bytes), c->own_stream)
#endif
    if (hipMemcpyAsync(c->d_posture, cfg->default_posture, MMW_NKP * sizeof(float), hipMemcpyHostToDevice, c->own_stream) != hipSuccess ||
        MMW_FILL0(c->d_stats, (size_t)kStatSlots * kStatWords * sizeof(unsigned long long)) != hipSuccess ||
        MMW_FILL0(c->d_db_count, 8 * sizeof(int32_t)) != hipSuccess ||
        MMW_FILL0(c->d_q, kQWords * sizeof(int32_t)) != hipSuccess || MMW_FILL0(c->d_db_list, 4 * S * sizeof(int32_t)) != hipSuccess ||
        MMW_FILL0(c->st.upd_count, 2 * (size_t)kUpdWords * sizeof(int32_t)) != hipSuccess ||
        MMW_FILL0(c->st.upd_list, 2 * (size_t)kUpdShards * upd_region((int)S, cap) * sizeof(int32_t)) != hipSuccess ||
        MMW_FILL0(c->st.spc_count, 2 * sizeof(int32_t)) != hipSuccess ||
        MMW_FILL0(c->st.trk, S * cap * sizeof(TrackRec)) != hipSuccess ||
        hipStreamSynchronize(c->own_stream) != hipSuccess) { mmw_destroy(c); return fail(nullptr, MMW_E_HIP, "device init failed"); }   // (cfg->default_posture is the caller's)
    if (hipHostMalloc((void **)&c->h_rows, kTickets * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) { mmw_destroy(c); return fail(nullptr, MMW_E_HIP, "hipHostMalloc failed"); }
    for (int k = 0; k < kTickets; k++) {
        c->h_rows[k] = 0;
        if (hipEventCreateWithFlags(&c->feat_ev[k], hipEventDisableTiming) != hipSuccess) { mmw_destroy(c); return fail(nullptr, MMW_E_HIP, "hipEventCreate failed"); }
    }
    c->side_wanted = d.side_worker;
    c->side_trusted = (d.side_worker && cfg->chain_side_stream == 2) ? 1 : 0;   // 2: taken on trust (counter collection serialises kernels: the probe would say no)
    c->gate_side = (d.side_worker && cfg->chain_side_stream == 3) ? 1 : 0;
    c->side_probed = c->side_trusted;
    if (d.side_worker && create_side_streams(c) != hipSuccess) { mmw_destroy(c); return fail(nullptr, MMW_E_HIP, "hipStreamCreate failed"); }
    size_t lds_b = dbscan_only_lds_bytes(c->UM);
    for (int k = 0; k < 3; k++) { const size_t v = dbscan_lds_bytes(k, c->UM, cap, cfg->db_min_samples); if (v > lds_b) lds_b = v; }
    const size_t lds_a = track_lds_bytes(d);
    if (lds_a > 160 * 1024 || lds_b > 160 * 1024) { mmw_destroy(c); return fail(nullptr, MMW_E_ARG, "LDS demand too large (track %zu B, dbscan %zu B > 160 KiB)", lds_a, lds_b); }
    if (d.seek_inner) {
        if (inner_lds_demand(d) > 160 * 1024) { mmw_destroy(c); return fail(nullptr, MMW_E_ARG, "seek_inner: LDS demand too large (%zu B)", inner_lds_demand(d)); }
        if (prepare_inner(d) != hipSuccess) { mmw_destroy(c); return fail(nullptr, MMW_E_HIP, "hipFuncSetAttribute(k_inner) failed"); }
        if (MMW_FILL0(c->st.inner_buf, S * (size_t)(kInnerHdr + c->st.inner_cap) * sizeof(int32_t)) != hipSuccess) { mmw_destroy(c); return fail(nullptr, MMW_E_HIP, "device init failed"); }
    }
    hipError_t e1 = prepare_track(d), e2 = prepare_dbscan(c->UM, cap, cfg->db_min_samples);
    if (e1 == hipSuccess && scene_lds_bytes(d) <= 160 * 1024) e1 = prepare_scene(d);
    if (e1 != hipSuccess || e2 != hipSuccess) { mmw_destroy(c); return fail(nullptr, MMW_E_HIP, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); }
    c->st.huge_scratch = nullptr;
    c->st.huge_stride = dbscan_huge_slab_bytes(c->UM, cap, cfg->db_min_samples);
    if (c->st.huge_stride) {   // a ring of this context can hold a cloud the LDS cannot: one BallTree slab per worker of k_dbscan_huge
        if (hipMalloc((void **)&c->st.huge_scratch, c->st.huge_stride * (size_t)dbscan_huge_workers(n_scenes)) != hipSuccess) {
            mmw_destroy(c);
            return fail(nullptr, MMW_E_HIP, "hipMalloc(%zu B of BallTree slabs) failed", c->st.huge_stride * (size_t)dbscan_huge_workers(n_scenes));
        }
    }
#undef MMW_FILL0
    launch_reset(d, c->st, nullptr, c->stream);
    // every fill above and the reset kernel have finished before the caller sees the context
    if (hipStreamSynchronize(c->stream) != hipSuccess) { mmw_destroy(c); return fail(nullptr, MMW_E_HIP, "reset kernel failed: %s", hipGetErrorString(hipGetLastError())); }
    *out = c;
    return MMW_OK;
}

// Read-back into the caller's (pageable) memory of what the context's stream has written: the stream is waited for first, then the
// copy runs on it alone and is waited for.  (On the context's stream, not as a blocking hipMemcpy: that one runs on the legacy default
// stream and would also wait for whatever another runtime -- torch -- has queued there.)
static int d2h_after_kernels(mmw_ctx *c, void *dst, const void *src, size_t bytes)
{
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MMW_OK;
}

int mmw_destroy(mmw_ctx *c)
{
    if (!c) return MMW_OK;
    hipSetDevice(c->device);
    // nothing of this context may still be running when its memory goes: the context's stream (the caller's or our own) and
    // the chain workers' side stream, which can poll the queues for a few ms after the last step
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->own_stream && c->own_stream != c->stream) hipStreamSynchronize(c->own_stream);
    if (c->side_stream) hipStreamSynchronize(c->side_stream);
    prof_fold(c);
    for (auto &ep : c->pool) { hipEventDestroy(ep.a); hipEventDestroy(ep.b); }
    void *ptrs[] = {c->st.hdr, c->st.order, c->st.trk, c->st.trk_ring, c->st.g_ring, c->d_posture, c->d_row_off, c->d_stats, c->d_db_list, c->d_db_count, c->d_q, c->d_probe, c->st.gate_buf, c->st.perm, c->st.upd_count, c->st.upd_list, c->st.spc_count, c->st.spc_list, c->st.inner_buf, c->d_in, c->d_out, c->d_raw, c->d_pchain,
                    c->d_export, c->st.huge_scratch};
    for (void *p : ptrs) if (p) hipFree(p);
    void *pinned[] = {c->h_in, c->h_out, c->h_hdr, c->h_q};
    for (void *p : pinned) if (p) hipHostFree(p);
    for (int k = 0; k < kTickets; k++) if (c->feat_ev[k]) hipEventDestroy(c->feat_ev[k]);
    if (c->handoff_ev) hipEventDestroy(c->handoff_ev);
    if (c->handback_ev) hipEventDestroy(c->handback_ev);
    if (c->h_rows) hipHostFree(c->h_rows);
    if (c->side_stream) hipStreamDestroy(c->side_stream);
    if (c->side_gate) hipEventDestroy(c->side_gate);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
    return MMW_OK;
}

int mmw_reset(mmw_ctx *c)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    launch_reset(c->dc, c->st, nullptr, c->stream);
    HIPCHK(c, hipGetLastError());
    c->dc.var_ring = 0;   // fresh BatchedData objects: default sizes again
    refresh_step_kind(c);
    c->ring_frames_bound = 0;
    return MMW_OK;
}

int mmw_reset_scenes(mmw_ctx *c, const int32_t *scene_flags)
{
    if (!c || !scene_flags) return fail(c, MMW_E_ARG, "mmw_reset_scenes: null argument");
    HIPCHK(c, hipSetDevice(c->device));
    // (staged in the feature-offset scratch: S + 1 words, not live between calls)
    HIPCHK(c, hipMemcpyAsync(c->d_row_off, scene_flags, sizeof(int32_t) * c->dc.n_scenes, hipMemcpyHostToDevice, c->stream));
    launch_reset(c->dc, c->st, c->d_row_off, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the caller's array may go away
    return MMW_OK;
}

int mmw_get_errors(mmw_ctx *c, int32_t *err_bits)
{
    if (!c || !err_bits) return MMW_E_ARG;
    std::vector<SceneHdr> h;
    int rc = read_headers(c, h);
    if (rc) return rc;
    for (size_t s = 0; s < h.size(); s++) err_bits[s] = h[s].err;
    return MMW_OK;
}

int mmw_clear_errors(mmw_ctx *c, const int32_t *scene_flags, int32_t bits)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int32_t *d_flags = nullptr;
    if (scene_flags) {  // (staged in the feature-offset scratch: S + 1 words, not live between calls)
        HIPCHK(c, hipMemcpyAsync(c->d_row_off, scene_flags, sizeof(int32_t) * c->dc.n_scenes, hipMemcpyHostToDevice, c->stream));
        d_flags = c->d_row_off;
    }
    launch_clear_errors(c->dc, c->st, d_flags, bits, c->stream);
    HIPCHK(c, hipGetLastError());
    if (scene_flags) HIPCHK(c, hipStreamSynchronize(c->stream));  // the caller's array may go away
    return MMW_OK;
}

int mmw_pop_frame(mmw_ctx *c, const int32_t *scene_flags)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int32_t *d_flags = nullptr;
    if (scene_flags) {  // (staged in the feature-offset scratch: S + 1 words, not live between calls)
        HIPCHK(c, hipMemcpyAsync(c->d_row_off, scene_flags, sizeof(int32_t) * c->dc.n_scenes, hipMemcpyHostToDevice, c->stream));
        d_flags = c->d_row_off;
    }
    launch_pop_frame(c->dc, c->st, d_flags, c->stream);
    HIPCHK(c, hipGetLastError());
    if (scene_flags) HIPCHK(c, hipStreamSynchronize(c->stream));  // the caller's array may go away
    return MMW_OK;
}

int mmw_set_batch_size(mmw_ctx *c, const int32_t *scene_flags, int32_t new_size)
{
    if (!c) return MMW_E_ARG;
    // a deque(maxlen = FB_FRAMES_BATCH + 1) never holds more than that whatever `size` says; size <= 0 would never leave
    // add_frame's `while len(buffer) >= size: pop_frame()` in the reference
    if (new_size < 1) return fail(c, MMW_E_ARG, "mmw_set_batch_size: new_size = %d (BatchedData.add_frame would not terminate)", new_size);
    if (new_size > c->dc.ring) new_size = c->dc.ring;
    HIPCHK(c, hipSetDevice(c->device));
    const int32_t *d_flags = nullptr;
    if (scene_flags) {
        HIPCHK(c, hipMemcpyAsync(c->d_row_off, scene_flags, sizeof(int32_t) * c->dc.n_scenes, hipMemcpyHostToDevice, c->stream));
        d_flags = c->d_row_off;
    }
    launch_set_batch_size(c->dc, c->st, d_flags, new_size, c->stream);
    HIPCHK(c, hipGetLastError());
    if (scene_flags) HIPCHK(c, hipStreamSynchronize(c->stream));
    c->dc.var_ring = 1;   // resized rings are k_track's (its INNER instantiations read the sizes per ring); the state layout is the same
    refresh_step_kind(c);
    return MMW_OK;
}

int mmw_set_batch_frame(mmw_ctx *c, int32_t scene, const double *rows, int32_t n)
{
    if (!c || scene < 0 || scene >= c->dc.n_scenes || n < 0 || n > c->dc.max_pts || (n > 0 && !rows)) return fail(c, MMW_E_ARG, "mmw_set_batch_frame: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    SceneHdr h;
    { const int rc_ = d2h_after_kernels(c, &h, c->st.hdr + scene, sizeof(h)); if (rc_) return rc_; }
    h.g_len = 1;
    if (c->ring_frames_bound < 1) c->ring_frames_bound = 1;
    for (int k = 0; k < MMW_RING_MAX; k++) h.g_n[k] = 0;
    h.g_n[0] = n;
    {   // the ring's non-finite flags (SceneHdr.skipped bits 16..23): this frame's, for the slot it is written to
        int bits = 0;
        for (size_t i = 0; i < (size_t)n * 8; i++) bits |= std::isnan(rows[i]) ? 1 : (std::isinf(rows[i]) ? 2 : 0);
        const int nff = nf_flags_with((h.skipped >> kSkipNfShift) & kSkipNfMask, h.g_slot[0], bits);
        h.skipped = (h.skipped & ~(kSkipNfMask << kSkipNfShift)) | (nff << kSkipNfShift);
    }
    double *dst = c->st.g_ring + ((size_t)scene * c->dc.ring + h.g_slot[0]) * (size_t)c->dc.max_pts * 8;
    if (n > 0) HIPCHK(c, hipMemcpyAsync(dst, rows, (size_t)n * 8 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->st.hdr + scene, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MMW_OK;
}

int mmw_set_chain_side_stream(mmw_ctx *c, int32_t on)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (on && c->dc.seek_inner) return fail(c, MMW_E_ARG, "mmw_set_chain_side_stream: not with seek_inner (k_inner may cancel queued scenes)");
    if (on && !c->side_stream) HIPCHK(c, create_side_streams(c));
    c->dc.side_worker = c->side_wanted = on ? 1 : 0;   // takes effect with the next mmw_step (the queues are empty between steps)
    refresh_step_kind(c);   // the workers claim scenes while k_track runs: the bulk kernels' step
    c->side_probed = c->side_trusted;
    return MMW_OK;
}

int mmw_set_stream(mmw_ctx *c, void *s)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStreamSynchronize(c->stream);
    c->stream = s ? (hipStream_t)s : c->own_stream;
    c->dc.side_worker = c->side_wanted;   // (checked against the new stream by the next mmw_step)
    c->side_probed = c->side_trusted;
    refresh_step_kind(c);
    return MMW_OK;
}

int mmw_synchronize(mmw_ctx *c)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MMW_OK;
}

int mmw_stream_wait(mmw_ctx *c, void *hip_stream)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t other = hip_stream == MMW_STREAM_LEGACY ? hipStreamLegacy : (hipStream_t)hip_stream;   // (NULL is the legacy default stream already)
    if (other == c->stream) return MMW_OK;   // one stream: ordered as it is
    if (!c->handoff_ev) HIPCHK(c, hipEventCreateWithFlags(&c->handoff_ev, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->handoff_ev, c->stream));
    HIPCHK(c, hipStreamWaitEvent(other, c->handoff_ev, 0));
    return MMW_OK;
}

int mmw_wait_stream(mmw_ctx *c, void *hip_stream)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t other = hip_stream == MMW_STREAM_LEGACY ? hipStreamLegacy : (hipStream_t)hip_stream;
    if (other == c->stream) return MMW_OK;
    if (!c->handback_ev) HIPCHK(c, hipEventCreateWithFlags(&c->handback_ev, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->handback_ev, other));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->handback_ev, 0));
    return MMW_OK;
}

int mmw_get_dims(const mmw_ctx *c, int32_t *n_scenes, int32_t *max_pts, int32_t *track_cap, int32_t *ring, int32_t *ring_rows)
{
    if (!c) return MMW_E_ARG;
    if (n_scenes) *n_scenes = c->dc.n_scenes;
    if (max_pts) *max_pts = c->dc.max_pts;
    if (track_cap) *track_cap = c->dc.t_cap;
    if (ring) *ring = c->dc.ring;
    if (ring_rows) *ring_rows = c->dc.ring_rows;
    return MMW_OK;
}

int mmw_dev_alloc(mmw_ctx *c, size_t bytes, void **dptr)
{
    if (!c || !dptr) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(dptr, bytes ? bytes : 8));
    return MMW_OK;
}
int mmw_dev_free(mmw_ctx *c, void *p)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (p) HIPCHK(c, hipFree(p));
    return MMW_OK;
}
int mmw_memcpy_h2d(mmw_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MMW_OK;
}
int mmw_memcpy_d2h(mmw_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    { const int rc_ = d2h_after_kernels(c, dst, src, bytes); if (rc_) return rc_; }
    return MMW_OK;
}

static int normalize_impl(mmw_ctx *c, const void *raw, bool f32, const int32_t *n_raw, double *pts, int32_t *n_out)
{
    if (!c || !raw || !n_raw || !pts || !n_out) return fail(c, MMW_E_ARG, "mmw_normalize: null pointer");
    if (((uintptr_t)pts & 15) != 0) return fail(c, MMW_E_ARG, "mmw_normalize: pts must be 16-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    EventPair ep;
    prof_arm(c, MMW_K_NORMALIZE, ep);
    launch_normalize(c->dc, raw, f32, n_raw, pts, n_out, c->stream);
    prof_armed_done(c, ep);
    HIPCHK(c, hipGetLastError());
    return MMW_OK;
}
int mmw_normalize(mmw_ctx *c, const double *raw, const int32_t *n_raw, double *pts, int32_t *n_out) { return normalize_impl(c, raw, false, n_raw, pts, n_out); }
int mmw_normalize_f32(mmw_ctx *c, const float *raw, const int32_t *n_raw, double *pts, int32_t *n_out) { return normalize_impl(c, raw, true, n_raw, pts, n_out); }

int mmw_normalize_tlv(mmw_ctx *c, const uint8_t *packets, size_t packets_bytes, const int64_t *tlv_offset, const mmw_uart_cfg *cfg, double *pts,
                      int32_t *n_out)
{
    if (!c || !packets || !tlv_offset || !cfg || !pts || !n_out) return fail(c, MMW_E_ARG, "mmw_normalize_tlv: null pointer");
    if (((uintptr_t)pts & 15) != 0 || ((uintptr_t)packets & 1) != 0) return fail(c, MMW_E_ARG, "mmw_normalize_tlv: pts must be 16-byte aligned, packets 2-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    EventPair ep;
    prof_arm(c, MMW_K_NORMALIZE, ep);
    static_assert(sizeof(long long) == sizeof(int64_t), "tlv offsets");
    launch_normalize_tlv(c->dc, packets, (long long)packets_bytes, reinterpret_cast<const long long *>(tlv_offset), cfg->num_doppler_bins / 2.0 - 1,
                         cfg->doppler_resolution_mps, pts, n_out, c->stream);
    prof_armed_done(c, ep);
    HIPCHK(c, hipGetLastError());
    return MMW_OK;
}

static int step_impl(mmw_ctx *c, const void *pts, bool f32, const int32_t *n_pts, const double *dt, int32_t *assoc, int32_t *db_labels, int32_t *db_n);
int mmw_step(mmw_ctx *c, const double *pts, const int32_t *n_pts, const double *dt, int32_t *assoc, int32_t *db_labels, int32_t *db_n)
{
    return step_impl(c, pts, false, n_pts, dt, assoc, db_labels, db_n);
}
int mmw_step_f32(mmw_ctx *c, const float *pts, const int32_t *n_pts, const double *dt, int32_t *assoc, int32_t *db_labels, int32_t *db_n)
{
    return step_impl(c, pts, true, n_pts, dt, assoc, db_labels, db_n);
}

static int step_impl(mmw_ctx *c, const void *pts, bool f32, const int32_t *n_pts, const double *dt, int32_t *assoc, int32_t *db_labels, int32_t *db_n)
{
    if (!c || !pts || !n_pts || !dt) return fail(c, MMW_E_ARG, "mmw_step: null input pointer");
    if (((uintptr_t)pts & 15) != 0) return fail(c, MMW_E_ARG, "mmw_step: pts must be 16-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    EventPair ep;
    // a cloud is the unassigned part of the ring's frames: in the first steps after a reset it cannot be larger than the
    // frames pushed so far, and the large-cloud launches are carved (LDS per workgroup -> workgroups per CU) for that bound
    if (c->ring_frames_bound < c->dc.ring) c->ring_frames_bound++;
    const int u_bound = c->ring_frames_bound * c->dc.max_pts;
    c->dc.big_live = (c->dc.side_worker && c->ring_frames_bound >= c->dc.ring) ? 1 : 0;  // (set again below if the probe turns the workers off)
    // the chain workers of this step wait on the side stream for what k_track queues.  Nothing orders them with the context's
    // stream but the queue protocol itself: they only touch scenes k_track has published, and they claim only while the stop
    // epoch says that the step in flight is THEIR step (k_chain, k_dbscan.hip) -- a launch that runs early (steps queued ahead of
    // a stalled context stream) polls and idles out, one that runs late leaves at once.
    if (c->dc.side_worker && !c->side_probed) {
        // first step on this stream set-up: the workers are only used if they really run BESIDE the context's stream
        const int ok = probe_side_streams(c);
        if (ok < 0) return fail(c, MMW_E_HIP, "side-stream probe failed: %s", hipGetErrorString(hipGetLastError()));
        c->side_probed = 1;
        if (!ok) c->dc.side_worker = c->dc.big_live = 0;
    }
    if (c->dc.side_worker) {
        // No event between the two streams: the side stream paces itself -- the workers of step f leave when k_post(f) has raised
        // its stop epoch, the workers of step f + 1 start behind them and find empty queues until k_track(f + 1) pushes (idle polls
        // with s_sleep: twelve workgroups, ~70 us early in a back-to-back loop).  The event recorded at the head of every step (so
        // that they would not start early) was a marker packet on the context's stream: 7 us of idle chip per step in the
        // kernel trace (profiles/NOTEBOOK.md, round 4).  gate_side = 1 (callers with their own work on the context's stream,
        // mmw_config.chain_side_stream = 3) keeps the event.
        if (c->gate_side) {
            HIPCHK(c, hipEventRecord(c->side_gate, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->side_stream, c->side_gate, 0));
        }
    }
    // the step number of the queue protocol counts COMMITTED steps: a step that left above (probe, gate event) has launched no
    // k_post, so no stop epoch was raised for it -- counting it would leave q[kQStop] one behind for good, and every later k_chain
    // would poll to its idle limit without ever claiming
    c->epoch++;
    c->dc.epoch = c->epoch;   // (k_track / k_scene tag the claim words of the NEXT step's queues with it: mmw_device.hpp, q_tag)
    if (c->dc.side_worker) {
        launch_chain(c->dc, c->st, c->UM, u_bound, c->step_parity, c->epoch, db_labels, db_n, c->side_stream);
    }
    // TrackBuffer.track (Tracking.py:683-703) = four launches on one stream:
    prof_arm(c, MMW_K_PREDICT, ep);
    launch_predict(c->dc, c->st, n_pts, dt, c->step_parity, c->stream);
    prof_armed_done(c, ep);
    prof_arm(c, MMW_K_TRACK, ep);
    if (c->dc.fused) launch_scene(c->dc, c->st, pts, f32, n_pts, dt, assoc, db_n, db_labels, c->UM, c->step_parity, c->stream);
    else launch_track(c->dc, c->st, pts, f32, n_pts, dt, assoc, db_n, db_labels, c->UM, c->step_parity, c->stream);
    prof_armed_done(c, ep);
    if (c->dc.seek_inner) launch_inner(c->dc, c->st, n_pts, db_n, c->stream);  // Tracking.py:656 active
    prof_arm(c, MMW_K_POST, ep);
    launch_post(c->dc, c->st, n_pts, c->UM, u_bound, c->step_parity, c->epoch, db_labels, db_n, c->stream);
    prof_armed_done(c, ep);
    prof_arm(c, MMW_K_DBSCAN, ep);
    launch_dbscan_big(c->dc, c->st, c->UM, u_bound, c->step_parity, db_labels, db_n, c->stream);
    launch_dbscan_huge(c->dc, c->st, c->UM, u_bound, c->step_parity, db_labels, db_n, c->stream);  // (contexts with ring * max_pts > 1920 only)
    prof_armed_done(c, ep);
    if (c->pending.size() >= 2048) prof_fold(c);
    c->step_parity ^= 1;
    HIPCHK(c, hipGetLastError());
    return MMW_OK;
}

// layout of the two staging blocks (bytes; every part 16-byte aligned)
struct StageLayout { size_t in_rows, in_dt, in_n, in_bytes, out_assoc, out_dbn, out_nout, out_prows, out_labels, out_bytes; };
static StageLayout stage_layout(const mmw_ctx *c)
{
    const size_t S = c->dc.n_scenes, NP = c->dc.max_pts;
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    StageLayout L;
    L.in_dt = 0;   // [dt | n | rows]: the two small arrays in front, so that a frame's upload is ONE copy that ends with its last valid row
    L.in_n = al(S * sizeof(double));
    L.in_rows = L.in_n + al(S * sizeof(int32_t));
    L.in_bytes = L.in_rows + al(S * NP * 8 * sizeof(double));
    L.out_assoc = 0;
    L.out_dbn = al(S * NP * sizeof(int32_t));
    L.out_nout = L.out_dbn + al(S * sizeof(int32_t));
    L.out_prows = L.out_nout + al(S * sizeof(int32_t));
    L.out_labels = L.out_prows + 16;
    L.out_bytes = L.out_labels + al(S * (size_t)c->UM * sizeof(int32_t));
    return L;
}

static int ensure_host_staging(mmw_ctx *c)
{
    if (c->d_in) return MMW_OK;
    const StageLayout L = stage_layout(c);
    const size_t S = c->dc.n_scenes, NP = c->dc.max_pts;
    HIPCHK(c, hipMalloc((void **)&c->d_in, L.in_bytes));
    HIPCHK(c, hipMalloc((void **)&c->d_out, L.out_bytes));
    HIPCHK(c, hipMalloc((void **)&c->d_raw, S * NP * 8 * sizeof(double)));   // (the raw form: normalize_data's rows, what the step reads)
    HIPCHK(c, hipHostMalloc((void **)&c->h_in, L.in_bytes, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&c->h_out, L.out_bytes, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&c->h_hdr, S * sizeof(SceneHdr), hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&c->h_q, kQWords * sizeof(int32_t), hipHostMallocDefault));
    c->d_pts = reinterpret_cast<double *>(c->d_in + L.in_rows);
    c->d_dt = reinterpret_cast<double *>(c->d_in + L.in_dt);
    c->d_n = reinterpret_cast<int32_t *>(c->d_in + L.in_n);
    c->d_assoc = reinterpret_cast<int32_t *>(c->d_out + L.out_assoc);
    c->d_dbn = reinterpret_cast<int32_t *>(c->d_out + L.out_dbn);
    c->d_nout = reinterpret_cast<int32_t *>(c->d_out + L.out_nout);
    c->d_prows = reinterpret_cast<int32_t *>(c->d_out + L.out_prows);
    c->d_labels = reinterpret_cast<int32_t *>(c->d_out + L.out_labels);
    return MMW_OK;
}

static int first_scene_error(mmw_ctx *c, const SceneHdr *h, size_t n, const int32_t *q);

// One frame of every scene from host memory in ONE round trip: the inputs leave as one copy from a pinned block, the kernels
// follow, the results, the scene headers (track counts, error bits) and the queue words come back as three copies into pinned
// memory, and the stream is waited for once.  (mmw_step_host used to wait four times: the step, mmw_check's two read-backs.)
static int frame_impl(mmw_ctx *c, const double *raw, const double *pts, const int32_t *n, const double *dt, double *pts_out, int32_t *n_out,
                      int32_t *assoc, int32_t *db_labels, int32_t *db_n, int32_t *n_tracks, bool posture, int32_t *posture_rows)
{
    if (!c || !n || !dt || (!raw && !pts) || (raw && pts)) return fail(c, MMW_E_ARG, "mmw_frame_host: exactly one of raw / pts, and n, dt");
    if (posture && !c->has_model) return fail(c, MMW_E_ARG, "mmw_frame_posture_host: no model attached (mmw_attach_posture)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_host_staging(c);
    if (rc) return rc;
    const StageLayout L = stage_layout(c);
    const size_t S = c->dc.n_scenes, NP = c->dc.max_pts, UM = (size_t)c->UM;
    // rows: only the scenes' valid rows travel inside their slots (the slots are max_pts apart); small frames stay small
    const size_t row_doubles = raw ? 5 : 8;
    double *h_rows = reinterpret_cast<double *>(c->h_in + L.in_rows);
    size_t rows_span = 0;   // bytes of the row area that must be sent: up to the end of the last scene's valid rows
    for (size_t s = 0; s < S; s++) {
        const int cnt = n[s];
        if (cnt > 0 && (size_t)cnt <= NP) {
            memcpy(h_rows + s * NP * row_doubles, (raw ? raw : pts) + s * NP * row_doubles, (size_t)cnt * row_doubles * sizeof(double));
            rows_span = (s * NP + (size_t)cnt) * row_doubles * sizeof(double);
        }
    }
    memcpy(c->h_in + L.in_dt, dt, S * sizeof(double));
    memcpy(c->h_in + L.in_n, n, S * sizeof(int32_t));
    HIPCHK(c, hipMemcpyAsync(c->d_in, c->h_in, L.in_rows + rows_span, hipMemcpyHostToDevice, c->stream));   // [dt | n | rows up to the last valid one]
    double *d_rows = c->d_pts;   // what the step reads
    if (raw) {
        d_rows = c->d_raw;
        rc = mmw_normalize(c, c->d_pts, c->d_n, d_rows, c->d_nout);   // (the raw rows sit in the row area of the upload block)
        if (rc) return rc;
        rc = mmw_step(c, d_rows, c->d_nout, c->d_dt, c->d_assoc, c->d_labels, c->d_dbn);
    } else {
        rc = mmw_step(c, d_rows, c->d_n, c->d_dt, c->d_assoc, c->d_labels, c->d_dbn);
    }
    if (rc) return rc;
    if (posture) {
        // TrackBuffer.estimate_posture behind the step, unless the frame was skipped (k_features reads the frame's row count): the
        // rows are counted on the device, every launch covers the capacity and its surplus workgroups leave on that word
        const mmw_posture_model &m = c->model;
        const int cap = c->dc.t_cap;
        constexpr int kFlat = 3 * 64 * 32;
        launch_features(c->dc, c->st, nullptr, c->pc_feat, c->pc_owner, nullptr, cap, c->stream, raw ? c->d_nout : c->d_n, c->d_prows);
        launch_mars_conv(c->pc_feat, m.conv1_w, m.conv1_b, m.conv2_w, m.conv2_b, c->pc_act, cap, c->stream, c->d_prows);
        launch_mars_head_small(c->pc_act, kFlat, m.dense1_w, m.dense1_ld, m.dense1_b, m.dense2_w, m.dense2_b, c->pc_hidden, c->pc_kp, cap, kFlat, 1536,
                               MMW_NKP, c->stream, c->d_prows);
        launch_set_kp(c->dc, c->st, c->pc_kp, c->pc_owner, cap, c->stream, c->d_prows);
        HIPCHK(c, hipGetLastError());
    }
    // results: [assoc | db_n | n_out | posture rows] always, the labels when asked for; headers and queue words for the error check
    const size_t head = db_labels ? L.out_bytes : L.out_labels;
    HIPCHK(c, hipMemcpyAsync(c->h_out, c->d_out, head, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_hdr, c->st.hdr, S * sizeof(SceneHdr), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_q, c->d_q, kQWords * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    double *h_pts_out = nullptr;
    if (raw && pts_out) {   // normalize_data's rows (the reference's `effective_data`): into the pinned row area, now free again
        h_pts_out = reinterpret_cast<double *>(c->h_in + L.in_rows);
        HIPCHK(c, hipMemcpyAsync(h_pts_out, d_rows, S * NP * 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (assoc) memcpy(assoc, c->h_out + L.out_assoc, S * NP * sizeof(int32_t));
    if (db_n) memcpy(db_n, c->h_out + L.out_dbn, S * sizeof(int32_t));
    if (n_out) memcpy(n_out, raw ? c->h_out + L.out_nout : c->h_in + L.in_n, S * sizeof(int32_t));
    if (db_labels) memcpy(db_labels, c->h_out + L.out_labels, S * UM * sizeof(int32_t));
    if (h_pts_out) memcpy(pts_out, h_pts_out, S * NP * 8 * sizeof(double));
    if (n_tracks) for (size_t s = 0; s < S; s++) n_tracks[s] = c->h_hdr[s].n_tracks;
    if (posture_rows) *posture_rows = posture ? *reinterpret_cast<const int32_t *>(c->h_out + L.out_prows) : 0;
    return first_scene_error(c, c->h_hdr, S, c->h_q);
}

int mmw_frame_host(mmw_ctx *c, const double *raw, const double *pts, const int32_t *n, const double *dt, double *pts_out, int32_t *n_out,
                   int32_t *assoc, int32_t *db_labels, int32_t *db_n, int32_t *n_tracks)
{
    return frame_impl(c, raw, pts, n, dt, pts_out, n_out, assoc, db_labels, db_n, n_tracks, false, nullptr);
}

int mmw_frame_posture_host(mmw_ctx *c, const double *raw, const double *pts, const int32_t *n, const double *dt, double *pts_out, int32_t *n_out,
                           int32_t *assoc, int32_t *db_labels, int32_t *db_n, int32_t *n_tracks, int32_t *posture_rows)
{
    return frame_impl(c, raw, pts, n, dt, pts_out, n_out, assoc, db_labels, db_n, n_tracks, true, posture_rows);
}

int mmw_attach_posture(mmw_ctx *c, const mmw_posture_model *m)
{
    if (!c) return fail(c, MMW_E_ARG, "mmw_attach_posture: null context");
    if (!m) { c->has_model = false; return MMW_OK; }
    if (c->dc.n_scenes != 1 || c->dc.ring != 3 || c->dc.t_cap > 64)
        return fail(c, MMW_E_ARG, "mmw_attach_posture: a one-scene context of the 3-frame model (FB_FRAMES_BATCH = 2) with track_cap <= 64");
    if (!m->conv1_w || !m->conv1_b || !m->conv2_w || !m->conv2_b || !m->dense1_w || !m->dense1_b || !m->dense2_w || !m->dense2_b ||
        m->dense1_ld < 3 * 64 * 32 || (m->dense1_ld & 3) != 0 || ((uintptr_t)m->dense1_w & 15) != 0)
        return fail(c, MMW_E_ARG, "mmw_attach_posture: null weight pointer, or Dense-1 not 16-byte aligned with a leading dimension >= 6144 that is a multiple of 4");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_pchain) {
        const size_t cap = (size_t)c->dc.t_cap, per = 960 + 6144 + 1536 + 64;   // floats per row: features, conv output, hidden, keypoints (57, padded)
        HIPCHK(c, hipMalloc((void **)&c->d_pchain, cap * (per * sizeof(float) + 2 * sizeof(int32_t))));
        c->pc_feat = reinterpret_cast<float *>(c->d_pchain);
        c->pc_act = c->pc_feat + cap * 960;
        c->pc_hidden = c->pc_act + cap * 6144;
        c->pc_kp = c->pc_hidden + cap * 1536;
        c->pc_owner = reinterpret_cast<int32_t *>(c->pc_kp + cap * 64);
    }
    c->model = *m;
    c->has_model = true;
    return MMW_OK;
}

int mmw_step_host(mmw_ctx *c, const double *pts, const int32_t *n_pts, const double *dt, int32_t *assoc, int32_t *db_labels, int32_t *db_n)
{
    if (!c || !pts || !n_pts || !dt) return fail(c, MMW_E_ARG, "mmw_step_host: null input pointer");
    return frame_impl(c, nullptr, pts, n_pts, dt, nullptr, nullptr, assoc, db_labels, db_n, nullptr, false, nullptr);
}

int mmw_dbscan(mmw_ctx *c, const double *pts, const int32_t *n, int32_t max_n, double eps, int32_t min_samples, int32_t *labels, int32_t *n_clusters)
{
    if (!c || !pts || !n || !labels) return fail(c, MMW_E_ARG, "mmw_dbscan: null pointer");
    if (max_n < 1 || max_n > c->UM) return fail(c, MMW_E_ARG, "mmw_dbscan: max_n=%d must be in [1, ring*max_pts=%d]", max_n, c->UM);
    if (!(eps > 0.0) || min_samples < 1) return fail(c, MMW_E_ARG, "mmw_dbscan: eps must be > 0 and min_samples >= 1 (sklearn's DBSCAN refuses anything else)");
    HIPCHK(c, hipSetDevice(c->device));
    launch_dbscan_only(c->dc, c->st, c->UM, pts, n, max_n, eps, min_samples, labels, n_clusters, c->stream);
    HIPCHK(c, hipGetLastError());
    return MMW_OK;
}

int mmw_features_async(mmw_ctx *c, float *feat, int32_t *owner, int32_t *uid, int32_t cap_rows, int32_t ticket)
{
    if (!c || !feat || !owner || cap_rows < 0 || ticket < 0 || ticket >= kTickets) return fail(c, MMW_E_ARG, "mmw_features_async: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    EventPair ep;
    launch_feat_scan(c->dc, c->st, c->d_row_off, c->stream);
    prof_begin(c, MMW_K_FEATURES, ep);
    launch_features(c->dc, c->st, c->d_row_off, feat, owner, uid, cap_rows, c->stream);
    prof_end(c, ep);
    HIPCHK(c, hipGetLastError());
    // the total travels to pinned host memory behind the kernels; only mmw_features_wait(ticket) waits for it
    HIPCHK(c, hipMemcpyAsync(c->h_rows + ticket, c->d_row_off + c->dc.n_scenes, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->feat_ev[ticket], c->stream));
    c->feat_cap[ticket] = cap_rows;
    return MMW_OK;
}

int mmw_features_wait(mmw_ctx *c, int32_t ticket, int32_t *n_rows)
{
    if (!c || !n_rows || ticket < 0 || ticket >= kTickets) return fail(c, MMW_E_ARG, "mmw_features_wait: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->feat_ev[ticket]));
    const int32_t total = c->h_rows[ticket], cap_rows = c->feat_cap[ticket];
    *n_rows = total < cap_rows ? total : cap_rows;
    if (total > cap_rows) return fail(c, MMW_E_CAPACITY, "mmw_features: %d eligible tracks but cap_rows=%d", total, cap_rows);
    return MMW_OK;
}

int mmw_features(mmw_ctx *c, float *feat, int32_t *owner, int32_t cap_rows, int32_t *n_rows)
{
    if (!c || !feat || !owner || !n_rows || cap_rows < 0) return fail(c, MMW_E_ARG, "mmw_features: bad argument");
    const int rc = mmw_features_async(c, feat, owner, nullptr, cap_rows, kTickets - 1);
    return rc ? rc : mmw_features_wait(c, kTickets - 1, n_rows);
}

int mmw_format_frames(mmw_ctx *c, const double *frames, const int32_t *counts, const double *ref, float *feat, int32_t n_items)
{
    if (!c || n_items < 0 || (n_items > 0 && (!frames || !counts || !ref || !feat))) return fail(c, MMW_E_ARG, "mmw_format_frames: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    launch_format_frames(c->dc, frames, counts, ref, feat, n_items, c->stream);
    HIPCHK(c, hipGetLastError());
    return MMW_OK;
}

int mmw_set_keypoints(mmw_ctx *c, const float *kp, const int32_t *owner, int32_t n_rows)
{
    if (!c || (n_rows > 0 && (!kp || !owner)) || n_rows < 0) return fail(c, MMW_E_ARG, "mmw_set_keypoints: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    launch_set_kp(c->dc, c->st, kp, owner, n_rows, c->stream);
    HIPCHK(c, hipGetLastError());
    return MMW_OK;
}

int mmw_set_keypoints_uid(mmw_ctx *c, const float *kp, const int32_t *owner, const int32_t *uid, int32_t n_rows)
{
    if (!c || (n_rows > 0 && (!kp || !owner || !uid)) || n_rows < 0) return fail(c, MMW_E_ARG, "mmw_set_keypoints_uid: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    launch_set_kp_uid(c->dc, c->st, kp, owner, uid, n_rows, c->stream);
    HIPCHK(c, hipGetLastError());
    return MMW_OK;
}

static int read_headers(mmw_ctx *c, std::vector<SceneHdr> &h)
{
    HIPCHK(c, hipSetDevice(c->device));
    h.resize(c->dc.n_scenes);
    { const int rc_ = d2h_after_kernels(c, h.data(), c->st.hdr, h.size() * sizeof(SceneHdr)); if (rc_) return rc_; }
    return MMW_OK;
}

// the first per-scene error of a header read-back (as mmw_check reports it), or a chain worker's give-up
static int first_scene_error(mmw_ctx *c, const SceneHdr *h, size_t n, const int32_t *q)
{
    for (size_t s = 0; s < n; s++) {
        const int e = h[s].err;
        if (!e) continue;
        if (e & ERR_BADCOUNT) return fail(c, MMW_E_ARG, "scene %zu: n_pts outside [0, max_pts=%d]", s, c->dc.max_pts);
        if (e & ERR_CAPACITY) return fail(c, MMW_E_CAPACITY, "scene %zu: more tracks than track_cap=%d", s, c->dc.t_cap);
        // (a zero denominator leaves inf / NaN in the track's state, which the next frames' 6x6 inversions then report as
        //  singular: when both bits are set the division came first -- as the reference's ZeroDivisionError would have)
        if (e & ERR_DIVZERO) return fail(c, MMW_E_DIVZERO, "scene %zu: (N_est-1)*N == 0 in _get_Rc / N_est == 0", s);
        if (e & ERR_SINGULAR) return fail(c, MMW_E_SINGULAR, "scene %zu: singular 6x6 gate/innovation matrix", s);
        // (the last thing track() can raise in a frame: sklearn's input validation in apply_DBscan, Utils.py:272-278.  The text is
        //  sklearn's own first line: NaN wins over infinity wherever the two sit in the cloud)
        if (e & ERR_NONFINITE_NAN) return fail(c, MMW_E_NONFINITE, "scene %zu: Input X contains NaN.", s);
        if (e & ERR_NONFINITE_INF) return fail(c, MMW_E_NONFINITE, "scene %zu: Input X contains infinity or a value too large for dtype('float64').", s);
    }
    if (q[kQTimeout] != 0) return fail(c, MMW_E_HIP, "a DBSCAN chain worker gave up waiting (%d time(s)): device hung or oversubscribed", q[kQTimeout]);
    return MMW_OK;
}

int mmw_check(mmw_ctx *c)
{
    if (!c) return MMW_E_ARG;
    std::vector<SceneHdr> h;
    int rc = read_headers(c, h);
    if (rc) return rc;
    int32_t q[kQWords];
    { const int rc_ = d2h_after_kernels(c, q, c->d_q, sizeof(q)); if (rc_) return rc_; }
    return first_scene_error(c, h.data(), h.size(), q);
}

int mmw_get_num_tracks(mmw_ctx *c, int32_t *n_tracks)
{
    if (!c || !n_tracks) return MMW_E_ARG;
    std::vector<SceneHdr> h;
    int rc = read_headers(c, h);
    if (rc) return rc;
    for (size_t s = 0; s < h.size(); s++) n_tracks[s] = h[s].n_tracks;
    return MMW_OK;
}

int mmw_get_batch_ring(mmw_ctx *c, int32_t *ring_len, int32_t *ring_n)
{
    if (!c || !ring_len || !ring_n) return MMW_E_ARG;
    std::vector<SceneHdr> h;
    int rc = read_headers(c, h);
    if (rc) return rc;
    for (size_t s = 0; s < h.size(); s++) {
        ring_len[s] = h[s].g_len;
        for (int k = 0; k < MMW_RING_MAX; k++) ring_n[s * MMW_RING_MAX + k] = k < h[s].g_len ? h[s].g_n[k] : 0;
    }
    return MMW_OK;
}

int mmw_get_tracks(mmw_ctx *c, mmw_track_record *out, int32_t cap)
{
    if (!c || !out || cap < 1) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)c->dc.n_scenes * cap * sizeof(mmw_track_record);
    if (c->export_cap < cap) {
        if (c->d_export) hipFree(c->d_export);
        c->d_export = nullptr;
        HIPCHK(c, hipMalloc((void **)&c->d_export, bytes));
        c->export_cap = cap;
    }
    launch_export(c->dc, c->st, c->d_export, cap, c->stream);   // (writes every record, the empty ones as zeros: no memset in front)
    HIPCHK(c, hipGetLastError());
    return d2h_after_kernels(c, out, c->d_export, bytes);
}

int mmw_get_track_ring_frame(mmw_ctx *c, int32_t scene, int32_t track, int32_t k, double *out, int32_t *n_rows)
{
    if (!c || !out || !n_rows || scene < 0 || scene >= c->dc.n_scenes) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    SceneHdr h;
    { const int rc_ = d2h_after_kernels(c, &h, c->st.hdr + scene, sizeof(h)); if (rc_) return rc_; }
    if (track < 0 || track >= h.n_tracks) return fail(c, MMW_E_ARG, "track %d out of range (%d tracks)", track, h.n_tracks);
    int32_t slot = 0;
    { const int rc_ = d2h_after_kernels(c, &slot, c->st.order + (size_t)scene * c->dc.t_cap + track, sizeof(slot)); if (rc_) return rc_; }
    TrackRec rec;
    { const int rc_ = d2h_after_kernels(c, &rec, c->st.trk + (size_t)scene * c->dc.t_cap + slot, sizeof(rec)); if (rc_) return rc_; }
    if (k < 0 || k >= rec.ring_len) return fail(c, MMW_E_ARG, "frame %d out of range (ring_len %d)", k, rec.ring_len);
    const int keep = rec.ring_n[k] < c->dc.ring_rows ? rec.ring_n[k] : c->dc.ring_rows;
    const double *src = c->st.trk_ring + ((((size_t)scene * c->dc.t_cap + slot) * c->dc.ring + rec.ring_slot[k]) * c->dc.ring_rows) * 8;
    { const int rc_ = d2h_after_kernels(c, out, src, (size_t)keep * 8 * sizeof(double)); if (rc_) return rc_; }
    *n_rows = keep;
    return MMW_OK;
}

int mmw_get_batch_ring_frame(mmw_ctx *c, int32_t scene, int32_t k, double *out, int32_t *n_rows)
{
    if (!c || !out || !n_rows || scene < 0 || scene >= c->dc.n_scenes) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    SceneHdr h;
    { const int rc_ = d2h_after_kernels(c, &h, c->st.hdr + scene, sizeof(h)); if (rc_) return rc_; }
    if (k < 0 || k >= h.g_len) return fail(c, MMW_E_ARG, "frame %d out of range (ring_len %d)", k, h.g_len);
    const double *src = c->st.g_ring + ((size_t)scene * c->dc.ring + h.g_slot[k]) * (size_t)c->dc.max_pts * 8;
    { const int rc_ = d2h_after_kernels(c, out, src, (size_t)h.g_n[k] * 8 * sizeof(double)); if (rc_) return rc_; }
    *n_rows = h.g_n[k];
    return MMW_OK;
}

int mmw_get_inner(mmw_ctx *c, int32_t *n_calls, int32_t *rows, int32_t *labels, int32_t cap_labels)
{
    if (!c || !n_calls || cap_labels < 0) return MMW_E_ARG;
    if (!c->dc.seek_inner) return fail(c, MMW_E_ARG, "mmw_get_inner: the context was created with seek_inner = 0");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t S = c->dc.n_scenes, W = kInnerHdr + c->st.inner_cap;
    std::vector<int32_t> h(S * W);
    { const int rc_ = d2h_after_kernels(c, h.data(), c->st.inner_buf, h.size() * sizeof(int32_t)); if (rc_) return rc_; }
    for (size_t s = 0; s < S; s++) {
        const int32_t *b = h.data() + s * W;
        n_calls[s] = b[0];
        if (rows) for (int k = 0; k < 16; k++) rows[s * 16 + k] = b[2 + k];
        if (labels) {
            const int m = b[1] < cap_labels ? b[1] : cap_labels;
            memcpy(labels + s * (size_t)cap_labels, b + kInnerHdr, sizeof(int32_t) * (size_t)m);
        }
    }
    return MMW_OK;
}

int mmw_track_table(mmw_ctx *c, mmw_track_summary *table, int32_t slots, int32_t scene_base)
{
    if (!c || !table || slots < 1) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    EventPair ep;
    prof_begin(c, MMW_K_TABLE, ep);
    launch_table(c->dc, c->st, table, slots, scene_base, c->stream);
    prof_end(c, ep);
    HIPCHK(c, hipGetLastError());
    return MMW_OK;
}

int mmw_mars_conv3d(void *hip_stream, const float *feat, const float *w1, const float *b1, const float *w2, const float *b2,
                    float *out, int32_t n)
{
    if (n < 0 || (n > 0 && (!feat || !w1 || !b1 || !w2 || !b2 || !out))) return fail(nullptr, MMW_E_ARG, "mmw_mars_conv3d: bad argument");
    launch_mars_conv(feat, w1, b1, w2, b2, out, n, (hipStream_t)hip_stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, MMW_E_HIP, "mmw_mars_conv3d launch -> %s", hipGetErrorString(e));
    return MMW_OK;
}

int mmw_mars_conv_split(void *hip_stream, int32_t frames, const float *feat, const float *w1, const float *b1, const float *w2,
                        const float *b2, void *out16, int64_t ld_out, int32_t n, int32_t *range_flag, int32_t *sample_flags)
{
    if ((frames != 3 && frames != 1) || n < 0 || ld_out < 2 * (int64_t)frames * 2048 || (ld_out & 7) != 0 || ((uintptr_t)out16 & 15) != 0 ||
        (n > 0 && (!feat || !w1 || !b1 || !w2 || !b2 || !out16)))
        return fail(nullptr, MMW_E_ARG, "mmw_mars_conv_split: bad argument (frames must be 3 or 1, ld_out >= 2 * frames * 2048 and a multiple of 8)");
    if (launch_mars_conv16(frames, feat, w1, b1, w2, b2, out16, ld_out, n, range_flag, sample_flags, (hipStream_t)hip_stream) != 0)
        return fail(nullptr, MMW_E_HIP, "mmw_mars_conv_split: hipFuncSetAttribute(max dynamic LDS) failed on this device");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, MMW_E_HIP, "mmw_mars_conv_split launch -> %s", hipGetErrorString(e));
    return MMW_OK;
}

int mmw_mars_dense1_split(void *hip_stream, const void *a2, int64_t lda, const void *w2, int64_t ldw, const float *bias, float *out,
                          int32_t rows_padded, int32_t k, int32_t n)
{
    if (rows_padded < 0 || (rows_padded & 255) != 0 || k < 32 || (k & 31) != 0 || n < 128 || (n & 127) != 0 || lda < 2 * (int64_t)k || ldw < 2 * (int64_t)k ||
        ((lda | ldw) & 7) != 0 || (rows_padded > 0 && (!a2 || !w2 || !bias || !out)) || (((uintptr_t)a2 | (uintptr_t)w2) & 15) != 0)
        return fail(nullptr, MMW_E_ARG, "mmw_mars_dense1_split: rows_padded must be a multiple of 256, k of 32, n of 128; fp16 operands 16-byte aligned with leading dimensions >= 2 k that are multiples of 8");
    if (rows_padded == 0) return MMW_OK;
    if (launch_mars_dense1(a2, lda, w2, ldw, bias, out, rows_padded, k, n, (hipStream_t)hip_stream) != 0)
        return fail(nullptr, MMW_E_HIP, "mmw_mars_dense1_split: hipFuncSetAttribute failed");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, MMW_E_HIP, "mmw_mars_dense1_split launch -> %s", hipGetErrorString(e));
    return MMW_OK;
}

int mmw_mars_head_small(void *hip_stream, const float *act, int64_t lda, const float *w1, int64_t ldw, const float *bias1, const float *w2,
                        const float *bias2, float *hidden, float *kp, int32_t n_rows, int32_t k, int32_t n1)
{
    if (n_rows < 0 || n_rows > 64 || k < 4 || (k & 3) != 0 || n1 < 1 || lda < k || ldw < k || ((lda | ldw) & 3) != 0 ||
        (n_rows > 0 && (!act || !w1 || !bias1 || !w2 || !bias2 || !hidden || !kp)) || (((uintptr_t)act | (uintptr_t)w1) & 15) != 0)
        return fail(nullptr, MMW_E_ARG, "mmw_mars_head_small: n_rows in [0, 64], k a multiple of 4, 16-byte aligned fp32 operands with leading dimensions >= k that are multiples of 4");
    if (n_rows == 0) return MMW_OK;
    launch_mars_head_small(act, lda, w1, ldw, bias1, w2, bias2, hidden, kp, n_rows, k, n1, MMW_NKP, (hipStream_t)hip_stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, MMW_E_HIP, "mmw_mars_head_small launch -> %s", hipGetErrorString(e));
    return MMW_OK;
}

int mmw_mars_range_fixup(void *hip_stream, const float *feat, int32_t *sample_flags, int32_t n, const float *cw1, const float *cb1,
                         const float *cw2, const float *cb2, const float *w1, int64_t ldw, const float *bias1, const float *w2, const float *bias2,
                         void *scratch, float *kp, int32_t *range_flag)
{
    constexpr int kCap = MMW_RANGE_FIXUP_CAP, kPer = 3 * 8 * 8 * 5, kFlat = 3 * 64 * 32;
    if (n < 0 || (n > 0 && (!feat || !sample_flags || !cw1 || !cb1 || !cw2 || !cb2 || !w1 || !bias1 || !w2 || !bias2 || !scratch || !kp)) ||
        ldw < kFlat || (ldw & 3) != 0 || (((uintptr_t)scratch | (uintptr_t)w1) & 15) != 0)
        return fail(nullptr, MMW_E_ARG, "mmw_mars_range_fixup: bad argument");
    if (n == 0) return MMW_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    // scratch: small_feat[64][960], act[64][6144], hidden[64][1536], kp_small[64][57] floats (behind 512 spare bytes);
    // sample_flags = the fix-up list k_mars_conv16 appended to: [0] running count, [1] taken by this call, [2 ..] sample indices
    float *small = reinterpret_cast<float *>(reinterpret_cast<char *>(scratch) + 512);
    float *act = small + (size_t)kCap * kPer, *hidden = act + (size_t)kCap * kFlat, *kps = hidden + (size_t)kCap * 1536;
    const int32_t *taken = sample_flags + 1;
    launch_range_gather(feat, sample_flags, n, kPer, kCap, small, range_flag, st);
    launch_mars_conv(small, cw1, cb1, cw2, cb2, act, kCap, st, taken);
    launch_mars_head_small(act, kFlat, w1, ldw, bias1, w2, bias2, hidden, kps, kCap, kFlat, 1536, MMW_NKP, st, taken);
    launch_range_scatter(kps, sample_flags, kp, kCap, MMW_NKP, n, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, MMW_E_HIP, "mmw_mars_range_fixup launch -> %s", hipGetErrorString(e));
    return MMW_OK;
}

// the device keeps kStatSlots partial copies of the counters (mmw_device.hpp); the totals are formed here
static int read_stats(mmw_ctx *c, uint64_t *out, int words)
{
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> h((size_t)kStatSlots * kStatWords);
    { const int rc_ = d2h_after_kernels(c, h.data(), c->d_stats, h.size() * sizeof(uint64_t)); if (rc_) return rc_; }
    for (int w = 0; w < words; w++) {
        uint64_t sum = 0;
        for (int k = 0; k < kStatSlots; k++) sum += h[(size_t)k * kStatWords + w];
        out[w] = sum;
    }
    return MMW_OK;
}

// The packet part of ReadIWR14xx.read (ReadDataIWR1443.py:47-113), shared by mmw_parse_uart and mmw_find_tlv: the LAST magic word
// that starts in buf[0 .. len-8), the whole packet present, objects announced, the first TLV = detected points and its objects
// inside the buffer.  Returns 1 with *body = offset of the TLV body (u16 numObj, u16 Q, objects) from buf, 0 otherwise.
static int find_tlv_body(const uint8_t *buf, size_t len, size_t *body, uint32_t *num_obj, uint32_t *qfmt, uint32_t *frame_number, size_t *packet_start,
                         size_t *packet_len)
{
    static const uint8_t magic[8] = {2, 1, 4, 3, 6, 5, 8, 7};
    if (frame_number) *frame_number = 0;
    if (packet_start) *packet_start = 0;
    if (packet_len) *packet_len = 0;
    if (len <= 16) return 0;
    size_t start = len;  // the last magic word that starts in buf[0 .. len-8)
    for (size_t loc = len - 8; loc-- > 0;)
        if (memcmp(buf + loc, magic, 8) == 0) { start = loc; break; }
    if (start == len) return 0;
    if (packet_start) *packet_start = start;
    const size_t rem = len - start;
    if (rem <= 16) return 0;
    const uint8_t *p = buf + start;  // offsets below are relative to the packet
    auto u32 = [&](size_t o) { return (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8) | ((uint32_t)p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24); };
    auto u16 = [&](size_t o) { return (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8); };
    const size_t total = u32(12);
    if (rem < total || total < 36) return 0;
    if (packet_len) *packet_len = total;
    if (frame_number) *frame_number = u32(20);
    const uint32_t num_det = u32(28);
    if (num_det == 0 || total < 36 + 8 + 4) return 0;
    size_t idx = 36;
    const uint32_t tlv_type = u32(idx);
    idx += 8;  // type, length
    if (tlv_type != 1) return 0;
    const uint32_t num = u16(idx), q = u16(idx + 2);
    if (idx + 4 + (size_t)num * 12 > rem) return 0;
    *body = start + idx;
    *num_obj = num;
    *qfmt = q;
    return 1;
}

int mmw_parse_uart(const uint8_t *buf, size_t len, const mmw_uart_cfg *cfg, double *raw, double *range_out, int32_t max_obj,
                   int32_t *n_obj, uint32_t *frame_number, size_t *packet_start, size_t *packet_len)
{
    if (!buf || !cfg || !raw || !n_obj || max_obj < 0) return MMW_E_ARG;
    *n_obj = 0;
    size_t body = 0;
    uint32_t num = 0, qfmt = 0;
    if (!find_tlv_body(buf, len, &body, &num, &qfmt, frame_number, packet_start, packet_len)) return 0;
    if ((int64_t)num > (int64_t)max_obj) return MMW_E_ARG;
    auto u16 = [&](size_t o) { return (uint32_t)buf[o] | ((uint32_t)buf[o + 1] << 8); };
    size_t idx = body + 4;
    const double q = ldexp(1.0, (int)qfmt);
    const double half = cfg->num_doppler_bins / 2.0 - 1;
    for (uint32_t o = 0; o < num; o++, idx += 12) {
        const int16_t range_idx = (int16_t)u16(idx), peak = (int16_t)u16(idx + 4);
        int16_t dop = (int16_t)u16(idx + 2);
        const int16_t x = (int16_t)u16(idx + 6), y = (int16_t)u16(idx + 8), z = (int16_t)u16(idx + 10);
        if ((double)dop > half) dop = (int16_t)((int32_t)dop - 65535);  // ReadDataIWR1443.py:150-157 (wraps in int16)
        raw[o * 5 + 0] = (double)x / q;
        raw[o * 5 + 1] = (double)y / q;
        raw[o * 5 + 2] = (double)z / q;
        raw[o * 5 + 3] = (double)dop * cfg->doppler_resolution_mps;
        raw[o * 5 + 4] = (double)peak;
        if (range_out) range_out[o] = (double)range_idx * cfg->range_idx_to_meters;
    }
    *n_obj = (int32_t)num;
    return 1;
}

int mmw_find_tlv(const uint8_t *buf, size_t len, int64_t *body_offset, int32_t *n_obj, uint32_t *frame_number, size_t *packet_start, size_t *packet_len)
{
    if (!buf || !body_offset || !n_obj) return MMW_E_ARG;
    *body_offset = -1;
    *n_obj = 0;
    size_t body = 0;
    uint32_t num = 0, qfmt = 0;
    if (!find_tlv_body(buf, len, &body, &num, &qfmt, frame_number, packet_start, packet_len)) return 0;
    *body_offset = (int64_t)body;
    *n_obj = (int32_t)num;
    return 1;
}

int mmw_stats_get(mmw_ctx *c, uint64_t *out)
{
    if (!c || !out) return MMW_E_ARG;
    return read_stats(c, out, 8);
}
int mmw_stats_get_ext(mmw_ctx *c, uint64_t *out)
{
    if (!c || !out) return MMW_E_ARG;
    return read_stats(c, out, kStatWords);
}
#ifdef MMW_STAMPS
int mmw_diag_probes(mmw_ctx *c, uint64_t *out /*[256 + 8192]*/)
{
    if (!c || !out) return MMW_E_ARG;
    { const int rc_ = d2h_after_kernels(c, out, c->d_stats + (size_t)kStatSlots * kStatWords, (256 + 8192) * sizeof(uint64_t)); if (rc_) return rc_; }
    return MMW_OK;
}
#endif
int mmw_side_workers(mmw_ctx *c)
{
    if (!c) return MMW_E_ARG;
    return c->dc.side_worker ? (c->side_probed ? 1 : 2) : 0;
}
int mmw_streams_concurrent(mmw_ctx *c, void *stream_a, void *stream_b)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int ok = probe_one(c, (hipStream_t)stream_b, (hipStream_t)stream_a);
    if (ok < 0) return fail(c, MMW_E_HIP, "mmw_streams_concurrent: probe failed: %s", hipGetErrorString(hipGetLastError()));
    return ok;
}
int mmw_step_kind(mmw_ctx *c)
{
    if (!c) return MMW_E_ARG;
    return c->dc.fused ? 1 : (pred_in_track(c->dc) ? 2 : 4);
}
int mmw_kalman_layout(mmw_ctx *c)
{
    if (!c) return MMW_E_ARG;
    int nq = (c->dc.tr_max_tracks + 3) / 4;
    if (nq < 1) nq = 1;
    return (!c->dc.fused && tracks_dense(c->dc, nq)) ? 1 : 0;
}
int mmw_diag_queue(mmw_ctx *c, int32_t *out /*[32]*/)
{
    if (!c || !out) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(out, c->d_q, kQWords * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MMW_OK;
}
int mmw_stats_reset(mmw_ctx *c)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemsetAsync(c->d_stats, 0, (size_t)kStatSlots * kStatWords * sizeof(uint64_t), c->stream));
    return MMW_OK;
}

int mmw_profile_enable(mmw_ctx *c, int32_t on)
{
    if (!c) return MMW_E_ARG;
    c->prof_mask = (on & 1) ? ~0u : ((unsigned)on >> 1);  // no synchronisation here: mmw_profile_get folds the pending pairs
    return MMW_OK;
}
int mmw_profile_reset(mmw_ctx *c)
{
    if (!c) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStreamSynchronize(c->stream);
    prof_fold(c);
    for (int k = 0; k < MMW_K_COUNT; k++) { c->tot_ms[k] = 0; c->launches[k] = 0; }
    return MMW_OK;
}
int mmw_profile_get(mmw_ctx *c, int32_t k, double *total_ms, int64_t *launches)
{
    if (!c || k < 0 || k >= MMW_K_COUNT) return MMW_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStreamSynchronize(c->stream);
    prof_fold(c);
    if (total_ms) *total_ms = c->tot_ms[k];
    if (launches) *launches = c->launches[k];
    return MMW_OK;
}

}  // extern "C"
