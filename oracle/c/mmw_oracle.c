/*
 * mmw_oracle.c -- see mmw_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C99, fp64, no FMA contraction (build with -ffp-contract=off).  Every
 * function cites the reference lines it restates (paths relative to
 * /root/reference/src unless noted).
 */
#include "mmw_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DX_MAX 9
#define DZ 6

typedef struct {
    double x[DX_MAX];
    double P[DX_MAX * DX_MAX];
    double centroid[6], minv[6], maxv[6];
    double spread_est[6];
    double gd[36];
    double n_est;
    double lifetime;
    int32_t point_num;
    int32_t is_static;
    int32_t ring_len;
    int32_t ring_n[ORC_RING_MAX];
    double *ring[ORC_RING_MAX]; /* each [ring_rows][8], oldest first */
    float keypoints[ORC_NKP];
    int32_t ring_size;          /* track.batch.size: FB_FRAMES_BATCH + 1 until change_buffer_size (Tracking.py:60-64) */
} trk_t;

struct orc_scene {
    orc_config cfg;
    int max_pts;
    int ring_size;
    int n_tracks;
    trk_t *tracks;
    /* global BatchedData (Tracking.py:21-71) */
    int g_size;                      /* batch.size: FB_FRAMES_BATCH + 1 until change_buffer_size */
    int g_len;
    int32_t g_n[ORC_RING_MAX];
    double *g_frame[ORC_RING_MAX]; /* each [max_pts][8] */
    /* scratch */
    double *concat;  /* [ring*max_pts][8] */
    double *cloud;   /* [ring*max_pts][8] */
    double *prod;    /* [ring*max_pts] */
    int32_t *tmp_i;
    /* seek_inner_clusters of the last frame (cfg.seek_inner only) */
    int inner_calls;
    int32_t *inner_track, *inner_n;  /* [track_cap] */
    int32_t *inner_labels;           /* [track_cap][ring*max_pts] */
    double *inner_rows;              /* [track_cap][ring*max_pts][8]: the cluster a call returned (label 1) */
    int32_t *inner_m;                /* [track_cap] its row count, 0 = none */
    int overflow;                    /* a ring frame longer than ring_rows under seek_inner */
};

/* ------------------------------------------------------------------ */
/* log(): fdlibm-style argument reduction + degree-14 minimax polynomial
 * (the published Sun fdlibm e_log.c algorithm), restated so that the CPU
 * oracle and the device code can use one arithmetic definition of log|det|
 * (Tracking.py:558 uses np.log).  < 1 ULP. */
double orc_log(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    uint64_t u;
    int k = 0;
    if (x != x) return x;
    if (x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (isinf(x)) return x;
    memcpy(&u, &x, 8);
    if ((u >> 52) == 0) { /* subnormal */
        x *= 18014398509481984.0; /* 2^54 */
        k -= 54;
        memcpy(&u, &x, 8);
    }
    {
        uint32_t hx = (uint32_t)(u >> 32);
        hx += 0x3ff00000u - 0x3fe6a09eu;
        k += (int)(hx >> 20) - 0x3ff;
        hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
        u = ((uint64_t)hx << 32) | (u & 0xffffffffu);
        memcpy(&x, &u, 8);
    }
    {
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
}

/* numpy pairwise_sum_DOUBLE (numpy/_core/src/umath/loops_utils.h.src), the
 * summation order of 1-D np.mean used by ClusterTrack._get_D (Tracking.py:286). */
double orc_np_pairwise_sum(const double *a, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        int i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return orc_np_pairwise_sum(a, n2) + orc_np_pairwise_sum(a + n2, n - n2);
    }
}

/* ------------------------------------------------------------------ */
void orc_config_default(orc_config *c)
{
    static const double lim[6] = {0.2, 0.2, 2, 1.2, 1.2, 0.2};
    static const float posture[ORC_NKP] = {
        0.0000f, -0.0007f, -0.0006f, -0.0038f, -0.1820f, -0.2540f, -0.2579f, 0.1830f, 0.2957f, 0.2940f,
        -0.0805f, -0.1141f, -0.1232f, -0.1358f, 0.0796f, 0.1436f, 0.1558f, 0.1720f, -0.0007f, 0.7699f,
        1.0906f, 1.4020f, 1.5513f, 1.2893f, 1.0360f, 0.7994f, 1.2865f, 1.0483f, 0.8117f, 0.7670f,
        0.3428f, 0.0000f, -0.0746f, 0.7713f, 0.3706f, -0.0128f, -0.0796f, 1.3255f, 0.0752f, 0.0533f,
        0.0203f, 0.0000f, 0.0496f, 0.1350f, 0.1303f, 0.0345f, 0.1277f, 0.1050f, 0.0392f, 0.0533f,
        0.0786f, -0.0056f, 0.0346f, -0.0007f, 0.0683f, -0.0082f, 0.0312f};
    memset(c, 0, sizeof(*c));
    c->fb_frames_batch = 2;
    c->db_min_samples = 35;
    c->tr_max_tracks = 4;
    c->kf_enable_est = 0;
    c->model_min_input = 0;
    c->dim_x = 9;
    c->ring_rows = 64;
    c->track_cap = 0;
    c->db_z_weight = 0.4;
    c->db_range_weight = 0.03;
    c->db_eps = 0.3;
    c->tr_lifetime_dynamic = 3;
    c->tr_lifetime_static = 7;
    c->tr_vel_thres = 0.12;
    c->tr_gate = 4.5;
    c->kf_q_std = 1;
    c->kf_p_init = 0.1;
    c->kf_group_disp_est_init = 0.1;
    c->kf_a_n = 0.9;
    c->kf_est_pointnum = 10;
    memcpy(c->kf_spread_lim, lim, sizeof(lim));
    c->kf_a_spr = 0.9;
    c->intensity_mu = 27.0187;
    c->intensity_std = 70.351;
    c->s_height = 1.8;
    {
        double ang = -5.0 * (M_PI / 180.0);
        c->tilt_cos = cos(ang);
        c->tilt_sin = sin(ang);
    }
    memcpy(c->default_posture, posture, sizeof(posture));
    c->seek_inner = 0;
    c->db_points_thres = 40;
    c->fb_frames_batch_static = 2;
    c->db_spread_thres = 0.7;
    c->db_inner_eps = 0.1;
}

static int default_track_cap(const orc_config *c, int max_pts)
{
    int ring = c->fb_frames_batch + 1;
    int ms = c->db_min_samples > 0 ? c->db_min_samples : 1;
    int cap = (c->tr_max_tracks > 0 ? c->tr_max_tracks - 1 : 0) + (ring * max_pts) / ms + 1;
    if (cap > 64) cap = 64;
    if (cap < 1) cap = 1;
    return cap;
}

orc_scene *orc_scene_new(const orc_config *cfg, int max_pts)
{
    orc_scene *s = (orc_scene *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->cfg = *cfg;
    if (s->cfg.ring_rows < 64) s->cfg.ring_rows = 64;
    if (s->cfg.fb_frames_batch < 0 || s->cfg.fb_frames_batch + 1 > ORC_RING_MAX) { free(s); return NULL; }
    if (s->cfg.dim_x != 9 && s->cfg.dim_x != 6) { free(s); return NULL; }
    s->max_pts = max_pts;
    s->ring_size = s->cfg.fb_frames_batch + 1;
    /* seek_inner_clusters clusters whole ring frames, and the first frame of a track it spawns is a cluster of up to
     * ring*max_pts rows: frames are stored whole (a longer one is a capacity error, not a truncation) */
    if (s->cfg.seek_inner && s->cfg.ring_rows < s->ring_size * max_pts) s->cfg.ring_rows = s->ring_size * max_pts;
    if (s->cfg.track_cap <= 0) s->cfg.track_cap = default_track_cap(&s->cfg, max_pts);
    s->tracks = (trk_t *)calloc((size_t)s->cfg.track_cap, sizeof(trk_t));
    for (int t = 0; t < s->cfg.track_cap; t++)
        for (int k = 0; k < s->ring_size; k++)
            s->tracks[t].ring[k] = (double *)calloc((size_t)s->cfg.ring_rows * 8, sizeof(double));
    for (int k = 0; k < s->ring_size; k++)
        s->g_frame[k] = (double *)calloc((size_t)max_pts * 8, sizeof(double));
    s->concat = (double *)calloc((size_t)s->ring_size * max_pts * 8, sizeof(double));
    s->cloud = (double *)calloc((size_t)s->ring_size * max_pts * 8, sizeof(double));
    s->prod = (double *)calloc((size_t)s->ring_size * max_pts + 8, sizeof(double));
    s->tmp_i = (int32_t *)calloc((size_t)s->ring_size * max_pts + 8, sizeof(int32_t));
    if (s->cfg.seek_inner) {
        const size_t cap = (size_t)s->cfg.track_cap, um = (size_t)s->ring_size * s->cfg.ring_rows;
        if (s->cfg.fb_frames_batch < 1 || s->cfg.fb_frames_batch > s->ring_size ||
            s->cfg.fb_frames_batch_static < 1 || s->cfg.fb_frames_batch_static > s->ring_size) {
            /* the inner DBSCAN reads whole clouds from the track rings, and a ring of size 0 never terminates add_frame */
            orc_scene_free(s);
            return NULL;
        }
        s->inner_track = (int32_t *)calloc(cap, sizeof(int32_t));
        s->inner_n = (int32_t *)calloc(cap, sizeof(int32_t));
        s->inner_m = (int32_t *)calloc(cap, sizeof(int32_t));
        s->inner_labels = (int32_t *)calloc(cap * um, sizeof(int32_t));
        s->inner_rows = (double *)calloc(cap * um * 8, sizeof(double));
    }
    orc_scene_reset(s);
    return s;
}

void orc_scene_free(orc_scene *s)
{
    if (!s) return;
    for (int t = 0; t < s->cfg.track_cap; t++)
        for (int k = 0; k < ORC_RING_MAX; k++) free(s->tracks[t].ring[k]);
    for (int k = 0; k < ORC_RING_MAX; k++) free(s->g_frame[k]);
    free(s->tracks); free(s->concat); free(s->cloud); free(s->prod); free(s->tmp_i);
    free(s->inner_track); free(s->inner_n); free(s->inner_m); free(s->inner_labels); free(s->inner_rows);
    free(s);
}

/* Fresh TrackBuffer() + BatchedData(): the global ring starts with ONE empty
 * frame (RingBuffer.__init__ appends init_val, Utils.py:35-41; Tracking.py:38-41). */
void orc_scene_reset(orc_scene *s)
{
    s->n_tracks = 0;
    s->g_len = 1;
    s->g_size = s->ring_size;
    s->inner_calls = 0;
    s->overflow = 0;
    memset(s->g_n, 0, sizeof(s->g_n));
}

/* ------------------------------------------------------------------ */
/* Motion model: constants.py:176-243 and filterpy.common.Q_discrete_white_noise(dim=3) */
static void build_F(int dx, double dt, double *F)
{
    memset(F, 0, sizeof(double) * dx * dx);
    for (int i = 0; i < dx; i++) F[i * dx + i] = 1.0;
    for (int i = 0; i < 3; i++) F[i * dx + (i + 3)] = dt;
    if (dx == 9) {
        double h = 0.5 * (dt * dt);
        for (int i = 0; i < 3; i++) {
            F[i * dx + (i + 6)] = h;
            F[(i + 3) * dx + (i + 6)] = dt;
        }
    }
}

static void build_Q(int dx, double dt, double var, double *Q)
{
    double dt2 = dt * dt, dt3 = dt2 * dt, dt4 = dt2 * dt2;
    double q[9] = {0.25 * dt4, 0.5 * dt3, 0.5 * dt2, 0.5 * dt3, dt2, dt, 0.5 * dt2, dt, 1.0};
    memset(Q, 0, sizeof(double) * dx * dx);
    for (int b = 0; b < dx / 3; b++)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Q[(3 * b + i) * dx + (3 * b + j)] = q[i * 3 + j] * var;
}

/* C[m x n] = A[m x k] * B[k x n] (or B^T when tb), sequential k, first product seeds the sum */
static void matmul(const double *A, const double *B, double *C, int m, int k, int n, int tb)
{
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = A[i * k] * (tb ? B[j * k] : B[j]);
            for (int t = 1; t < k; t++) s += A[i * k + t] * (tb ? B[j * k + t] : B[t * n + j]);
            C[i * n + j] = s;
        }
}

/* filterpy KalmanFilter.predict (F=, Q=), Tracking.py:372-385 */
static void kf_predict(const orc_config *c, trk_t *t, double dtm)
{
    int dx = c->dim_x;
    double F[81], Q[81], A[81], B[81], xn[9];
    build_F(dx, dtm, F);
    build_Q(dx, dtm, c->kf_q_std, Q);
    matmul(F, t->x, xn, dx, dx, 1, 0);
    memcpy(t->x, xn, sizeof(double) * dx);
    matmul(F, t->P, A, dx, dx, dx, 0);
    matmul(A, F, B, dx, dx, dx, 1);
    for (int i = 0; i < dx * dx; i++) t->P[i] = B[i] + Q[i];
}

/* 6x6 LU with partial pivoting -> determinant and inverse (stands for
 * np.linalg.det / np.linalg.inv at Tracking.py:558-560 and filterpy's inv(S)). */
static int lu6(const double *M, double *inv, double *det)
{
    double A[36], rp[6];   /* rp[k] = 1 / pivot k: ONE division per pivot, every other "division"
                            * is a multiplication by it (the GPU kernel does the same, bit for bit) */
    int perm[6];
    int neg = 0;
    memcpy(A, M, sizeof(A));
    for (int i = 0; i < 6; i++) perm[i] = i;
    for (int k = 0; k < 6; k++) {
        int p = k;
        double best = fabs(A[k * 6 + k]);
        for (int r = k + 1; r < 6; r++) {
            double v = fabs(A[r * 6 + k]);
            if (v > best) { best = v; p = r; }
        }
        if (!(best > 0.0)) return -1;
        if (p != k) {
            for (int cc = 0; cc < 6; cc++) { double tmp = A[k * 6 + cc]; A[k * 6 + cc] = A[p * 6 + cc]; A[p * 6 + cc] = tmp; }
            { int tp = perm[k]; perm[k] = perm[p]; perm[p] = tp; }
            neg ^= 1;
        }
        rp[k] = 1.0 / A[k * 6 + k];
        for (int r = k + 1; r < 6; r++) {
            double l = A[r * 6 + k] * rp[k];
            A[r * 6 + k] = l;
            for (int cc = k + 1; cc < 6; cc++) A[r * 6 + cc] = A[r * 6 + cc] - l * A[k * 6 + cc];
        }
    }
    {
        double d = A[0];
        for (int k = 1; k < 6; k++) d = d * A[k * 6 + k];
        *det = neg ? -d : d;
    }
    for (int col = 0; col < 6; col++) {
        double y[6];
        for (int r = 0; r < 6; r++) {
            double s = (perm[r] == col) ? 1.0 : 0.0;
            for (int k = 0; k < r; k++) s = s - A[r * 6 + k] * y[k];
            y[r] = s;
        }
        /* back substitution in axpy (column-sweep) order: x_r subtracts U[r][k]*x_k for k = 5 down
         * to r+1, then scales by 1/U[r][r] -- the order a lane-parallel solver produces naturally */
        for (int r = 5; r >= 0; r--) {
            double s = y[r];
            for (int k = 5; k > r; k--) s = s - A[r * 6 + k] * inv[k * 6 + col];
            inv[r * 6 + col] = s * rp[r];
        }
    }
    return 0;
}

/* PointCluster.__init__ Tracking.py:120-136 */
static void cluster_stats(const orc_config *c, trk_t *t, const double *rows, int n)
{
    double sum[6];
    for (int m = 0; m < 6; m++) { sum[m] = 0.0; t->minv[m] = rows[m]; t->maxv[m] = rows[m]; }
    for (int i = 0; i < n; i++)
        for (int m = 0; m < 6; m++) {
            double v = rows[i * 8 + m];
            sum[m] += v;
            if (v < t->minv[m]) t->minv[m] = v;
            if (v > t->maxv[m]) t->maxv[m] = v;
        }
    for (int m = 0; m < 6; m++) t->centroid[m] = sum[m] / (double)n;
    t->point_num = n;
    {
        double v3 = t->centroid[3], v4 = t->centroid[4], v5 = t->centroid[5];
        double nrm = sqrt((v3 * v3 + v4 * v4) + v5 * v5);
        t->is_static = nrm < c->tr_vel_thres;
    }
}

/* BatchedData.add_frame for a per-track ring (Tracking.py:43-51); only the first
 * ring_rows rows are stored (format_single_frame reads rows [:64], Utils.py:505-510). */
static void track_ring_push(const orc_scene *s, trk_t *t, const double *rows, int n)
{
    int keep = n < s->cfg.ring_rows ? n : s->cfg.ring_rows;
    if (s->cfg.seek_inner && n > s->cfg.ring_rows) ((orc_scene *)s)->overflow = 1;
    while (t->ring_len >= t->ring_size) {
        double *first = t->ring[0];
        for (int k = 1; k < t->ring_len; k++) { t->ring[k - 1] = t->ring[k]; t->ring_n[k - 1] = t->ring_n[k]; }
        t->ring[t->ring_len - 1] = first;
        t->ring_len--;
    }
    memcpy(t->ring[t->ring_len], rows, sizeof(double) * 8 * (size_t)keep);
    t->ring_n[t->ring_len] = n;
    t->ring_len++;
}

/* ClusterTrack.__init__ + KalmanState.__init__ Tracking.py:87-97, 210-230 */
static void track_init(const orc_scene *s, trk_t *t, const double *rows, int n)
{
    const orc_config *c = &s->cfg;
    int dx = c->dim_x;
    double *ring[ORC_RING_MAX];
    memcpy(ring, t->ring, sizeof(ring));
    memset(t, 0, sizeof(*t));
    memcpy(t->ring, ring, sizeof(ring));
    cluster_stats(c, t, rows, n);
    t->ring_len = 0;
    t->ring_size = s->ring_size;   /* a fresh BatchedData (Tracking.py:38-41) */
    track_ring_push(s, t, rows, n);
    for (int i = 0; i < 6; i++) t->x[i] = t->centroid[i];
    for (int i = 0; i < dx; i++) t->P[i * dx + i] = 1.0 * c->kf_p_init;
    for (int i = 0; i < 6; i++) t->gd[i * 6 + i] = 1.0 * c->kf_group_disp_est_init;
    t->n_est = 0.0;
    t->lifetime = 0.0;
    memcpy(t->keypoints, c->default_posture, sizeof(t->keypoints));
}

/* ClusterTrack.associate_pointcloud Tracking.py:314-341 (+232-297) */
static int track_associate(orc_scene *s, trk_t *t, const double *rows, int n)
{
    const orc_config *c = &s->cfg;
    cluster_stats(c, t, rows, n);
    track_ring_push(s, t, rows, n);
    /* _estimate_point_num Tracking.py:232-244 */
    if (c->kf_enable_est) {
        if ((double)n > t->n_est) t->n_est = (double)n;
        else t->n_est = (1 - c->kf_a_n) * t->n_est + c->kf_a_n * (double)n;
    } else {
        t->n_est = c->kf_est_pointnum > (double)n ? c->kf_est_pointnum : (double)n;
    }
    /* _estimate_measurement_spread Tracking.py:246-268 */
    for (int m = 0; m < 6; m++) {
        double spread = t->maxv[m] - t->minv[m];
        double lim2 = 2 * c->kf_spread_lim[m];
        if (n != 1) spread = spread * (double)(n + 1) / (double)(n - 1);
        spread = spread < lim2 ? spread : lim2;
        spread = spread > c->kf_spread_lim[m] ? spread : c->kf_spread_lim[m];
        if (spread > t->spread_est[m]) t->spread_est[m] = spread;
        else t->spread_est[m] = (1.0 - c->kf_a_spr) * t->spread_est[m] + c->kf_a_spr * spread;
    }
    /* _estimate_group_disp_matrix + _get_D Tracking.py:270-297 */
    {
        double a;
        double *prod = s->prod;
        if (t->n_est == 0.0) return -3;
        a = (double)n / t->n_est;
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) {
                double d;
                for (int r = 0; r < n; r++)
                    prod[r] = (rows[r * 8 + i] - t->centroid[i]) * (rows[r * 8 + j] - t->centroid[j]);
                d = orc_np_pairwise_sum(prod, n) / (double)n;
                t->gd[i * 6 + j] = (1 - a) * t->gd[i * 6 + j] + a * d;
            }
    }
    return 0;
}

int orc_dbscan(const orc_config *cfg, const double *pts, int n, double eps, int min_samples, int32_t *labels);

/* ClusterTrack.seek_inner_clusters Tracking.py:409-448, for the track at list position j whose associate_pointcloud
 * has just run on `rows`.  Quirks kept: `spread.any() > DB_SPREAD_THRES` compares a bool (x-spread != 0) with the
 * threshold; the cluster's cloud is added to the ring a second time; change_buffer_size is permanent;
 * apply_DBscan is called with eps = DB_INNER_EPS and the DEFAULT min_samples (DB_INNER_MIN_SAMPLES is unused);
 * only the cluster with label 1 is returned. */
static int seek_inner(orc_scene *s, trk_t *t, int j, const double *rows, int n)
{
    const orc_config *c = &s->cfg;
    const size_t um = (size_t)s->ring_size * s->cfg.ring_rows;
    const double xspread = t->maxv[0] - t->minv[0];
    const double any = (xspread != 0.0) ? 1.0 : 0.0;   /* numpy: bool(spread.any()) */
    int k = s->inner_calls, U = 0, ncl;
    double *cat;
    int32_t *lab;
    if (!(t->point_num > c->db_points_thres && any > c->db_spread_thres)) return 0;
    t->ring_size = t->is_static ? c->fb_frames_batch_static : c->fb_frames_batch;
    track_ring_push(s, t, rows, n);
    cat = s->inner_rows + (size_t)k * um * 8;   /* effective_data, then overwritten by the returned cluster */
    lab = s->inner_labels + (size_t)k * um;
    for (int f = 0; f < t->ring_len; f++) {
        memcpy(cat + (size_t)U * 8, t->ring[f], sizeof(double) * 8 * (size_t)t->ring_n[f]);
        U += t->ring_n[f];
    }
    ncl = orc_dbscan(c, cat, U, c->db_inner_eps, c->db_min_samples, lab);
    if (ncl < 0) return ncl;   /* sklearn's ValueError on a non-finite row (an assigned point's doppler / peakVal: the gate only sees
                                  columns 0..5) leaves track() here, in the middle of _associate_points_to_tracks */
    s->inner_track[k] = j;
    s->inner_n[k] = U;
    s->inner_m[k] = 0;
    if (ncl > 1) {
        int m = 0;
        for (int i = 0; i < U; i++)
            if (lab[i] == 1) { if (m != i) memcpy(cat + (size_t)m * 8, cat + (size_t)i * 8, 64); m++; }
        s->inner_m[k] = m;
    }
    s->inner_calls = k + 1;
    return 0;
}

/* ClusterTrack.update_state Tracking.py:387-398 + filterpy update + _get_Rc 299-312 */
static int kf_update(const orc_config *c, trk_t *t)
{
    int dx = c->dim_x;
    double N = (double)t->point_num;
    double den = (t->n_est - 1) * N;
    double coef, Rc[36], S[36], SI[36], K[54], y[6], det;
    double IKH[81], A[81], B[81], C1[54], C2[81];
    if (den == 0.0) return -3;
    coef = (t->n_est - N) / den;
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) {
            double rm = 0.0;
            if (a == b) { double h = t->spread_est[a] / 2; rm = h * h; }
            Rc[a * 6 + b] = rm / N + coef * t->gd[a * 6 + b];
        }
    for (int a = 0; a < 6; a++) y[a] = t->centroid[a] - t->x[a];
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) S[a * 6 + b] = t->P[a * dx + b] + Rc[a * 6 + b];
    if (lu6(S, SI, &det) != 0) return -2;
    for (int i = 0; i < dx; i++)
        for (int j = 0; j < 6; j++) {
            double s = t->P[i * dx] * SI[j];
            for (int k = 1; k < 6; k++) s += t->P[i * dx + k] * SI[k * 6 + j];
            K[i * 6 + j] = s;
        }
    for (int i = 0; i < dx; i++) {
        double s = K[i * 6] * y[0];
        for (int k = 1; k < 6; k++) s += K[i * 6 + k] * y[k];
        t->x[i] = t->x[i] + s;
    }
    for (int i = 0; i < dx; i++)
        for (int j = 0; j < dx; j++) {
            double d = (i == j) ? 1.0 : 0.0;
            IKH[i * dx + j] = j < 6 ? d - K[i * 6 + j] : d;
        }
    matmul(IKH, t->P, A, dx, dx, dx, 0);
    matmul(A, IKH, B, dx, dx, dx, 1);
    matmul(K, Rc, C1, dx, 6, 6, 0);
    matmul(C1, K, C2, dx, 6, dx, 1);
    for (int i = 0; i < dx * dx; i++) t->P[i] = B[i] + C2[i];
    /* Tracking.py:396-398: abs(variance.any()) > 0.6  <=>  z[0] != x[0] */
    {
        double var = t->centroid[0] - t->x[0];
        if (!(var == 0.0) && t->lifetime == 0.0) t->x[0] += var * 0.4;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* altered_EuclideanDist Utils.py:222-247 (operation order kept) */
static inline double alt_dist(const orc_config *c, const double *p1, const double *p2)
{
    double w = 1 - ((p1[1] + p2[1]) / 2) * c->db_range_weight;
    double dx = p1[0] - p2[0], dy = p1[1] - p2[1], dz = p1[2] - p2[2];
    return w * ((dx * dx + dy * dy) + c->db_z_weight * (dz * dz));
}

typedef struct {
    int n, n_nodes;
    int *idx;          /* tree order -> point index */
    int *start, *end;  /* per node */
    uint8_t *leaf;
    double *sum;       /* [n_nodes][3] */
    double *cen;       /* [n_nodes][3] */
    double *rad;       /* [n_nodes] */
    const double *X;   /* [n][8] */
    const orc_config *cfg;
} btree_t;

typedef struct { double v; int i; } keyidx_t;
static int cmp_keyidx(const void *a, const void *b)
{
    const keyidx_t *x = (const keyidx_t *)a, *y = (const keyidx_t *)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return (x->i > y->i) - (x->i < y->i);
}
static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }

/* BinaryTree._recursive_build + init_node (sklearn/neighbors/_binary_tree.pxi.tp:1040-1084,
 * _ball_tree.pyx.tp:84-144).  std::nth_element's internal permutation is NOT
 * emulated: both halves are kept in ascending point-index order and a node's
 * coordinate sum is defined as (left-child sum) + (right-child sum), leaves
 * summed in ascending index order.  The partition SETS are identical to
 * sklearn's (strict total order (value, index), _partition_nodes.pyx:35-39);
 * centroids can differ from sklearn's by a few ULP (pinned: 0 neighbour-set
 * mismatches against sklearn 1.7.2 BallTree.query_radius, tests/test_oracle_dbscan.py). */
static void bt_build(btree_t *t, int node, int s, int e, keyidx_t *scratch)
{
    const double *X = t->X;
    t->start[node] = s;
    t->end[node] = e;
    if (2 * node + 1 >= t->n_nodes) {
        double acc[3] = {0.0, 0.0, 0.0};
        t->leaf[node] = 1;
        for (int i = s; i < e; i++)
            for (int m = 0; m < 3; m++) acc[m] += X[t->idx[i] * 8 + m];
        for (int m = 0; m < 3; m++) t->sum[node * 3 + m] = acc[m];
    } else {
        int jmax = 0, np_ = e - s, nmid = np_ / 2;
        double max_spread = 0;
        t->leaf[node] = 0;
        /* find_node_split_dim _binary_tree.pxi.tp:598-645 (all 8 features) */
        for (int j = 0; j < 8; j++) {
            double mx = X[t->idx[s] * 8 + j], mn = mx, spread;
            for (int i = s + 1; i < e; i++) {
                double v = X[t->idx[i] * 8 + j];
                mx = fmax(mx, v);
                mn = fmin(mn, v);
            }
            spread = mx - mn;
            if (spread > max_spread) { max_spread = spread; jmax = j; }
        }
        for (int i = 0; i < np_; i++) { scratch[i].v = X[t->idx[s + i] * 8 + jmax]; scratch[i].i = t->idx[s + i]; }
        qsort(scratch, (size_t)np_, sizeof(keyidx_t), cmp_keyidx);
        for (int i = 0; i < np_; i++) t->idx[s + i] = scratch[i].i;
        qsort(t->idx + s, (size_t)nmid, sizeof(int), cmp_int);
        qsort(t->idx + s + nmid, (size_t)(np_ - nmid), sizeof(int), cmp_int);
        bt_build(t, 2 * node + 1, s, s + nmid, scratch);
        bt_build(t, 2 * node + 2, s + nmid, e, scratch);
        for (int m = 0; m < 3; m++) t->sum[node * 3 + m] = t->sum[(2 * node + 1) * 3 + m] + t->sum[(2 * node + 2) * 3 + m];
    }
    {
        double cpt[3], r = 0;
        for (int m = 0; m < 3; m++) { cpt[m] = t->sum[node * 3 + m] / (double)(e - s); t->cen[node * 3 + m] = cpt[m]; }
        for (int i = s; i < e; i++) r = fmax(r, alt_dist(t->cfg, cpt, X + t->idx[i] * 8));
        t->rad[node] = r;
    }
}

/* BinaryTree._query_radius_single _binary_tree.pxi.tp:1903-1980 */
static void bt_query(const btree_t *t, int node, const double *pt, double r, uint8_t *row)
{
    double d = alt_dist(t->cfg, pt, t->cen + node * 3);
    double lb = fmax(0, d - t->rad[node]);
    double ub = d + t->rad[node];
    if (lb > r) {
        return;
    } else if (ub <= r) {
        for (int i = t->start[node]; i < t->end[node]; i++) row[t->idx[i]] = 1;
    } else if (t->leaf[node]) {
        for (int i = t->start[node]; i < t->end[node]; i++)
            if (alt_dist(t->cfg, pt, t->X + t->idx[i] * 8) <= r) row[t->idx[i]] = 1;
    } else {
        bt_query(t, 2 * node + 1, pt, r, row);
        bt_query(t, 2 * node + 2, pt, r, row);
    }
}

int orc_dbscan_neighbors(const orc_config *cfg, const double *pts, int n, double eps, uint8_t *adj)
{
    btree_t t;
    int n_levels, ok = 0;
    keyidx_t *scratch;
    if (n <= 0) return 0;
    /* NearestNeighbors._fit, algorithm="auto" (sklearn/neighbors/_base.py:622-633): DBSCAN builds its NearestNeighbors with the
     * default n_neighbors = 5, and `self.n_neighbors >= n_samples // 2` sends clouds of 1 .. 11 points to "brute" -- the exact
     * pairwise matrix of the callable, `d <= radius` (radius_neighbors' "brute" arm, _base.py:1221-1250 -> pairwise_distances_chunked ->
     * metrics/pairwise.py:_pairwise_callable -> _radius_neighbors_reduce_func, _base.py:1054-1081): no BallTree, hence no PRUNE / take-all shortcut.  (The other arms of
     * that test never hold here: 8 columns <= 15, the metric is not "precomputed".) */
    if (n / 2 <= ORC_SK_N_NEIGHBORS) {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                adj[(size_t)i * n + j] = alt_dist(cfg, pts + (size_t)i * 8, pts + (size_t)j * 8) <= eps;
        return 0;
    }
    /* BinaryTree.__init__ _binary_tree.pxi.tp:876-878, leaf_size = 30 */
    {
        double q = (double)(n - 1) / 30.0;
        if (q < 1) q = 1;
        n_levels = (int)(log2(q) + 1);
    }
    t.n = n;
    t.n_nodes = (1 << n_levels) - 1;
    t.X = pts;
    t.cfg = cfg;
    t.idx = (int *)malloc(sizeof(int) * (size_t)n);
    t.start = (int *)malloc(sizeof(int) * (size_t)t.n_nodes);
    t.end = (int *)malloc(sizeof(int) * (size_t)t.n_nodes);
    t.leaf = (uint8_t *)malloc((size_t)t.n_nodes);
    t.sum = (double *)malloc(sizeof(double) * 3 * (size_t)t.n_nodes);
    t.cen = (double *)malloc(sizeof(double) * 3 * (size_t)t.n_nodes);
    t.rad = (double *)malloc(sizeof(double) * (size_t)t.n_nodes);
    scratch = (keyidx_t *)malloc(sizeof(keyidx_t) * (size_t)n);
    for (int i = 0; i < n; i++) t.idx[i] = i;
    bt_build(&t, 0, 0, n, scratch);
    memset(adj, 0, (size_t)n * (size_t)n);
    for (int i = 0; i < n; i++) bt_query(&t, 0, pts + (size_t)i * 8, eps, adj + (size_t)i * n);
    free(t.idx); free(t.start); free(t.end); free(t.leaf); free(t.sum); free(t.cen); free(t.rad); free(scratch);
    return ok;
}

/* sklearn's input validation in front of DBSCAN.fit (the reference's call site: Utils.py:272-278 -> DBSCAN.fit_predict ->
 * validate_data -> check_array(ensure_all_finite=True) -> _assert_all_finite, sklearn/utils/validation.py): ALL 8 columns of
 * the cloud.  ValueError("Input X contains NaN. ...") when any value is NaN, else ValueError("Input X contains infinity or a
 * value too large for dtype('float64').") when any is +-inf -- the precedence of sklearn 1.7.2 as it runs here (get_namespace
 * hands _assert_all_finite_element_wise array_api_compat.numpy, so `xp is np` is false and it takes the any(isinf) /
 * any(isnan) branch: NaN wins wherever it sits; the Cython scan of sklearn/utils/_isfinite.pyx, which would report the FIRST
 * non-finite value in row-major order, is not reached.  Only the message text depends on it).  Returns 0,
 * ORC_E_NONFINITE_NAN or ORC_E_NONFINITE_INF.  (The fast path -- isfinite(sum(X)) -- only skips the scan for all-finite
 * input: a sum that overflows sends sklearn to the element-wise test, which then finds nothing.) */
int orc_check_finite(const double *pts, int n)
{
    int has_inf = 0;
    for (size_t i = 0; i < (size_t)(n > 0 ? n : 0) * 8; i++) {
        if (isnan(pts[i])) return ORC_E_NONFINITE_NAN;
        if (isinf(pts[i])) has_inf = 1;
    }
    return has_inf ? ORC_E_NONFINITE_INF : 0;
}

/* DBSCAN.fit (sklearn/cluster/_dbscan.py:410-432) + dbscan_inner (_dbscan_inner.pyx).  Returns the number of clusters, or
 * ORC_E_NONFINITE_* (< 0) where sklearn raises on a NaN / infinite coordinate (labels untouched). */
int orc_dbscan(const orc_config *cfg, const double *pts, int n, double eps, int min_samples, int32_t *labels)
{
    uint8_t *adj, *core;
    int *stack;
    int sp = 0, label_num = 0;
    size_t cap;
    if (n <= 0) return 0;
    {
        const int nf = orc_check_finite(pts, n);
        if (nf) return nf;
    }
    adj = (uint8_t *)malloc((size_t)n * (size_t)n);
    core = (uint8_t *)malloc((size_t)n);
    orc_dbscan_neighbors(cfg, pts, n, eps, adj);
    for (int i = 0; i < n; i++) {
        int cnt = 0;
        for (int j = 0; j < n; j++) cnt += adj[(size_t)i * n + j];
        core[i] = cnt >= min_samples;
        labels[i] = -1;
    }
    cap = (size_t)n * 4 + 16;
    stack = (int *)malloc(sizeof(int) * cap);
    for (int seed = 0; seed < n; seed++) {
        int i = seed;
        if (labels[i] != -1 || !core[i]) continue;
        for (;;) {
            if (labels[i] == -1) {
                labels[i] = label_num;
                if (core[i]) {
                    for (int v = 0; v < n; v++)
                        if (adj[(size_t)i * n + v] && labels[v] == -1) {
                            if ((size_t)sp == cap) { cap *= 2; stack = (int *)realloc(stack, sizeof(int) * cap); }
                            stack[sp++] = v;
                        }
                }
            }
            if (sp == 0) break;
            i = stack[--sp];
        }
        label_num++;
    }
    free(adj); free(core); free(stack);
    return label_num;
}

/* BatchedData.pop_frame Tracking.py:66-71 (called by the dataset pre-processing, preprocessing.py:264) */
void orc_pop_frame(orc_scene *s)
{
    if (s->g_len <= 0) return;
    double *first = s->g_frame[0];
    for (int k = 1; k < s->g_len; k++) { s->g_frame[k - 1] = s->g_frame[k]; s->g_n[k - 1] = s->g_n[k]; }
    s->g_frame[s->g_len - 1] = first;
    s->g_n[s->g_len - 1] = 0;
    s->g_len--;
}

/* BatchedData.change_buffer_size on the global ring (Tracking.py:60-64); the deque keeps maxlen FB_FRAMES_BATCH + 1 */
int orc_set_batch_size(orc_scene *s, int new_size)
{
    if (new_size < 1) return -1;
    s->g_size = new_size > s->ring_size ? s->ring_size : new_size;
    return 0;
}
/* BatchedData(init_data) (Tracking.py:38-41): one frame holding rows[n][8] */
int orc_set_batch_frame(orc_scene *s, const double *rows, int n)
{
    if (n < 0 || n > s->max_pts) return -1;
    s->g_len = 1;
    memset(s->g_n, 0, sizeof(s->g_n));
    s->g_n[0] = n;
    if (n > 0) memcpy(s->g_frame[0], rows, sizeof(double) * 8 * (size_t)n);
    return 0;
}

/* ------------------------------------------------------------------ */
/* BatchedData.add_frame on the global ring Tracking.py:43-51 */
static void global_ring_push(orc_scene *s, const double *rows, int n)
{
    while (s->g_len >= s->g_size && s->g_len > 0) {
        double *first = s->g_frame[0];
        for (int k = 1; k < s->g_len; k++) { s->g_frame[k - 1] = s->g_frame[k]; s->g_n[k - 1] = s->g_n[k]; }
        s->g_frame[s->g_len - 1] = first;
        s->g_len--;
    }
    memcpy(s->g_frame[s->g_len], rows, sizeof(double) * 8 * (size_t)n);
    s->g_n[s->g_len] = n;
    s->g_len++;
}

/* TrackBuffer.track Tracking.py:664-703 */
int orc_track_frame(orc_scene *s, const double *pts, int n, double dt, int32_t *assoc,
                    int32_t *db_labels, int32_t *db_n)
{
    const orc_config *c = &s->cfg;
    int dx = c->dim_x;
    int T = s->n_tracks;
    int rc = 0;
    if (n > s->max_pts || n < 0) return -1;
    *db_n = -1;
    s->inner_calls = 0;

    /* _predict_all Tracking.py:591-596 */
    for (int j = 0; j < T; j++) kf_predict(c, &s->tracks[j], s->tracks[j].lifetime + dt);

    /* _calc_dist_fun Tracking.py:530-574 */
    {
        double *best = s->concat; /* scratch: best d^2 per point */
        for (int i = 0; i < n; i++) assoc[i] = -1;
        for (int j = 0; j < T; j++) {
            trk_t *t = &s->tracks[j];
            double C[36], Ci[36], det, logdet;
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++) {
                    double rm = 0.0;
                    if (a == b) { double h = t->spread_est[a] / 2; rm = h * h; }
                    C[a * 6 + b] = (t->P[a * dx + b] + rm) + t->gd[a * 6 + b];
                }
            if (lu6(C, Ci, &det) != 0) return -2;
            logdet = orc_log(fabs(det));
            for (int i = 0; i < n; i++) {
                double y[6], q, d;
                for (int a = 0; a < 6; a++) y[a] = pts[i * 8 + a] - t->x[a];
                /* y' C^-1 y as k-ordered FUSED multiply-add chains (one rounding per step, like the dot kernels of the
                 * BLAS numpy calls at Tracking.py:556-560): v_k = fma(y_a, Ci[a][k], v_k), q = fma(v_k, y_k, q).
                 * The kernels evaluate the same chain (k_track.hip); built with -mfma so fma() is one instruction. */
                q = 0;
                for (int k = 0; k < 6; k++) {
                    double v = y[0] * Ci[k];
                    for (int a = 1; a < 6; a++) v = fma(y[a], Ci[a * 6 + k], v);
                    if (k == 0) q = v * y[0]; else q = fma(v, y[k], q);
                }
                d = logdet + q;
                if (d < c->tr_gate) {
                    if (assoc[i] < 0) { assoc[i] = j; best[i] = d; }
                    else if (d < best[i]) { assoc[i] = j; best[i] = d; }
                }
            }
        }
    }

    /* _get_gated_clouds + _associate_points_to_tracks Tracking.py:605-662 */
    for (int j = 0; j < T; j++) {
        trk_t *t = &s->tracks[j];
        int m = 0;
        for (int i = 0; i < n; i++)
            if (assoc[i] == j) { memcpy(s->cloud + (size_t)m * 8, pts + (size_t)i * 8, 64); m++; }
        if (m == 0) {
            t->lifetime += dt;
        } else {
            t->lifetime = 0.0;
            rc = track_associate(s, t, s->cloud, m);
            if (rc) return rc;
            if (c->seek_inner) {   /* Tracking.py:656 */
                rc = seek_inner(s, t, j, s->cloud, m);
                if (rc) return rc;
            }
        }
    }
    /* Tracking.py:658-660: a track for every inner cluster found, appended BEFORE _maintain_tracks / _update_all */
    for (int k = 0; k < s->inner_calls; k++) {
        if (s->inner_m[k] == 0) continue;
        if (s->n_tracks >= c->track_cap) return -4;
        track_init(s, &s->tracks[s->n_tracks], s->inner_rows + (size_t)k * (size_t)s->ring_size * s->cfg.ring_rows * 8, s->inner_m[k]);
        s->n_tracks++;
    }
    T = s->n_tracks;

    /* _maintain_tracks Tracking.py:513-528 */
    {
        int w = 0;
        for (int j = 0; j < T; j++) {
            trk_t *t = &s->tracks[j];
            double lim = t->is_static ? c->tr_lifetime_static : c->tr_lifetime_dynamic;
            if (t->lifetime > lim) continue;
            if (w != j) { trk_t tmp = s->tracks[w]; s->tracks[w] = s->tracks[j]; s->tracks[j] = tmp; }
            w++;
        }
        s->n_tracks = T = w;
    }

    /* _update_all Tracking.py:598-603 */
    for (int j = 0; j < T; j++) {
        rc = kf_update(c, &s->tracks[j]);
        if (rc) return rc;
    }

    /* batch.add_frame(unassigned) + DBSCAN trigger Tracking.py:689-703 */
    {
        int m = 0, U = 0;
        for (int i = 0; i < n; i++)
            if (assoc[i] < 0) { memcpy(s->cloud + (size_t)m * 8, pts + (size_t)i * 8, 64); m++; }
        global_ring_push(s, s->cloud, m);
        for (int k = 0; k < s->g_len; k++) {
            memcpy(s->concat + (size_t)U * 8, s->g_frame[k], sizeof(double) * 8 * (size_t)s->g_n[k]);
            U += s->g_n[k];
        }
        if (U > 0 && T < c->tr_max_tracks) {
            int ncl = orc_dbscan(c, s->concat, U, c->db_eps, c->db_min_samples, db_labels);
            if (ncl < 0) {   /* sklearn raised (a NaN / inf in the ring): track() ends here with the frame in the ring, nothing clustered,
                                nothing cleared -- and again on every frame the row stays in the ring and the trigger holds */
                *db_n = ORC_DB_RAISED;
                return ncl;
            }
            *db_n = U;
            if (ncl > 0) {
                s->g_len = 0; /* batch.clear() Tracking.py:53-58 */
                /* _add_tracks Tracking.py:576-589: clusters in ascending label order, rows in input order */
                for (int k = 0; k < ncl; k++) {
                    int m2 = 0;
                    if (s->n_tracks >= c->track_cap) return -4;
                    for (int i = 0; i < U; i++)
                        if (db_labels[i] == k) { memcpy(s->cloud + (size_t)m2 * 8, s->concat + (size_t)i * 8, 64); m2++; }
                    track_init(s, &s->tracks[s->n_tracks], s->cloud, m2);
                    s->n_tracks++;
                }
            }
        }
    }
    return s->overflow ? -4 : 0;
}

int orc_num_tracks(const orc_scene *s) { return s->n_tracks; }

int orc_get_inner(const orc_scene *s, int32_t *track, int32_t *n, int32_t *labels, int cap_calls, int cap_labels)
{
    const size_t um = (size_t)s->ring_size * s->cfg.ring_rows;
    int off = 0;
    for (int k = 0; k < s->inner_calls && k < cap_calls; k++) {
        if (track) track[k] = s->inner_track[k];
        if (n) n[k] = s->inner_n[k];
        if (labels && off + s->inner_n[k] <= cap_labels) memcpy(labels + off, s->inner_labels + (size_t)k * um, sizeof(int32_t) * (size_t)s->inner_n[k]);
        off += s->inner_n[k];
    }
    return s->inner_calls;
}

int orc_get_track_ring_size(const orc_scene *s, int t) { return (t >= 0 && t < s->n_tracks) ? s->tracks[t].ring_size : -1; }

int orc_get_tracks(const orc_scene *s, orc_track_record *out, int cap)
{
    int dx = s->cfg.dim_x;
    int n = s->n_tracks < cap ? s->n_tracks : cap;
    for (int j = 0; j < n; j++) {
        const trk_t *t = &s->tracks[j];
        orc_track_record *o = &out[j];
        memset(o, 0, sizeof(*o));
        for (int i = 0; i < dx; i++) o->x[i] = t->x[i];
        for (int i = 0; i < dx; i++)
            for (int k = 0; k < dx; k++) o->P[i * 9 + k] = t->P[i * dx + k];
        memcpy(o->centroid, t->centroid, sizeof(o->centroid));
        memcpy(o->min_vals, t->minv, sizeof(o->min_vals));
        memcpy(o->max_vals, t->maxv, sizeof(o->max_vals));
        memcpy(o->spread_est, t->spread_est, sizeof(o->spread_est));
        memcpy(o->group_disp_est, t->gd, sizeof(o->group_disp_est));
        o->n_est = t->n_est;
        o->lifetime = t->lifetime;
        o->point_num = t->point_num;
        o->is_static = t->is_static;
        o->ring_len = t->ring_len;
        for (int k = 0; k < t->ring_len; k++) o->ring_n[k] = t->ring_n[k];
        memcpy(o->keypoints, t->keypoints, sizeof(o->keypoints));
    }
    return s->n_tracks;
}

int orc_get_batch_ring(const orc_scene *s, int32_t *ring_n)
{
    for (int k = 0; k < s->g_len; k++) ring_n[k] = s->g_n[k];
    return s->g_len;
}

int orc_get_track_ring_frame(const orc_scene *s, int t, int k, double *rows, int cap_rows)
{
    const trk_t *tr;
    int keep;
    if (t < 0 || t >= s->n_tracks) return -1;
    tr = &s->tracks[t];
    if (k < 0 || k >= tr->ring_len) return -1;
    keep = tr->ring_n[k] < s->cfg.ring_rows ? tr->ring_n[k] : s->cfg.ring_rows;
    if (keep > cap_rows) keep = cap_rows;
    memcpy(rows, tr->ring[k], sizeof(double) * 8 * (size_t)keep);
    return keep;
}

/* relative_coordinates + format_single_frame Utils.py:437-520, gathered over
 * tracks as TrackBuffer.estimate_posture does (Tracking.py:718-728).  Ties in x
 * are ordered by row position (np.argsort's default sort is not stable; ties
 * among real rows do not occur with continuous data, zero-pad rows are identical). */
int orc_features(const orc_scene *s, float *feat, int32_t *owner)
{
    const orc_config *c = &s->cfg;
    int count = 0;
    size_t per = (size_t)s->ring_size * 64 * 5;
    for (int j = 0; j < s->n_tracks; j++) {
        const trk_t *t = &s->tracks[j];
        int total = 0;
        float *dst = feat + per * (size_t)count;
        for (int k = 0; k < t->ring_len; k++) total += t->ring_n[k];
        if (!(total > c->model_min_input)) continue;
        memset(dst, 0, per * sizeof(float));
        for (int k = 0; k < t->ring_len; k++) {
            double rows[64][5];
            int ord[64];
            int m = t->ring_n[k] < 64 ? t->ring_n[k] : 64;
            memset(rows, 0, sizeof(rows));
            for (int r = 0; r < m; r++) {
                const double *p = t->ring[k] + (size_t)r * 8;
                rows[r][0] = p[0] - t->centroid[0];
                rows[r][1] = p[1] - t->centroid[1];
                rows[r][2] = p[2] - 0;
                rows[r][3] = p[6] - 0;
                rows[r][4] = ((p[7] - 0) - c->intensity_mu) / c->intensity_std;
            }
            for (int r = 0; r < 64; r++) ord[r] = r;
            for (int a = 1; a < 64; a++) { /* stable insertion sort on x */
                int v = ord[a], b = a - 1;
                while (b >= 0 && rows[ord[b]][0] > rows[v][0]) { ord[b + 1] = ord[b]; b--; }
                ord[b + 1] = v;
            }
            for (int r = 0; r < 64; r++)
                for (int q = 0; q < 5; q++) dst[((size_t)k * 64 + r) * 5 + q] = (float)rows[ord[r]][q];
        }
        owner[count] = j;
        count++;
    }
    return count;
}

int orc_set_keypoints(orc_scene *s, const float *kp, const int32_t *owner, int count)
{
    for (int i = 0; i < count; i++) {
        if (owner[i] < 0 || owner[i] >= s->n_tracks) return -1;
        memcpy(s->tracks[owner[i]].keypoints, kp + (size_t)i * ORC_NKP, sizeof(float) * ORC_NKP);
    }
    return 0;
}

/* normalize_data + point_transform_to_standard_axis Utils.py:294-434 */
int orc_normalize(const orc_config *c, const double *raw, int n, double *out)
{
    int m = 0;
    for (int i = 0; i < n; i++) {
        double x = raw[i * 5], y = raw[i * 5 + 1], z = raw[i * 5 + 2], dop = raw[i * 5 + 3], pk = raw[i * 5 + 4];
        double r = sqrt((x * x + y * y) + z * z);
        double vx, vy, vz, o[8];
        if (r == 0) { vx = 0; vy = dop; vz = 0; }
        else { vx = dop * x / r; vy = dop * y / r; vz = dop * z / r; }
        o[0] = x;
        o[1] = c->tilt_cos * y + (-c->tilt_sin) * z;
        o[2] = (c->tilt_sin * y + c->tilt_cos * z) + c->s_height;
        o[3] = vx;
        o[4] = c->tilt_cos * vy + (-c->tilt_sin) * vz;
        o[5] = c->tilt_sin * vy + c->tilt_cos * vz;
        o[6] = dop;
        o[7] = pk;
        /* Non-finite rows.  point_transform_to_standard_axis multiplies full homogeneous 4-vectors by full 4x4 matrices
         * (Utils.py:311-326: np.dot(T, np.dot(R_inv, [x, y, z, 1]))), zeros included: a NaN or an infinite coordinate meets a
         * zero in the last row of R_inv (0 * inf = NaN), that NaN meets every row of T -- so ONE non-finite coordinate makes
         * all three transformed coordinates NaN, and the scene filter (NaN compares false, Utils.py:422-427) drops the row.
         * The velocity vector [vx, vy, vz, 0] goes the same way: one non-finite component (a NaN / infinite doppler) makes
         * all three NaN; the row is kept with them when its coordinates pass.  (For finite rows the zero terms change nothing.) */
        if (!isfinite(x) || !isfinite(y) || !isfinite(z)) o[0] = o[1] = o[2] = NAN;
        if (!isfinite(vx) || !isfinite(vy) || !isfinite(vz)) o[3] = o[4] = o[5] = NAN;
        if (o[2] <= 2.5 && o[2] > 0 && o[1] > 0) { memcpy(out + (size_t)m * 8, o, sizeof(o)); m++; }
    }
    return m;
}

/* ------------------------------------------------------------------ */
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int orc_batch_run_f32(orc_scene **scenes, int n_scenes, int max_pts, int n_frames, const float *pts,
                      const int32_t *n, const double *dt, int n_threads)
{
    int err = 0;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : omp_get_max_threads())
#endif
    {
        double *buf = (double *)malloc(sizeof(double) * 8 * (size_t)max_pts);
        int32_t *assoc = (int32_t *)malloc(sizeof(int32_t) * (size_t)max_pts);
        int32_t *labels = (int32_t *)malloc(sizeof(int32_t) * (size_t)max_pts * ORC_RING_MAX);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int s = 0; s < n_scenes; s++) {
            for (int f = 0; f < n_frames; f++) {
                const float *src = pts + ((size_t)f * n_scenes + s) * (size_t)max_pts * 8;
                const int cnt = n[(size_t)f * n_scenes + s];
                int32_t dbn;
                int rc;
                if (cnt <= 0) continue; /* offline_main.py:56 */
                for (int e = 0; e < cnt * 8; e++) buf[e] = (double)src[e];
                rc = orc_track_frame(scenes[s], buf, cnt, dt[(size_t)f * n_scenes + s], assoc, labels, &dbn);
                if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
                    err = rc;
                }
            }
        }
        free(buf); free(assoc); free(labels);
    }
    return err;
}

int orc_batch_track(orc_scene **scenes, int n_scenes, int max_pts, const double *pts,
                    const int32_t *n, const double *dt, int32_t *assoc, int32_t *db_labels,
                    int32_t *db_n, int n_threads)
{
    int err = 0;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : omp_get_max_threads())
#endif
    for (int s = 0; s < n_scenes; s++) {
        int ring = scenes[s]->ring_size;
        int rc = orc_track_frame(scenes[s], pts + (size_t)s * max_pts * 8, n[s], dt[s],
                                 assoc + (size_t)s * max_pts,
                                 db_labels + (size_t)s * ring * max_pts, db_n + s);
        if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
            err = rc;
        }
    }
    return err;
}
