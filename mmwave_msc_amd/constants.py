"""Config surface of the drop-in: the module-level names the reference keeps in
`src/constants.py` (imported everywhere as `const`), with the same defaults.

Only the names the hot path reads are honoured by the GPU path (marked [hot]);
the rest are carried so code written against the reference module keeps
importing.  `to_config()` snapshots the current values into the C-ABI struct --
like the reference's default-argument binding (Utils.py:250, 468-470), values
are captured when a TrackBuffer is created, not re-read per frame.
"""
import numpy as np

from . import _lib

PIXEL_TO_METERS = 0.000265

# ---- flags / paths (reference constants.py:8-27; not used by the GPU path) ----
PROFILING = False
SCREEN_CONNECTED = False
P_CONFIG_PATH = "./config_cases/our_config_8.5m.cfg"
P_MODEL_PATH = "../trained_cases/Our_system/model/MARS.h5"
P_DATA_PATH = "./dataset"
P_LOG_PATH = f"{P_DATA_PATH}/log"
P_PREPROCESS_PATH = f"{P_DATA_PATH}/preprocessed"
P_FORMATTED_PATH = f"{P_DATA_PATH}/formatted"
P_KINECT_DIR = "/kinect/"
P_MMWAVE_DIR = "/mmWave/"
P_PROFILING_PATH = "./profiling/"
P_CLI_PORT = "/dev/ttyACM0"
P_DATA_PORT = "/dev/ttyACM1"

# ---- scene geometry (constants.py:29-54) ----
M_X, M_Y, M_Z = 0.32, -0.6, 1.3
SCREEN_SIZE = [1.6, 1.1]
SCREEN_HEIGHT = 1.3
S_HEIGHT = 1.8          # [hot] sensor height, normalize_data
S_TILT = -5             # [hot] sensor tilt in degrees, normalize_data
V_SCALLING = 1
V_3D_AXIS = [[-2.5, 2.5], [0, 5], [0, 3]]
V_SCREEN_FADE_SIZE_MAX = 0.3
V_SCREEN_FADE_SIZE_MIN = 0.2
V_SCREEN_FADE_WEIGHT = 0.08
V_BBOX_HEIGHT = 1.8
V_BBOX_EYESIGHT_HEIGHT = 1.75

# ---- experiment logging (constants.py:57-61) ----
FB_FRAMES_SKIP = 0
FB_EXPERIMENT_FILE_SIZE = 200
FB_WRITE_BUFFER_SIZE = 40
FB_READ_BUFFER_SIZE = 40    # [hot] OfflineManager refill size

# ---- clustering (constants.py:64-80) ----
FB_FRAMES_BATCH = 2          # [hot] ring length - 1
FB_FRAMES_BATCH_STATIC = 2
DB_Z_WEIGHT = 0.4            # [hot]
DB_RANGE_WEIGHT = 0.03       # [hot]
DB_EPS = 0.3                 # [hot]
DB_MIN_SAMPLES_MIN = 35      # [hot]
DB_POINTS_THRES = 40
DB_SPREAD_THRES = 0.7
DB_INNER_EPS = 0.1
DB_INNER_MIN_SAMPLES = 8
DB_MIN_SAMPLES_MAX = 25

# ---- tracking / Kalman (constants.py:83-104) ----
TR_MAX_TRACKS = 4            # [hot]
TR_LIFETIME_DYNAMIC = 3      # [hot] s
TR_LIFETIME_STATIC = 7       # [hot] s
TR_VEL_THRES = 0.12          # [hot]
TR_GATE = 4.5                # [hot]
KF_R_STD = 0.1
KF_Q_STD = 1                 # [hot]
KF_P_INIT = 0.1              # [hot]
KF_GROUP_DISP_EST_INIT = 0.1  # [hot]
KF_ENABLE_EST = False        # [hot]
KF_A_N = 0.9                 # [hot]
KF_EST_POINTNUM = 10         # [hot]
KF_SPREAD_LIM = [0.2, 0.2, 2, 1.2, 1.2, 0.2]  # [hot]
KF_A_SPR = 0.9               # [hot]

# ---- model (constants.py:106-172) ----
INTENSITY_MU = 27.0187       # [hot]
INTENSITY_STD = 70.351       # [hot]
MODEL_MIN_INPUT = 0          # [hot]
MODEL_DEFAULT_POSTURE = np.array([float(v) for v in (
    "0.0000 -0.0007 -0.0006 -0.0038 -0.1820 -0.2540 -0.2579 0.1830 0.2957 0.2940 -0.0805 -0.1141 "
    "-0.1232 -0.1358 0.0796 0.1436 0.1558 0.1720 -0.0007 0.7699 1.0906 1.4020 1.5513 1.2893 1.0360 "
    "0.7994 1.2865 1.0483 0.8117 0.7670 0.3428 0.0000 -0.0746 0.7713 0.3706 -0.0128 -0.0796 1.3255 "
    "0.0752 0.0533 0.0203 0.0000 0.0496 0.1350 0.1303 0.0345 0.1277 0.1050 0.0392 0.0533 0.0786 "
    "-0.0056 0.0346 -0.0007 0.0683 -0.0082 0.0312").split()])   # [hot] 19 joints: x0..x18 | y0..y18 | z0..z18


def _white_noise_block(dt, var):
    """filterpy.common.Q_discrete_white_noise(dim=3, dt, var) (absent from this image)."""
    return np.array([[0.25 * dt**4, 0.5 * dt**3, 0.5 * dt**2],
                     [0.5 * dt**3, dt**2, dt],
                     [0.5 * dt**2, dt, 1.0]]) * var


def _block_diag(blocks):
    n = sum(b.shape[0] for b in blocks)
    out = np.zeros((n, n))
    o = 0
    for b in blocks:
        out[o:o + b.shape[0], o:o + b.shape[0]] = b
        o += b.shape[0]
    return out


class CONST_ACC_MODEL:
    """9-state constant-acceleration model (reference constants.py:176-215)."""
    KF_DIM = [9, 6]
    KF_H = np.eye(6, 9)

    @staticmethod
    def STATE_VEC(init):
        return [init[k] for k in range(6)] + [0, 0, 0]

    @staticmethod
    def KF_F(dt):
        f = np.eye(9)
        for i in range(6):
            f[i, i + 3] = dt
        for i in range(3):
            f[i, i + 6] = 0.5 * dt**2
        return f

    @staticmethod
    def KF_Q_DISCR(dt):
        return _block_diag([_white_noise_block(dt, KF_Q_STD)] * 3)


class CONST_VEL_MODEL:
    """6-state constant-velocity model (reference constants.py:218-243)."""
    KF_DIM = [6, 6]
    KF_H = np.eye(6)

    @staticmethod
    def STATE_VEC(init):
        return [init[k] for k in range(6)]

    @staticmethod
    def KF_F(dt):
        f = np.eye(6)
        for i in range(3):
            f[i, i + 3] = dt
        return f

    @staticmethod
    def KF_Q_DISCR(dt):
        return _block_diag([_white_noise_block(dt, KF_Q_STD)] * 2)


MOTION_MODEL = CONST_ACC_MODEL   # [hot]


def to_config(**overrides) -> "_lib.MmwConfig":
    """Snapshot the current module values into a `mmw_config` (include/mmw.h)."""
    g = globals()
    kw = dict(
        fb_frames_batch=int(g["FB_FRAMES_BATCH"]), db_min_samples=int(g["DB_MIN_SAMPLES_MIN"]),
        tr_max_tracks=int(g["TR_MAX_TRACKS"]), kf_enable_est=int(bool(g["KF_ENABLE_EST"])),
        model_min_input=int(g["MODEL_MIN_INPUT"]), dim_x=int(g["MOTION_MODEL"].KF_DIM[0]),
        db_z_weight=float(g["DB_Z_WEIGHT"]), db_range_weight=float(g["DB_RANGE_WEIGHT"]), db_eps=float(g["DB_EPS"]),
        tr_lifetime_dynamic=float(g["TR_LIFETIME_DYNAMIC"]), tr_lifetime_static=float(g["TR_LIFETIME_STATIC"]),
        tr_vel_thres=float(g["TR_VEL_THRES"]), tr_gate=float(g["TR_GATE"]), kf_q_std=float(g["KF_Q_STD"]),
        kf_p_init=float(g["KF_P_INIT"]), kf_group_disp_est_init=float(g["KF_GROUP_DISP_EST_INIT"]),
        kf_a_n=float(g["KF_A_N"]), kf_est_pointnum=float(g["KF_EST_POINTNUM"]),
        kf_spread_lim=list(g["KF_SPREAD_LIM"]), kf_a_spr=float(g["KF_A_SPR"]),
        intensity_mu=float(g["INTENSITY_MU"]), intensity_std=float(g["INTENSITY_STD"]),
        s_height=float(g["S_HEIGHT"]), s_tilt=float(g["S_TILT"]),
        default_posture=[float(v) for v in np.asarray(g["MODEL_DEFAULT_POSTURE"], dtype=np.float32)],
        db_points_thres=int(g["DB_POINTS_THRES"]), fb_frames_batch_static=int(g["FB_FRAMES_BATCH_STATIC"]),
        db_spread_thres=float(g["DB_SPREAD_THRES"]), db_inner_eps=float(g["DB_INNER_EPS"]),
        m_x=float(g["M_X"]), m_y=float(g["M_Y"]), m_z=float(g["M_Z"]),
        v_screen_fade_size_max=float(g["V_SCREEN_FADE_SIZE_MAX"]), v_screen_fade_size_min=float(g["V_SCREEN_FADE_SIZE_MIN"]),
        v_screen_fade_weight=float(g["V_SCREEN_FADE_WEIGHT"]),
    )
    kw.update(overrides)
    return _lib.default_config(**kw)
