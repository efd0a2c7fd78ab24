"""ctypes binding of the plain-C oracle (oracle/c/mmw_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  The library is built by
`make -C oracle/c` (also done by `__graft_entry__.build()`).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
RING_MAX = 4
NKP = 57


class OrcConfig(C.Structure):
    _fields_ = [
        ("fb_frames_batch", C.c_int32),
        ("db_min_samples", C.c_int32),
        ("tr_max_tracks", C.c_int32),
        ("kf_enable_est", C.c_int32),
        ("model_min_input", C.c_int32),
        ("dim_x", C.c_int32),
        ("ring_rows", C.c_int32),
        ("track_cap", C.c_int32),
        ("db_z_weight", C.c_double),
        ("db_range_weight", C.c_double),
        ("db_eps", C.c_double),
        ("tr_lifetime_dynamic", C.c_double),
        ("tr_lifetime_static", C.c_double),
        ("tr_vel_thres", C.c_double),
        ("tr_gate", C.c_double),
        ("kf_q_std", C.c_double),
        ("kf_p_init", C.c_double),
        ("kf_group_disp_est_init", C.c_double),
        ("kf_a_n", C.c_double),
        ("kf_est_pointnum", C.c_double),
        ("kf_spread_lim", C.c_double * 6),
        ("kf_a_spr", C.c_double),
        ("intensity_mu", C.c_double),
        ("intensity_std", C.c_double),
        ("s_height", C.c_double),
        ("tilt_cos", C.c_double),
        ("tilt_sin", C.c_double),
        ("default_posture", C.c_float * NKP),
        ("seek_inner", C.c_int32),
        ("db_points_thres", C.c_int32),
        ("fb_frames_batch_static", C.c_int32),
        ("db_spread_thres", C.c_double),
        ("db_inner_eps", C.c_double),
    ]


class OrcTrackRecord(C.Structure):
    _fields_ = [
        ("x", C.c_double * 9),
        ("P", C.c_double * 81),
        ("centroid", C.c_double * 6),
        ("min_vals", C.c_double * 6),
        ("max_vals", C.c_double * 6),
        ("spread_est", C.c_double * 6),
        ("group_disp_est", C.c_double * 36),
        ("n_est", C.c_double),
        ("lifetime", C.c_double),
        ("point_num", C.c_int32),
        ("is_static", C.c_int32),
        ("ring_len", C.c_int32),
        ("ring_n", C.c_int32 * RING_MAX),
        ("keypoints", C.c_float * NKP),
    ]


TRACK_DTYPE = np.dtype(
    [
        ("x", "f8", (9,)), ("P", "f8", (9, 9)), ("centroid", "f8", (6,)),
        ("min_vals", "f8", (6,)), ("max_vals", "f8", (6,)), ("spread_est", "f8", (6,)),
        ("group_disp_est", "f8", (6, 6)), ("n_est", "f8"), ("lifetime", "f8"),
        ("point_num", "i4"), ("is_static", "i4"), ("ring_len", "i4"),
        ("ring_n", "i4", (RING_MAX,)), ("keypoints", "f4", (NKP,)),
    ],
    align=True,
)
assert TRACK_DTYPE.itemsize == C.sizeof(OrcTrackRecord), (TRACK_DTYPE.itemsize, C.sizeof(OrcTrackRecord))

_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "c", "mmw_oracle.c")
    if force or not os.path.isfile(LIB_PATH) or (
        os.path.isfile(src) and os.path.getmtime(src) > os.path.getmtime(LIB_PATH)
    ):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "c")], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    vp, i32p, f64p, f32p, u8p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    L.orc_config_default.argtypes = [C.POINTER(OrcConfig)]
    L.orc_scene_new.argtypes = [C.POINTER(OrcConfig), C.c_int]
    L.orc_scene_new.restype = vp
    L.orc_scene_free.argtypes = [vp]
    L.orc_scene_reset.argtypes = [vp]
    L.orc_track_frame.argtypes = [vp, f64p, C.c_int, C.c_double, i32p, i32p, i32p]
    L.orc_num_tracks.argtypes = [vp]
    L.orc_get_tracks.argtypes = [vp, C.c_void_p, C.c_int]
    L.orc_get_batch_ring.argtypes = [vp, i32p]
    L.orc_set_batch_size.argtypes = [vp, C.c_int]
    L.orc_set_batch_frame.argtypes = [vp, f64p, C.c_int]
    L.orc_get_inner.argtypes = [vp, i32p, i32p, i32p, C.c_int, C.c_int]
    L.orc_get_track_ring_size.argtypes = [vp, C.c_int]
    L.orc_pop_frame.argtypes = [vp]
    L.orc_pop_frame.restype = None
    L.orc_get_track_ring_frame.argtypes = [vp, C.c_int, C.c_int, f64p, C.c_int]
    L.orc_features.argtypes = [vp, f32p, i32p]
    L.orc_set_keypoints.argtypes = [vp, f32p, i32p, C.c_int]
    L.orc_normalize.argtypes = [C.POINTER(OrcConfig), f64p, C.c_int, f64p]
    L.orc_dbscan.argtypes = [C.POINTER(OrcConfig), f64p, C.c_int, C.c_double, C.c_int, i32p]
    L.orc_dbscan_neighbors.argtypes = [C.POINTER(OrcConfig), f64p, C.c_int, C.c_double, u8p]
    L.orc_log.argtypes = [C.c_double]
    L.orc_log.restype = C.c_double
    L.orc_np_pairwise_sum.argtypes = [f64p, C.c_int]
    L.orc_np_pairwise_sum.restype = C.c_double
    L.orc_batch_track.argtypes = [C.POINTER(vp), C.c_int, C.c_int, f64p, i32p, f64p, i32p, i32p, i32p, C.c_int]
    L.orc_max_threads.restype = C.c_int
    L.orc_batch_run_f32.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, f32p, i32p, f64p, C.c_int]
    _lib = L
    return L


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def default_config(**overrides) -> OrcConfig:
    cfg = OrcConfig()
    lib().orc_config_default(C.byref(cfg))
    for k, v in overrides.items():
        if k == "kf_spread_lim":
            for i in range(6):
                cfg.kf_spread_lim[i] = float(v[i])
        elif k == "default_posture":
            for i in range(NKP):
                cfg.default_posture[i] = float(v[i])
        elif k == "s_tilt":
            ang = np.radians(v)
            cfg.tilt_cos, cfg.tilt_sin = float(np.cos(ang)), float(np.sin(ang))
        else:
            if not hasattr(cfg, k):
                raise AttributeError(k)
            setattr(cfg, k, v)
    return cfg


def config_from_constants(const, **overrides) -> OrcConfig:
    """Build an OrcConfig from a `constants`-like module (reference constants.py)."""
    kw = dict(
        fb_frames_batch=int(const.FB_FRAMES_BATCH), db_min_samples=int(const.DB_MIN_SAMPLES_MIN),
        tr_max_tracks=int(const.TR_MAX_TRACKS), kf_enable_est=int(bool(const.KF_ENABLE_EST)),
        model_min_input=int(const.MODEL_MIN_INPUT), dim_x=int(const.MOTION_MODEL.KF_DIM[0]),
        db_z_weight=float(const.DB_Z_WEIGHT), db_range_weight=float(const.DB_RANGE_WEIGHT),
        db_eps=float(const.DB_EPS), tr_lifetime_dynamic=float(const.TR_LIFETIME_DYNAMIC),
        tr_lifetime_static=float(const.TR_LIFETIME_STATIC), tr_vel_thres=float(const.TR_VEL_THRES),
        tr_gate=float(const.TR_GATE), kf_q_std=float(const.KF_Q_STD), kf_p_init=float(const.KF_P_INIT),
        kf_group_disp_est_init=float(const.KF_GROUP_DISP_EST_INIT), kf_a_n=float(const.KF_A_N),
        kf_est_pointnum=float(const.KF_EST_POINTNUM), kf_spread_lim=list(const.KF_SPREAD_LIM),
        kf_a_spr=float(const.KF_A_SPR), intensity_mu=float(const.INTENSITY_MU),
        intensity_std=float(const.INTENSITY_STD), s_height=float(const.S_HEIGHT),
        s_tilt=float(const.S_TILT), default_posture=list(np.asarray(const.MODEL_DEFAULT_POSTURE, dtype=np.float32)),
        db_points_thres=int(const.DB_POINTS_THRES), fb_frames_batch_static=int(const.FB_FRAMES_BATCH_STATIC),
        db_spread_thres=float(const.DB_SPREAD_THRES), db_inner_eps=float(const.DB_INNER_EPS),
    )
    kw.update(overrides)
    return default_config(**kw)


E_NONFINITE_NAN, E_NONFINITE_INF, DB_RAISED = -7, -8, -2   # mmw_oracle.h


class OracleNonFinite(RuntimeError, ValueError):
    """apply_DBscan was reached with a NaN / an infinite value in its cloud: sklearn's input validation raises ValueError
    there (Utils.py:272-278).  `kind` = "NaN" | "infinity" (sklearn's message: NaN when any value is NaN, else infinity); `.rc` = the oracle's return code."""

    def __init__(self, rc, where="orc_track_frame"):
        self.rc = int(rc)
        self.kind = "NaN" if rc == E_NONFINITE_NAN else "infinity"
        super().__init__(f"{where} failed rc={rc}: " + sklearn_message(self.kind))


def sklearn_message(kind: str) -> str:
    """First line of the ValueError sklearn raises (sklearn/utils/validation.py:_assert_all_finite_element_wise)."""
    return "Input X contains NaN." if kind == "NaN" else "Input X contains infinity or a value too large for dtype('float64')."


class OracleScene:
    """One scene = one reference `TrackBuffer` + its global `BatchedData`."""

    def __init__(self, cfg: OrcConfig, max_pts: int):
        self.L = lib()
        self.cfg = cfg
        self.max_pts = int(max_pts)
        self.ring = cfg.fb_frames_batch + 1
        self.h = self.L.orc_scene_new(C.byref(cfg), self.max_pts)
        if not self.h:
            raise RuntimeError("orc_scene_new failed (bad config)")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_scene_free(self.h)
            self.h = None

    def reset(self):
        self.L.orc_scene_reset(self.h)

    def track(self, pts: np.ndarray, dt: float):
        """Returns (assoc[n] int32, db_labels[U] int32 or None)."""
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 8)
        n = pts.shape[0]
        assoc = np.empty(max(n, 1), dtype=np.int32)
        labels = np.empty(self.ring * self.max_pts, dtype=np.int32)
        dbn = C.c_int32(-1)
        rc = self.L.orc_track_frame(self.h, _p(pts, C.c_double), n, float(dt), _p(assoc, C.c_int32),
                                    _p(labels, C.c_int32), C.byref(dbn))
        self.last_assoc = assoc[:n].copy()   # (valid also when the frame's apply_DBscan raised: the association came first)
        self.last_db_n = int(dbn.value)
        if rc in (E_NONFINITE_NAN, E_NONFINITE_INF):
            raise OracleNonFinite(rc)
        if rc != 0:
            raise RuntimeError(f"orc_track_frame failed rc={rc}")
        return assoc[:n].copy(), (labels[: dbn.value].copy() if dbn.value >= 0 else None)

    @property
    def n_tracks(self) -> int:
        return self.L.orc_num_tracks(self.h)

    def tracks(self) -> np.ndarray:
        n = self.n_tracks
        out = np.zeros(max(n, 1), dtype=TRACK_DTYPE)
        self.L.orc_get_tracks(self.h, out.ctypes.data_as(C.c_void_p), n)
        return out[:n]

    def set_batch_size(self, new_size: int):
        if self.L.orc_set_batch_size(self.h, int(new_size)):
            raise ValueError(new_size)

    def set_batch_frame(self, rows: np.ndarray):
        rows = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 8)
        if self.L.orc_set_batch_frame(self.h, _p(rows, C.c_double), len(rows)):
            raise ValueError(len(rows))

    def inner_calls(self):
        """seek_inner_clusters calls of the last frame: list of (pre-maintenance track position, labels[n])."""
        cap = max(self.cfg.track_cap, 64)
        trk = np.zeros(cap, dtype=np.int32)
        n = np.zeros(cap, dtype=np.int32)
        lab = np.zeros(cap * self.ring * self.ring * self.max_pts, dtype=np.int32)
        k = self.L.orc_get_inner(self.h, _p(trk, C.c_int32), _p(n, C.c_int32), _p(lab, C.c_int32), cap, lab.size)
        out, off = [], 0
        for i in range(k):
            out.append((int(trk[i]), lab[off: off + n[i]].copy()))
            off += int(n[i])
        return out

    def track_ring_size(self, t: int) -> int:
        """track.batch.size (what BatchedData.change_buffer_size sets, Tracking.py:60-64)."""
        return self.L.orc_get_track_ring_size(self.h, int(t))

    def pop_frame(self):
        """BatchedData.pop_frame() (Tracking.py:66-71)."""
        self.L.orc_pop_frame(self.h)

    def batch_ring(self):
        a = np.zeros(RING_MAX, dtype=np.int32)
        k = self.L.orc_get_batch_ring(self.h, _p(a, C.c_int32))
        return a[:k].copy()

    def track_ring_frame(self, t: int, k: int) -> np.ndarray:
        rows = np.zeros((max(self.cfg.ring_rows, 64, self.ring * self.max_pts if self.cfg.seek_inner else 0), 8))
        m = self.L.orc_get_track_ring_frame(self.h, t, k, _p(rows, C.c_double), rows.shape[0])
        if m < 0:
            raise IndexError((t, k))
        return rows[:m].copy()

    def features(self):
        n = self.n_tracks
        shape = (max(n, 1), self.ring, 8, 8, 5) if self.ring > 1 else (max(n, 1), 8, 8, 5)
        feat = np.zeros(shape, dtype=np.float32)
        owner = np.zeros(max(n, 1), dtype=np.int32)
        cnt = self.L.orc_features(self.h, _p(feat, C.c_float), _p(owner, C.c_int32))
        return feat[:cnt].copy(), owner[:cnt].copy()

    def set_keypoints(self, kp: np.ndarray, owner: np.ndarray):
        kp = np.ascontiguousarray(kp, dtype=np.float32)
        owner = np.ascontiguousarray(owner, dtype=np.int32)
        rc = self.L.orc_set_keypoints(self.h, _p(kp, C.c_float), _p(owner, C.c_int32), len(owner))
        if rc:
            raise RuntimeError("orc_set_keypoints failed")


def normalize(cfg: OrcConfig, raw: np.ndarray) -> np.ndarray:
    raw = np.ascontiguousarray(raw, dtype=np.float64).reshape(-1, 5)
    out = np.zeros((max(raw.shape[0], 1), 8))
    m = lib().orc_normalize(C.byref(cfg), _p(raw, C.c_double), raw.shape[0], _p(out, C.c_double))
    return out[:m].copy()


def dbscan(cfg: OrcConfig, pts: np.ndarray, eps=None, min_samples=None) -> np.ndarray:
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 8)
    labels = np.full(max(pts.shape[0], 1), -1, dtype=np.int32)
    rc = lib().orc_dbscan(C.byref(cfg), _p(pts, C.c_double), pts.shape[0],
                          cfg.db_eps if eps is None else float(eps),
                          cfg.db_min_samples if min_samples is None else int(min_samples), _p(labels, C.c_int32))
    if rc < 0:
        raise OracleNonFinite(rc, "orc_dbscan")
    return labels[: pts.shape[0]]


def dbscan_neighbors(cfg: OrcConfig, pts: np.ndarray, eps=None) -> np.ndarray:
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 8)
    n = pts.shape[0]
    adj = np.zeros((n, n), dtype=np.uint8)
    lib().orc_dbscan_neighbors(C.byref(cfg), _p(pts, C.c_double), n, cfg.db_eps if eps is None else float(eps),
                               _p(adj, C.c_uint8))
    return adj


class OracleBatch:
    """S independent scenes stepped together (OpenMP over scenes): the CPU baseline."""

    def __init__(self, cfg: OrcConfig, n_scenes: int, max_pts: int):
        self.L = lib()
        self.scenes = [OracleScene(cfg, max_pts) for _ in range(n_scenes)]
        self.S, self.max_pts, self.ring = n_scenes, max_pts, cfg.fb_frames_batch + 1
        self.handles = (C.c_void_p * n_scenes)(*[s.h for s in self.scenes])
        self.assoc = np.zeros((n_scenes, max_pts), dtype=np.int32)
        self.labels = np.zeros((n_scenes, self.ring * max_pts), dtype=np.int32)
        self.db_n = np.zeros(n_scenes, dtype=np.int32)

    def step(self, pts: np.ndarray, n: np.ndarray, dt: np.ndarray, threads: int = 0):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        n = np.ascontiguousarray(n, dtype=np.int32)
        dt = np.ascontiguousarray(dt, dtype=np.float64)
        assert pts.shape == (self.S, self.max_pts, 8)
        rc = self.L.orc_batch_track(self.handles, self.S, self.max_pts, _p(pts, C.c_double), _p(n, C.c_int32),
                                    _p(dt, C.c_double), _p(self.assoc, C.c_int32), _p(self.labels, C.c_int32),
                                    _p(self.db_n, C.c_int32), int(threads))
        if rc:
            raise RuntimeError(f"orc_batch_track failed rc={rc}")
        return self.assoc, self.labels, self.db_n


def batch_run_f32(batch: "OracleBatch", pts: np.ndarray, n: np.ndarray, dt: np.ndarray, threads: int = 0):
    """All frames of all scenes, scene-major (each thread owns whole scenes).  pts[F,S,N,8] float32."""
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    n = np.ascontiguousarray(n, dtype=np.int32)
    dt = np.ascontiguousarray(dt, dtype=np.float64)
    F, S = pts.shape[0], pts.shape[1]
    assert S == batch.S and pts.shape[2] == batch.max_pts and n.shape == (F, S) and dt.shape == (F, S)
    rc = batch.L.orc_batch_run_f32(batch.handles, S, batch.max_pts, F, _p(pts, C.c_float), _p(n, C.c_int32),
                                   _p(dt, C.c_double), int(threads))
    if rc:
        raise RuntimeError(f"orc_batch_run_f32 failed rc={rc}")


def max_threads() -> int:
    return lib().orc_max_threads()
