"""ctypes binding of libmmw_hip.so (C-ABI declared in include/mmw.h).

There is deliberately no fallback: if the HIP library is missing or cannot be
loaded, importing the compute path raises.  `build()` compiles it in-tree with
hipcc for gfx950 (no GPU needed to compile).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("MMW_LIB_NAME", "libmmw_hip.so"))  # MMW_LIB_NAME: diagnostic builds only
CSRC = os.path.join(_HERE, "csrc")
RING_MAX = 4
NKP = 57

MMW_OK = 0
E_ARG, E_SINGULAR, E_DIVZERO, E_CAPACITY, E_HIP, E_NODEVICE, E_NONFINITE = -1, -2, -3, -4, -5, -6, -7
DB_RAISED = -2     # MMW_DB_RAISED: db_n of a scene whose apply_DBscan call of the frame raised (E_NONFINITE)
ERRBIT_SINGULAR, ERRBIT_DIVZERO, ERRBIT_CAPACITY, ERRBIT_BADCOUNT, ERRBIT_NONFINITE_NAN, ERRBIT_NONFINITE_INF = 1, 2, 4, 8, 16, 32
K_TRACK, K_DBSCAN, K_FEATURES, K_NORMALIZE, K_TABLE, K_PREDICT, K_POST = range(7)

EMPTY_FRAME = -1   # MMW_EMPTY_FRAME (include/mmw.h)
BAD_FRAME = -3     # MMW_BAD_FRAME: n_out of mmw_normalize_tlv for a TLV body it refused to decode

EXPORTS = [
    "mmw_config_default", "mmw_create", "mmw_destroy", "mmw_last_error", "mmw_reset", "mmw_pop_frame", "mmw_set_stream",
    "mmw_synchronize", "mmw_get_dims", "mmw_dev_alloc", "mmw_dev_free", "mmw_memcpy_h2d", "mmw_memcpy_d2h",
    "mmw_normalize", "mmw_step", "mmw_step_host", "mmw_dbscan", "mmw_features", "mmw_set_keypoints", "mmw_check",
    "mmw_get_num_tracks", "mmw_get_tracks", "mmw_get_batch_ring", "mmw_get_track_ring_frame",
    "mmw_get_batch_ring_frame", "mmw_track_table", "mmw_profile_enable", "mmw_profile_reset", "mmw_profile_get",
    "mmw_kernel_name", "mmw_version", "mmw_stats_get", "mmw_stats_reset", "mmw_format_frames", "mmw_stats_get_ext", "mmw_mars_conv3d",
    "mmw_parse_uart", "mmw_features_async", "mmw_features_wait", "mmw_set_keypoints_uid", "mmw_get_inner",
    "mmw_set_batch_size", "mmw_set_batch_frame", "mmw_mars_conv_split", "mmw_mars_dense1_split", "mmw_diag_queue", "mmw_set_chain_side_stream", "mmw_side_workers", "mmw_step_kind", "mmw_streams_concurrent", "mmw_reset_scenes", "mmw_get_errors",
    "mmw_kalman_layout", "mmw_step_f32", "mmw_normalize_f32", "mmw_frame_host", "mmw_mars_head_small", "mmw_mars_range_fixup",
    "mmw_attach_posture", "mmw_frame_posture_host", "mmw_clear_errors", "mmw_stream_wait", "mmw_wait_stream", "mmw_find_tlv", "mmw_normalize_tlv",
]


class MmwPostureModel(C.Structure):
    """struct mmw_posture_model (include/mmw.h): device pointers of define_CNN_3D's weights, BatchNormalization folded."""
    _fields_ = [("conv1_w", C.c_void_p), ("conv1_b", C.c_void_p), ("conv2_w", C.c_void_p), ("conv2_b", C.c_void_p),
                ("dense1_w", C.c_void_p), ("dense1_ld", C.c_int64), ("dense1_b", C.c_void_p), ("dense2_w", C.c_void_p),
                ("dense2_b", C.c_void_p)]


class MmwUartCfg(C.Structure):
    """struct mmw_uart_cfg (include/mmw.h): the scales ReadIWR14xx.__parseConfigFile derives from the radar .cfg."""
    _fields_ = [("range_idx_to_meters", C.c_double), ("doppler_resolution_mps", C.c_double),
                ("num_doppler_bins", C.c_int32), ("reserved", C.c_int32)]


class MmwConfig(C.Structure):
    """struct mmw_config (include/mmw.h) -- mirrors the reference constants.py."""
    _fields_ = [
        ("fb_frames_batch", C.c_int32), ("db_min_samples", C.c_int32), ("tr_max_tracks", C.c_int32),
        ("kf_enable_est", C.c_int32), ("model_min_input", C.c_int32), ("dim_x", C.c_int32),
        ("ring_rows", C.c_int32), ("track_cap", C.c_int32),
        ("db_z_weight", C.c_double), ("db_range_weight", C.c_double), ("db_eps", C.c_double),
        ("tr_lifetime_dynamic", C.c_double), ("tr_lifetime_static", C.c_double), ("tr_vel_thres", C.c_double),
        ("tr_gate", C.c_double), ("kf_q_std", C.c_double), ("kf_p_init", C.c_double),
        ("kf_group_disp_est_init", C.c_double), ("kf_a_n", C.c_double), ("kf_est_pointnum", C.c_double),
        ("kf_spread_lim", C.c_double * 6), ("kf_a_spr", C.c_double), ("intensity_mu", C.c_double),
        ("intensity_std", C.c_double), ("s_height", C.c_double), ("tilt_cos", C.c_double), ("tilt_sin", C.c_double),
        ("default_posture", C.c_float * NKP),
        ("kalman_dense_min_units", C.c_int32), ("seek_inner", C.c_int32), ("db_points_thres", C.c_int32),
        ("fb_frames_batch_static", C.c_int32), ("chain_side_stream", C.c_int32),
        ("db_spread_thres", C.c_double), ("db_inner_eps", C.c_double),
        ("m_x", C.c_double), ("m_y", C.c_double), ("m_z", C.c_double),
        ("v_screen_fade_size_max", C.c_double), ("v_screen_fade_size_min", C.c_double), ("v_screen_fade_weight", C.c_double),
        ("fused_step", C.c_int32), ("reserved_", C.c_int32),
    ]


TRACK_DTYPE = np.dtype(
    [
        ("x", "f8", (9,)), ("P", "f8", (9, 9)), ("centroid", "f8", (6,)), ("min_vals", "f8", (6,)),
        ("max_vals", "f8", (6,)), ("spread_est", "f8", (6,)), ("group_disp_est", "f8", (6, 6)),
        ("n_est", "f8"), ("lifetime", "f8"), ("point_num", "i4"), ("is_static", "i4"), ("ring_len", "i4"),
        ("ring_n", "i4", (RING_MAX,)), ("uid", "i4"), ("keypoints", "f4", (NKP,)),
    ],
    align=True,
)
SUMMARY_DTYPE = np.dtype(
    [("scene", "i4"), ("slot", "i4"), ("alive", "i4"), ("is_static", "i4"), ("point_num", "i4"),
     ("lifetime", "f4"), ("x", "f4", (9,)), ("centroid", "f4", (6,)), ("keypoints", "f4", (NKP,)),
     ("fade_x", "f4"), ("fade_z", "f4"), ("fade_size", "f4")],
    align=True,
)

_lib = None


class MmwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmmw_hip error {code}: {msg}")
        self.code = code


# The exceptions the reference's track() can raise on this path, as subclasses of BOTH MmwError and the reference's type, so
# that `except ValueError:` / `except np.linalg.LinAlgError:` around a drop-in TrackBuffer.track() keeps working:
#   sklearn's input validation in apply_DBscan (Utils.py:272-278)      -> ValueError
#   np.linalg.inv / det on a singular 6x6 (Tracking.py:556-560, filterpy) -> numpy.linalg.LinAlgError
#   (N_est - 1) * N == 0 in _get_Rc (Tracking.py:310-312)              -> ZeroDivisionError
class MmwNonFinite(MmwError, ValueError):
    pass


class MmwSingular(MmwError, np.linalg.LinAlgError):
    pass


class MmwDivZero(MmwError, ZeroDivisionError):
    pass


def error_for(code, msg) -> MmwError:
    cls = {E_NONFINITE: MmwNonFinite, E_SINGULAR: MmwSingular, E_DIVZERO: MmwDivZero}.get(code, MmwError)
    return cls(code, msg)


def source_hash() -> str:
    """First 16 hex digits of the SHA-256 over csrc/*.hip, csrc/*.hpp (byte order of their names) and include/mmw.h:
    what csrc/Makefile compiles into mmw_version().  The sources travel with the package, so a library built from other
    sources -- a stale .so with a newer mtime, one copied in from another checkout -- is recognisable."""
    import hashlib
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))
    h = hashlib.sha256()
    for p in [os.path.join(CSRC, f) for f in names] + [os.path.join(os.path.dirname(_HERE), "include", "mmw.h")]:
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def built_hash(path: str = None):
    """The src:<hash> a built library carries, read from the file (no dlopen); None if absent or unhashed."""
    import re
    path = path or LIB_PATH
    if not os.path.isfile(path):
        return None
    with open(path, "rb") as fh:
        m = re.search(rb"\(gfx950\) src:([0-9a-f]{16})", fh.read())
    return m.group(1).decode() if m else None


def build(force: bool = False) -> str:
    """Compile the HIP library in-tree (hipcc --offload-arch=gfx950) unless the one present was built from exactly
    these sources (mmw_version()'s hash, not file times)."""
    if force or built_hash() != source_hash():
        subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
        if built_hash() != source_hash():
            raise RuntimeError(f"{LIB_PATH}: built, but its source hash {built_hash()} is not {source_hash()}")
    return LIB_PATH


def _preload_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's, different file).
    Two HIP runtimes in one process cannot both own the GPU ("No HIP GPUs are available" in
    whichever comes second), so when torch is installed we bind to ITS runtime up front --
    without importing torch -- and torch later finds the same object already mapped."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        p = os.path.join(libdir, name)
        if os.path.isfile(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """Load libmmw_hip.so and declare every prototype.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_hip_runtime()
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950).  mmwave_msc_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.mmw_version.restype = C.c_char_p
    ver = (L.mmw_version() or b"").decode()
    if os.path.isdir(CSRC) and not ver.endswith("src:" + source_hash()):
        raise ImportError(
            f"{LIB_PATH} reports {ver!r} but the sources beside it hash to {source_hash()}: it was built from other "
            "sources.  Rebuild it: `python -c 'import __graft_entry__ as g; g.build()'`.")
    vp, vpp = C.c_void_p, C.POINTER(C.c_void_p)
    i32, i32p, i64p = C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    f64p = C.POINTER(C.c_double)
    cfgp = C.POINTER(MmwConfig)
    sig = {
        "mmw_config_default": (C.c_int, [cfgp]),
        "mmw_create": (C.c_int, [cfgp, i32, i32, i32, vpp]),
        "mmw_destroy": (C.c_int, [vp]),
        "mmw_last_error": (C.c_char_p, [vp]),
        "mmw_reset": (C.c_int, [vp]),
        "mmw_set_stream": (C.c_int, [vp, vp]),
        "mmw_pop_frame": (C.c_int, [vp, vp]),
        "mmw_synchronize": (C.c_int, [vp]),
        "mmw_stream_wait": (C.c_int, [vp, vp]),
        "mmw_wait_stream": (C.c_int, [vp, vp]),
        "mmw_get_dims": (C.c_int, [vp, i32p, i32p, i32p, i32p, i32p]),
        "mmw_dev_alloc": (C.c_int, [vp, C.c_size_t, vpp]),
        "mmw_dev_free": (C.c_int, [vp, vp]),
        "mmw_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "mmw_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "mmw_normalize": (C.c_int, [vp, vp, vp, vp, vp]),
        "mmw_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp]),
        "mmw_step_host": (C.c_int, [vp, vp, vp, vp, vp, vp, vp]),
        "mmw_step_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp]),
        "mmw_frame_host": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "mmw_frame_posture_host": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "mmw_attach_posture": (C.c_int, [vp, vp]),
        "mmw_normalize_f32": (C.c_int, [vp, vp, vp, vp, vp]),
        "mmw_dbscan": (C.c_int, [vp, vp, vp, i32, C.c_double, i32, vp, vp]),
        "mmw_features": (C.c_int, [vp, vp, vp, i32, i32p]),
        "mmw_set_keypoints": (C.c_int, [vp, vp, vp, i32]),
        "mmw_features_async": (C.c_int, [vp, vp, vp, vp, i32, i32]),
        "mmw_features_wait": (C.c_int, [vp, i32, i32p]),
        "mmw_set_keypoints_uid": (C.c_int, [vp, vp, vp, vp, i32]),
        "mmw_get_inner": (C.c_int, [vp, vp, vp, vp, i32]),
        "mmw_set_batch_size": (C.c_int, [vp, vp, i32]),
        "mmw_set_batch_frame": (C.c_int, [vp, i32, vp, i32]),
        "mmw_check": (C.c_int, [vp]),
        "mmw_get_num_tracks": (C.c_int, [vp, vp]),
        "mmw_get_tracks": (C.c_int, [vp, vp, i32]),
        "mmw_get_batch_ring": (C.c_int, [vp, vp, vp]),
        "mmw_get_track_ring_frame": (C.c_int, [vp, i32, i32, i32, vp, i32p]),
        "mmw_get_batch_ring_frame": (C.c_int, [vp, i32, i32, vp, i32p]),
        "mmw_track_table": (C.c_int, [vp, vp, i32, i32]),
        "mmw_profile_enable": (C.c_int, [vp, i32]),
        "mmw_profile_reset": (C.c_int, [vp]),
        "mmw_profile_get": (C.c_int, [vp, i32, f64p, i64p]),
        "mmw_kernel_name": (C.c_char_p, [i32]),
        "mmw_version": (C.c_char_p, []),
        "mmw_stats_get": (C.c_int, [vp, vp]),
        "mmw_stats_reset": (C.c_int, [vp]),
        "mmw_diag_queue": (C.c_int, [vp, vp]),
        "mmw_side_workers": (C.c_int, [vp]),
        "mmw_step_kind": (C.c_int, [vp]),
        "mmw_kalman_layout": (C.c_int, [vp]),
        "mmw_streams_concurrent": (C.c_int, [vp, vp, vp]),
        "mmw_reset_scenes": (C.c_int, [vp, vp]),
        "mmw_get_errors": (C.c_int, [vp, vp]),
        "mmw_clear_errors": (C.c_int, [vp, vp, i32]),
        "mmw_set_chain_side_stream": (C.c_int, [vp, i32]),
        "mmw_stats_get_ext": (C.c_int, [vp, vp]),
        "mmw_mars_conv3d": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32]),
        "mmw_mars_conv_split": (C.c_int, [vp, i32, vp, vp, vp, vp, vp, vp, C.c_int64, i32, vp, vp]),
        "mmw_mars_range_fixup": (C.c_int, [vp, vp, vp, i32, vp, vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp]),
        "mmw_mars_dense1_split": (C.c_int, [vp, vp, C.c_int64, vp, C.c_int64, vp, vp, i32, i32, i32]),
        "mmw_mars_head_small": (C.c_int, [vp, vp, C.c_int64, vp, C.c_int64, vp, vp, vp, vp, vp, i32, i32, i32]),
        "mmw_parse_uart": (C.c_int, [vp, C.c_size_t, vp, vp, vp, i32, vp, vp, vp, vp]),
        "mmw_find_tlv": (C.c_int, [vp, C.c_size_t, vp, vp, vp, vp, vp]),
        "mmw_normalize_tlv": (C.c_int, [vp, vp, C.c_size_t, vp, vp, vp, vp]),
        "mmw_format_frames": (C.c_int, [vp, vp, vp, vp, vp, i32]),
    }
    assert sorted(sig) == sorted(EXPORTS)
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def default_config(**overrides) -> MmwConfig:
    cfg = MmwConfig()
    load().mmw_config_default(C.byref(cfg))
    apply_overrides(cfg, overrides)
    return cfg


def apply_overrides(cfg: MmwConfig, overrides: dict):
    for k, v in overrides.items():
        if k == "kf_spread_lim":
            for i in range(6):
                cfg.kf_spread_lim[i] = float(v[i])
        elif k == "default_posture":
            for i in range(NKP):
                cfg.default_posture[i] = float(v[i])
        elif k == "s_tilt":
            ang = np.radians(v)  # Utils.py:315
            cfg.tilt_cos, cfg.tilt_sin = float(np.cos(ang)), float(np.sin(ang))
        else:
            if not hasattr(cfg, k):
                raise AttributeError(f"mmw_config has no field {k!r}")
            setattr(cfg, k, v)
    return cfg
