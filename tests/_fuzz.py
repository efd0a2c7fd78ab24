"""Random configurations for the differential tests (tests/test_gpu_fuzz.py against oracle/c on the GPU box,
tests/test_reference_fuzz.py against the live reference in the build container).  Everything derives from the seed.

The keys of a case's "cfg" are mmw_config / orc_config field names (include/mmw.h); `reference_overrides` maps them to
the reference's constants.py names."""
import os

import numpy as np

# the suite runs seeds 0..63; MMW_FUZZ_SEED0 / MMW_FUZZ_CASES move and widen the window for a one-off run
# (profiles/NOTEBOOK.md round 4: seeds 64..575 after the k_post / k_track changes)
SEED0 = int(os.environ.get("MMW_FUZZ_SEED0", "0"))
N_CASES = int(os.environ.get("MMW_FUZZ_CASES", "64"))

_CONST = {  # mmw_config field -> constants.py name
    "fb_frames_batch": "FB_FRAMES_BATCH", "db_min_samples": "DB_MIN_SAMPLES_MIN", "tr_max_tracks": "TR_MAX_TRACKS",
    "kf_enable_est": "KF_ENABLE_EST", "db_z_weight": "DB_Z_WEIGHT", "db_range_weight": "DB_RANGE_WEIGHT", "db_eps": "DB_EPS",
    "tr_lifetime_dynamic": "TR_LIFETIME_DYNAMIC", "tr_lifetime_static": "TR_LIFETIME_STATIC", "tr_vel_thres": "TR_VEL_THRES",
    "tr_gate": "TR_GATE", "kf_q_std": "KF_Q_STD", "kf_p_init": "KF_P_INIT", "kf_group_disp_est_init": "KF_GROUP_DISP_EST_INIT",
    "kf_a_n": "KF_A_N", "kf_est_pointnum": "KF_EST_POINTNUM", "kf_a_spr": "KF_A_SPR", "kf_spread_lim": "KF_SPREAD_LIM",
    "model_min_input": "MODEL_MIN_INPUT", "fb_frames_batch_static": "FB_FRAMES_BATCH_STATIC", "db_points_thres": "DB_POINTS_THRES",
    "db_spread_thres": "DB_SPREAD_THRES", "db_inner_eps": "DB_INNER_EPS",
}


def reference_overrides(cfg_kw: dict) -> dict:
    out = {}
    for k, v in cfg_kw.items():
        if k == "dim_x":
            out["MOTION_MODEL"] = "CONST_VEL_MODEL" if v == 6 else "CONST_ACC_MODEL"
        elif k == "seek_inner":
            out["SEEK_INNER"] = bool(v)
        elif k == "kf_enable_est":
            out["KF_ENABLE_EST"] = bool(v)
        elif k in _CONST:
            out[_CONST[k]] = v
        elif k in ("track_cap",):
            pass
        else:
            raise KeyError(k)
    return out


ARMS = ("wide", "small", "threshold")
# apply_DBscan's input sizes at which scikit-learn changes what it does: 11 | 12 brute force -> BallTree
# (NearestNeighbors: n_neighbors 5 >= n // 2), 60 | 61, 120 | 121, 240 | 241 one more tree level (leaf_size 30)
SK_THRESHOLDS = (11, 12, 60, 61, 120, 121, 240, 241)


def draw_case(seed: int, max_pts: int = 1024, max_scenes: int = 8, frames: int = 12, arm: str = "wide") -> dict:
    """arm "wide": the whole configuration surface (the draws of a seed never change: recorded seeds stay what they were).
    arm "small": N <= 16 points per frame, DB_MIN_SAMPLES_MIN 1 .. 4, ring 1 .. 2 -- clouds sklearn answers by brute force, or
    right behind that switch, with cores in them.  arm "threshold": the cloud sizes of SK_THRESHOLDS (+- 1), reached by one
    frame (ring 1) or by the sum of ring frames, with a min_samples small enough for clusters."""
    assert arm in ARMS, arm
    rng = np.random.default_rng(770000 + int(seed))
    seek_inner = seed % 8 == 5
    # points per frame: small, medium and large clouds; exact multiples of 64 only by accident
    lim = 200 if seek_inner else max_pts
    n = int({0: rng.integers(1, 48), 1: rng.integers(48, 200), 2: rng.integers(200, min(520, lim) + 1),
             3: rng.integers(1, lim + 1)}[int(rng.integers(0, 4))])
    n = max(1, min(n, lim))
    ring = int(rng.integers(2, 4)) if seek_inner else int(rng.integers(1, 5))
    tr_max = int(rng.integers(1, 13))
    if arm != "wide":
        arng = np.random.default_rng(771000 + int(seed))
        if arm == "small":
            n, ring = int(arng.integers(1, 17)), (int(arng.integers(2, 4)) if seek_inner else int(arng.integers(1, 3)))
        else:
            ring = int(arng.integers(2, 4)) if seek_inner else int(arng.integers(1, 4))
            u = int(arng.choice([t for t in SK_THRESHOLDS if t <= min(lim, max_pts) * ring])) + int(arng.integers(-1, 2))
            n = max(1, min(lim, -(-u // ring) if arng.random() < 0.5 else u))
    # every cluster apply_DBscan finds becomes a track (Tracking.py:576-589): keep (TR_MAX_TRACKS - 1) + U / min_samples below
    # the 63 tracks a scene's list may hold in every layout
    ms_floor = -(-ring * n // (62 - tr_max)) + 1
    min_samples = int(max(ms_floor, rng.integers(3, 41) if n >= 120 else rng.integers(2, max(4, n // 3 + 1))))   # (small clouds can cluster too)
    if arm == "small":
        min_samples = int(max(-(-ring * n // (62 - tr_max)), arng.integers(1, 5)))      # (min_samples 1: every point a core point)
    elif arm == "threshold":
        min_samples = int(max(ms_floor, arng.integers(2, 7) if n <= 16 else arng.integers(3, 13)))
    cfg = dict(
        fb_frames_batch=ring - 1, db_min_samples=min_samples, tr_max_tracks=tr_max,
        kf_enable_est=int(rng.integers(0, 2)), dim_x=int(rng.choice([9, 6])),
        db_eps=float(np.round(rng.uniform(0.1, 0.6), 3)), db_z_weight=float(np.round(rng.uniform(0.0, 1.0), 2)),
        db_range_weight=float(np.round(rng.uniform(0.0, 0.06), 3)),
        tr_gate=float(np.round(rng.uniform(2.5, 7.0), 2)), tr_vel_thres=float(np.round(rng.uniform(0.05, 0.4), 2)),
        tr_lifetime_dynamic=float(np.round(rng.uniform(0.2, 1.0), 2)), tr_lifetime_static=float(np.round(rng.uniform(0.4, 1.5), 2)),
        kf_a_n=float(np.round(rng.uniform(0.5, 0.99), 2)), kf_est_pointnum=float(rng.integers(5, 40)),
        kf_a_spr=float(np.round(rng.uniform(0.5, 0.99), 2)), kf_p_init=float(np.round(rng.uniform(0.05, 0.5), 2)),
        kf_group_disp_est_init=float(np.round(rng.uniform(0.05, 0.5), 2)), kf_q_std=float(np.round(rng.uniform(0.3, 2.0), 2)),
        model_min_input=int(rng.choice([0, 0, 30, 100])),
    )
    if seek_inner:
        cfg.update(seek_inner=1, fb_frames_batch_static=int(rng.integers(1, ring + 1)), db_points_thres=int(rng.integers(20, 60)),
                   db_spread_thres=float(np.round(rng.uniform(0.4, 1.0), 2)), db_inner_eps=float(np.round(rng.uniform(0.05, 0.2), 3)))
    return dict(seed=int(seed), S=int(rng.integers(1, max_scenes + 1)), N=n, F=int(frames), cfg=cfg, seek_inner=seek_inner, arm=arm)


def scene_inputs(case: dict):
    """points[F,S,N,8] float32 (fp32-representable: exact on every path), counts[F,S] int32 -- 0 = the frame is skipped,
    -1 = track() on an empty cloud (MMW_EMPTY_FRAME) --, dt[F,S] float64."""
    from mmwave_msc_amd.synth import make_pair_scene, make_scene
    rng = np.random.default_rng(880000 + case["seed"])
    S, N, F = case["S"], case["N"], case["F"]
    pts = np.zeros((F, S, N, 8), np.float32)
    cnt = np.zeros((F, S), np.int32)
    dts = np.zeros((F, S))
    for s in range(S):
        k = int(rng.integers(0, 13))
        k = min(k, max(N // 4, 0))                          # (a target needs a few points to be one)
        dt_seq = np.round(rng.uniform(0.05, 0.2, size=F), 3)
        presence = None
        if k > 0 and rng.random() < 0.5:                    # targets that appear late or vanish for good
            presence = np.ones((F, k), dtype=bool)
            for j in range(k):
                if rng.random() < 0.4:
                    a = int(rng.integers(1, F))
                    if rng.random() < 0.5:
                        presence[:a, j] = False
                    else:
                        presence[a:, j] = False
        if case["seek_inner"] and N >= 64 and s % 2 == 0:
            p, c, d = make_pair_scene(int(rng.integers(1 << 30)), F, N, 1 + s % 2, sep=0.6 + 0.1 * (s % 4), static=(s % 3 == 0))
            d = dt_seq
        else:
            p, c, d = make_scene(int(rng.integers(1 << 30)), F, N, k, ragged=bool(rng.random() < 0.5), presence=presence, dt_seq=dt_seq)
        c = c.copy()
        for f in range(F):
            u = rng.random()
            if u < 0.08:
                c[f] = 0
            elif u < 0.16:
                c[f] = -1
        pts[:, s], cnt[:, s], dts[:, s] = p, c, d
    return pts, cnt, dts


def plant_nonfinite(case: dict, pts: np.ndarray, cnt: np.ndarray, rate: float = 0.25):
    """The non-finite arm: NaN / +inf / -inf written into random columns (all 8: x, y, z, vx, vy, vz, doppler, peakVal) of random
    rows of about `rate` of the (frame, scene) pairs -- rows that will be unassigned (start-up frames, clutter) and rows a track
    would have taken.  The reference gates on columns 0..5 (a non-finite innovation never passes `d2 < TR_GATE`, Tracking.py:559-563),
    pushes such rows into the global ring, and sklearn's input validation raises ValueError out of apply_DBscan (Utils.py:272-278)
    on every frame the row is in the ring and the trigger holds (Tracking.py:693-697).  Returns the planted (frame, scene, row,
    column) list; `pts` is modified in place."""
    rng = np.random.default_rng(990000 + case["seed"])
    F, S = cnt.shape
    planted = []
    for f in range(F):
        for s in range(S):
            c = int(cnt[f, s])
            if c <= 0 or rng.random() >= rate:
                continue
            for _ in range(int(rng.integers(1, 4))):
                r, col = int(rng.integers(0, c)), int(rng.integers(0, 8))
                pts[f, s, r, col] = (np.nan, np.inf, -np.inf)[int(rng.integers(0, 3))]
                planted.append((f, s, r, col))
    if not planted:   # (a short case whose every draw missed -- three of 1536 seeds of a wide window: one plant, so that the arm tests something)
        nz = np.argwhere(cnt > 0)
        if len(nz):
            f, s = (int(v) for v in nz[int(rng.integers(0, len(nz)))])
            r, col = int(rng.integers(0, int(cnt[f, s]))), int(rng.integers(0, 8))
            pts[f, s, r, col] = (np.nan, np.inf, -np.inf)[int(rng.integers(0, 3))]
            planted.append((f, s, r, col))
    return planted


def nonfinite_raw_rows(seed: int, n: int = 160) -> np.ndarray:
    """Raw radar rows (x, y, z, doppler, peakVal) most of which pass normalize_data's scene filter, with NaN / +inf / -inf planted
    in random columns of every third row (some rows twice), the r == 0 row, and x = 0 with an infinite doppler."""
    rng = np.random.default_rng(4400 + seed)
    raw = np.zeros((n, 5))
    raw[:, 0] = rng.uniform(-3, 3, n)
    raw[:, 1] = rng.uniform(0.5, 7, n)
    raw[:, 2] = rng.uniform(-1.6, 0.5, n)
    raw[:, 3] = rng.normal(0, 0.6, n)
    raw[:, 4] = rng.integers(0, 400, n)
    raw[0, :3] = 0.0
    raw = raw.astype(np.float32).astype(np.float64)
    for i in range(4, n, 3):
        raw[i, int(rng.integers(0, 5))] = (np.nan, np.inf, -np.inf)[int(rng.integers(0, 3))]
        if rng.random() < 0.3:
            raw[i, int(rng.integers(0, 5))] = (np.nan, np.inf, -np.inf)[int(rng.integers(0, 3))]
    raw[1, 0], raw[1, 3] = 0.0, np.inf
    return raw
