"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

TEST INFRASTRUCTURE ONLY; container-only (needs /root/reference, which never
travels).  Usage:  python -m oracle.gen_golden  [--only NAME]

What is recorded: for seeded synthetic scenes the inputs
(fp32-representable points, counts, dt) and, per frame, the reference's
association vector, DBSCAN labels, track list (every float field of every
track), ring lengths and `format_single_frame` feature tensors; plus
`normalize_data` in/out pairs, `apply_DBscan` labels at sizes that cross the
BallTree node-count thresholds, and the `OfflineManager` frame sequence on a
synthetic 2-shard CSV (captures the 40th-frame truncation quirk).

Environment of the recorded run is stored in every file (`meta`): numpy /
scikit-learn / scipy versions (the reference pins numpy 1.26.3, sklearn 1.3.2,
filterpy 1.4.5; here: numpy 2.2.x, sklearn 1.7.x, filterpy = own shim).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from mmwave_msc_amd.synth import make_pair_scene, make_scene  # noqa: E402
from oracle.c_oracle import TRACK_DTYPE  # noqa: E402
from oracle.ref_import import have_reference, load_reference  # noqa: E402
from oracle.ref_runner import RefScene  # noqa: E402

GOLDEN_DIR = os.path.join(_ROOT, "tests", "golden")

# name -> dict(seed, N, K, F, overrides, ragged, presence/dt builders, feat_every)
SCENARIOS = {
    "n60_k1": dict(seed=101, N=60, K=1, F=24, over={}),
    "n200_k2": dict(seed=102, N=200, K=2, F=40, over={}),
    "n256_k4": dict(seed=103, N=256, K=4, F=24, over={}),
    "n256_k3_ragged": dict(seed=104, N=256, K=3, F=24, over={}, ragged=True),
    "n512_k8": dict(seed=105, N=512, K=8, F=12, over={"TR_MAX_TRACKS": 8}),
    "n512_k5": dict(seed=106, N=512, K=5, F=12, over={"TR_MAX_TRACKS": 8}),
    "expiry": dict(seed=107, N=128, K=2, F=60, over={}, presence="gap"),
    "var_dt": dict(seed=108, N=160, K=2, F=40, over={}, presence="flicker", dt="var"),
    "const_vel": dict(seed=109, N=128, K=2, F=24, over={"MOTION_MODEL": "CONST_VEL_MODEL"}),
    "dense_k1": dict(seed=110, N=512, K=1, F=10, over={}),
    "clutter_only": dict(seed=111, N=400, K=0, F=5, over={}),
    "fb0": dict(seed=112, N=128, K=2, F=16, over={"FB_FRAMES_BATCH": 0}),
    "empty_frames": dict(seed=113, N=96, K=1, F=16, over={}, zero_frames=(3, 9)),
    # track() CALLED on empty point clouds (offline_main.py:56 never does; a drop-in caller may): long enough runs of them
    # for the dynamic track to expire (TR_LIFETIME_DYNAMIC = 3 s) while the ring empties
    "empty_tracked": dict(seed=120, N=128, K=2, F=60, over={"TRACK_EMPTY": True}, zero_frames=tuple(range(8, 12)) + tuple(range(20, 54))),
    "kf_est": dict(seed=114, N=128, K=2, F=20, over={"KF_ENABLE_EST": True}, presence="flicker"),
    "max_size": dict(seed=115, N=640, K=0, F=4, over={}),
    # BatchedData(init_data) (Tracking.py:38-41) and BatchedData.change_buffer_size (Tracking.py:60-64) on the GLOBAL ring:
    # the ring starts with 40 clutter rows, shrinks to size 2 before frame 6 and to 1 before frame 14, back to 3 at 20
    "batch_init_resize": dict(seed=121, N=160, K=2, F=26, over={"BATCH_INIT": 40, "BATCH_RESIZE": [[6, 2], [14, 1], [20, 3]]}, presence="flicker"),
    # ClusterTrack.seek_inner_clusters with its call site (Tracking.py:656) active: pairs of people one outer cluster wide
    "inner_pair": dict(seed=116, N=256, K=1, F=14, over={"SEEK_INNER": True}, pairs=dict(sep=0.8)),
    "inner_static": dict(seed=117, N=256, K=1, F=12, over={"SEEK_INNER": True, "FB_FRAMES_BATCH_STATIC": 3}, pairs=dict(sep=0.9, static=True)),
    "inner_two_pairs": dict(seed=118, N=512, K=2, F=10, over={"SEEK_INNER": True, "TR_MAX_TRACKS": 4}, pairs=dict(sep=0.8)),
    "inner_close": dict(seed=119, N=200, K=1, F=10, over={"SEEK_INNER": True, "DB_POINTS_THRES": 60}, pairs=dict(sep=0.45)),
    # NaN / +-inf rows: the gate never takes a point whose columns 0..5 are not finite (Tracking.py:559-563), such rows enter the
    # global ring, and sklearn's input validation raises ValueError out of apply_DBscan (Utils.py:272-278) on every frame the
    # trigger holds (Tracking.py:693-697) while one is in the ring -- `raised` records it per frame (1 = "contains NaN",
    # 2 = "contains infinity"), with the state the exception leaves behind.  TR_MAX_TRACKS = 2 with two targets: while both tracks
    # live (from frame 12 on, when the second target appears) apply_DBscan is not called and a non-finite row sits in the ring
    # without raising; before, it raises on every frame it is in the ring.
    # ["all", 6]: every row of the frame gets a NaN doppler -- the assigned ones carry it into their track's ring and feature map.
    "nonfinite": dict(seed=122, N=160, K=2, F=26, over={"TR_MAX_TRACKS": 2}, presence="late",
                      nonfinite=[[1, 3, 0, "nan"], [6, 7, 2, "inf"], [7, 2, 3, "nan"], [13, 5, 1, "-inf"], [14, "all", 6, "nan"],
                                 [20, 1, 7, "inf"]]),
}


def _meta():
    import sklearn, scipy  # noqa: E401
    return json.dumps({
        "numpy": np.__version__, "sklearn": sklearn.__version__, "scipy": scipy.__version__,
        "filterpy": "own shim (oracle/filterpy_shim), filterpy 1.4.5 absent from image",
        "reference": "AsteriosPar/mmWave_MSc @ /root/reference/src (Tracking.py, Utils.py, constants.py)",
    })


def _presence(kind, f, k):
    if kind is None:
        return None
    p = np.ones((f, k), dtype=bool)
    if kind == "gap":  # both targets vanish for > 3 s, then come back
        p[10:48, :] = False
    elif kind == "flicker":
        p[8:12, 0] = False
        p[20:30, 1] = False
    elif kind == "late":
        p[:12, 1] = False
    return p


def gen_scenario(name, sc):
    f, n, k = sc["F"], sc["N"], sc["K"]
    rng = np.random.default_rng(sc["seed"] + 7000)
    dt_seq = None
    if sc.get("dt") == "var":
        dt_seq = np.round(rng.uniform(0.05, 0.3, size=f), 3)
    if "pairs" in sc:
        pts, cnt, dt = make_pair_scene(sc["seed"], f, n, k, **sc["pairs"])
    else:
        pts, cnt, dt = make_scene(sc["seed"], f, n, k, ragged=sc.get("ragged", False),
                                  presence=_presence(sc.get("presence"), f, k), dt_seq=dt_seq)
    for zf in sc.get("zero_frames", ()):
        cnt[zf] = 0
        pts[zf] = 0
    for fr, row, col, val in sc.get("nonfinite", ()):
        v = {"nan": np.nan, "inf": np.inf, "-inf": -np.inf}[val]
        if row == "all":
            pts[fr, : cnt[fr], col] = v
        else:
            assert row < cnt[fr]
            pts[fr, row, col] = v
    over = dict(sc["over"])
    if "nonfinite" in sc:
        over["NONFINITE"] = sc["nonfinite"]
    const, _, _ = load_reference()
    if "MOTION_MODEL" in over:
        over["MOTION_MODEL"] = getattr(const, over["MOTION_MODEL"])
    init_rows = None
    if "BATCH_INIT" in over:   # the rows BatchedData(init_data) starts with: clutter of another seed
        init_rows = make_scene(sc["seed"] + 500, 1, int(over["BATCH_INIT"]), 0)[0][0].astype(np.float64)
    resize = {int(a): int(b) for a, b in over.get("BATCH_RESIZE", [])}
    ref = RefScene({k2: v for k2, v in over.items() if k2 not in ("TRACK_EMPTY", "BATCH_INIT", "BATCH_RESIZE", "NONFINITE")}, init_data=init_rows)
    ring = ref.const.FB_FRAMES_BATCH + 1
    tmax = 0
    rec = dict(assoc=np.full((f, n), -2, np.int16), db_n=np.full(f, -1, np.int32),
               labels=np.full((f, ring * n), -2, np.int16), n_tracks=np.zeros(f, np.int32),
               ring_len=np.zeros(f, np.int32), ring_n=np.zeros((f, 4), np.int32))
    if "NONFINITE" in over:
        rec["raised"] = np.zeros(f, np.int8)
    inner = over.get("SEEK_INNER", False)
    if inner:   # per frame: the seek_inner_clusters calls (pre-maintenance track position, labels) and every track's batch.size
        rec.update(inner_calls=np.zeros(f, np.int32), inner_track=np.full((f, 8), -1, np.int32), inner_n=np.zeros((f, 8), np.int32),
                   inner_labels=np.full((f, 8, ring * n), -2, np.int16), ring_size=np.zeros((f, 16), np.int32))
    tracks, feats, owners = [], [], []
    for i in range(f):
        if cnt[i] == 0 and not over.get("TRACK_EMPTY"):
            # offline_main.py:56 skips track()/estimate_posture() for empty frames
            rec["n_tracks"][i] = ref.n_tracks
            rec["ring_len"][i] = len(ref.batch_ring())
            rec["ring_n"][i, : rec["ring_len"][i]] = ref.batch_ring()
            tracks.append(ref.tracks())
            feats.append(None)
            owners.append(None)
            continue
        if i in resize:
            ref.batch.change_buffer_size(resize[i])
        try:
            with np.errstate(invalid="ignore"):
                a, lab = ref.track(pts[i, : cnt[i]].astype(np.float64), float(dt[i]))
        except ValueError as e:
            if "NONFINITE" not in over or not str(e).startswith("Input X contains"):
                raise
            from oracle.ref_runner import _Recorder
            rec["raised"][i] = 1 if str(e).startswith("Input X contains NaN") else 2
            a, lab = _Recorder.assoc.copy(), None     # (_calc_dist_fun had returned before apply_DBscan raised)
        rec["assoc"][i, : cnt[i]] = a
        if lab is not None:
            rec["db_n"][i] = len(lab)
            rec["labels"][i, : len(lab)] = lab
        rec["n_tracks"][i] = ref.n_tracks
        if inner:
            rec["inner_calls"][i] = len(ref.inner)
            for q, (tpos, lab_in) in enumerate(ref.inner):
                rec["inner_track"][i, q] = tpos
                rec["inner_n"][i, q] = len(lab_in)
                rec["inner_labels"][i, q, : len(lab_in)] = lab_in
            rs = ref.track_ring_sizes()
            rec["ring_size"][i, : len(rs)] = rs
        br = ref.batch_ring()
        rec["ring_len"][i] = len(br)
        rec["ring_n"][i, : len(br)] = br
        tracks.append(ref.tracks())
        fe, ow = ref.features()
        feats.append(fe)
        owners.append(ow)
        tmax = max(tmax, ref.n_tracks)
    ref.close()
    tmax = max(tmax, 1)
    trk = np.zeros((f, tmax), dtype=TRACK_DTYPE)
    fshape = (3, 8, 8, 5) if ring > 1 else (8, 8, 5)
    if ring > 1:
        fshape = (ring, 8, 8, 5)
    feat = np.zeros((f, tmax) + fshape, np.float32)
    n_feat = np.full(f, -1, np.int32)
    owner = np.full((f, tmax), -1, np.int32)
    for i in range(f):
        trk[i, : len(tracks[i])] = tracks[i]
        if feats[i] is not None:
            n_feat[i] = len(owners[i])
            if len(owners[i]):
                feat[i, : len(owners[i])] = feats[i]
                owner[i, : len(owners[i])] = owners[i]
    over_json = {k2: (v.__name__ if hasattr(v, "__name__") else v) for k2, v in over.items()}
    if init_rows is not None:
        rec["batch_init"] = init_rows
    np.savez_compressed(
        os.path.join(GOLDEN_DIR, f"track_{name}.npz"), pts=pts, cnt=cnt, dt=dt, tracks=trk,
        feat=feat, n_feat=n_feat, owner=owner, overrides=json.dumps(over_json), meta=_meta(), **rec)
    print(f"  {name}: F={f} N={n} K={k} tracks(max)={tmax} dbscan_frames={(rec['db_n'] >= 0).sum()}"
          + (f" raised={rec['raised'].tolist()}" if "raised" in rec else ""))


def gen_normalize():
    const, utils, _ = load_reference()
    rng = np.random.default_rng(2024)
    n = 300
    raw = np.zeros((n, 5))
    raw[:, 0] = rng.uniform(-4, 4, n)
    raw[:, 1] = rng.uniform(-1, 9, n)
    raw[:, 2] = rng.uniform(-3, 2, n)
    raw[:, 3] = rng.normal(0, 0.6, n)
    raw[:, 4] = rng.integers(0, 400, n)
    raw[0, :3] = 0.0          # r == 0 branch (Utils.py:387-390)
    raw[1, 2] = (2.5 - const.S_HEIGHT)  # near the z <= 2.5 edge
    raw = raw.astype(np.float32).astype(np.float64)
    det = {"x": list(raw[:, 0]), "y": list(raw[:, 1]), "z": list(raw[:, 2]),
           "doppler": list(raw[:, 3]), "peakVal": list(raw[:, 4])}
    out = utils.normalize_data(det)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "normalize.npz"), raw=raw, out=out,
                        s_height=const.S_HEIGHT, s_tilt=const.S_TILT, meta=_meta())
    print(f"  normalize: {n} -> {out.shape[0]} rows")


def gen_dbscan():
    const, utils, _ = load_reference()
    from sklearn.cluster import DBSCAN
    sizes = [1, 2, 30, 60, 61, 120, 121, 240, 241, 480, 481, 700, 961, 1536]
    data = {}
    for n in sizes:
        k = max(1, min(8, n // 60))
        f = 3 if n >= 90 else 1
        per = -(-n // f)
        pts, _, _ = make_scene(5000 + n, f, per, k)
        x = pts.reshape(-1, 8)[:n].astype(np.float64)
        for ms in (35, 8):
            lab = DBSCAN(eps=const.DB_EPS, min_samples=ms, metric=utils.altered_EuclideanDist).fit_predict(x)
            data[f"labels_{n}_{ms}"] = lab.astype(np.int16)
        data[f"pts_{n}"] = x.astype(np.float32)
        # apply_DBscan cluster lists (row order kept): store sizes only, rows follow from labels
        cl = utils.apply_DBscan(x)
        data[f"sizes_{n}"] = np.array([len(c) for c in cl], dtype=np.int32)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "dbscan.npz"), sizes=np.array(sizes), meta=_meta(), **data)
    print(f"  dbscan: sizes {sizes}")


def gen_dbscan_huge():
    """apply_DBscan on clouds of more than 1920 points (64 .. 128 BallTree leaves): the reference has no size limit
    (Utils.py:250-291); here they run on slabs in global memory (k_dbscan_huge).  A separate file so that dbscan.npz
    stays byte-stable."""
    const, utils, _ = load_reference()
    from sklearn.cluster import DBSCAN
    sizes = [1921, 2500, 3072, 3841, 4096]
    data = {}
    for n in sizes:
        f = 4
        per = -(-n // f)
        pts, _, _ = make_scene(7000 + n, f, per, 8)
        x = pts.reshape(-1, 8)[:n].astype(np.float64)
        for ms in (35, 8):
            lab = DBSCAN(eps=const.DB_EPS, min_samples=ms, metric=utils.altered_EuclideanDist).fit_predict(x)
            data[f"labels_{n}_{ms}"] = lab.astype(np.int16)
        data[f"pts_{n}"] = x.astype(np.float32)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "dbscan_huge.npz"), sizes=np.array(sizes), meta=_meta(), **data)
    print(f"  dbscan_huge: sizes {sizes}, clusters {[int(data[f'labels_{n}_35'].max()) + 1 for n in sizes]} / {[int(data[f'labels_{n}_8'].max()) + 1 for n in sizes]}")


SMALL_SIZES = list(range(1, 14))
SMALL_MIN_SAMPLES = list(range(1, 7)) + [8, 10]
SMALL_CLOUDS = 40


def small_cloud(seed, n):
    """One cloud of `n` points around a random centre whose pairwise metric straddles DB_EPS: a third of the clouds tight
    (most pairs inside eps), a third at the threshold, a third loose; every fifth cloud repeats one of its points (duplicate
    rows: metric 0); every other cloud of three or more points is a SHELL (below).  All 8 columns filled, fp32-representable."""
    rng = np.random.default_rng(31000 + seed)
    sig = (0.18, 0.30, 0.45)[seed % 3]
    x = np.zeros((n, 8))
    cx, cy = rng.uniform(-2, 2), rng.uniform(1.5, 6)
    if seed % 2 == 1 and n >= 3:
        # a shell around the centre, squared metric radius 0.3 .. 0.5 eps: "metric(p, centroid) + radius <= eps" holds for most
        # points (the tree would hand each of them the whole cloud) while points on opposite sides are up to 4 radii apart
        u = rng.standard_normal((n, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        r = np.sqrt(rng.uniform(0.3, 0.5, (n, 1)) * 0.3 / (1 - 0.03 * cy))
        x[:, 0], x[:, 1], x[:, 2] = cx + r[:, 0] * u[:, 0], cy + r[:, 0] * u[:, 1], 0.9 + r[:, 0] * u[:, 2] / np.sqrt(0.4)
    else:
        x[:, 0] = cx + sig * rng.standard_normal(n)
        x[:, 1] = cy + sig * rng.standard_normal(n)
        x[:, 2] = rng.uniform(0.05, 1.8, n)
    x[:, 3:6] = 0.3 * rng.standard_normal((n, 3))
    x[:, 6] = 0.3 * rng.standard_normal(n)
    x[:, 7] = rng.gamma(1.0, 30.0, n)
    if n > 1 and seed % 5 == 4:
        x[int(rng.integers(1, n))] = x[0]
    return x.astype(np.float32)


def gen_dbscan_small():
    """apply_DBscan on clouds of 1 .. 13 points with DB_MIN_SAMPLES_MIN 1 .. 6, 8, 10.  DBSCAN.fit builds its NearestNeighbors with the
    default n_neighbors = 5 (sklearn/cluster/_dbscan.py:410-418), and NearestNeighbors._fit answers `n_neighbors >= n_samples // 2`
    -- 1 .. 11 points -- by BRUTE FORCE (neighbors/_base.py:622-633); the BallTree and its prune / take-all shortcuts start at
    12 points.  `tree_differs` records, per (size, min_samples), how many of the clouds the BallTree rule would have labelled
    differently (DBSCAN(algorithm="ball_tree") on the same cloud): the file FAILS an implementation that runs the tree's
    shortcuts below 12 points.  Plus one named case: the 3-point cloud of the round-5 review (fuzz case 12002 of
    tests/_fuzz.py, scene 1, frame 3; its own eps / weights / min_samples)."""
    const, utils, _ = load_reference()
    from sklearn.cluster import DBSCAN
    data = {}
    differs = np.zeros((len(SMALL_SIZES), len(SMALL_MIN_SAMPLES)), np.int32)
    for a, n in enumerate(SMALL_SIZES):
        pts = np.stack([small_cloud(100 * n + c, n) for c in range(SMALL_CLOUDS)])
        data[f"pts_{n}"] = pts
        for b, ms in enumerate(SMALL_MIN_SAMPLES):
            labs = np.zeros((SMALL_CLOUDS, n), np.int8)
            for c in range(SMALL_CLOUDS):
                x = pts[c].astype(np.float64)
                labs[c] = DBSCAN(eps=const.DB_EPS, min_samples=ms, metric=utils.altered_EuclideanDist).fit_predict(x)
                cl = utils.apply_DBscan(x, min_samples=ms)
                assert [len(k) for k in cl] == [int((labs[c] == k).sum()) for k in range(labs[c].max() + 1)]
                tree = DBSCAN(eps=const.DB_EPS, min_samples=ms, metric=utils.altered_EuclideanDist, algorithm="ball_tree").fit_predict(x)
                differs[a, b] += int(not np.array_equal(tree, labs[c]))
            data[f"labels_{n}_{ms}"] = labs
    # the named case
    from tests._fuzz import draw_case, plant_nonfinite, scene_inputs
    case = draw_case(12002, max_pts=260, max_scenes=2, frames=12)
    fp, fc, _ = scene_inputs(case)
    plant_nonfinite(case, fp, fc, rate=0.3)
    named = fp[3, 1, [0, 1, 3]].astype(np.float32)
    kw = case["cfg"]
    saved = (const.DB_Z_WEIGHT, const.DB_RANGE_WEIGHT)
    const.DB_Z_WEIGHT, const.DB_RANGE_WEIGHT = kw["db_z_weight"], kw["db_range_weight"]
    try:
        lab = DBSCAN(eps=kw["db_eps"], min_samples=kw["db_min_samples"], metric=utils.altered_EuclideanDist).fit_predict(named.astype(np.float64))
        tree = DBSCAN(eps=kw["db_eps"], min_samples=kw["db_min_samples"], metric=utils.altered_EuclideanDist,
                      algorithm="ball_tree").fit_predict(named.astype(np.float64))
    finally:
        const.DB_Z_WEIGHT, const.DB_RANGE_WEIGHT = saved
    assert list(lab) == [-1, -1, -1] and list(tree) == [0, 0, 0], (lab, tree)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "dbscan_small.npz"), sizes=np.array(SMALL_SIZES), min_samples=np.array(SMALL_MIN_SAMPLES),
                        tree_differs=differs, named_pts=named, named_labels=lab.astype(np.int8),
                        named_cfg=json.dumps({k: kw[k] for k in ("db_eps", "db_z_weight", "db_range_weight", "db_min_samples")}),
                        db_eps=const.DB_EPS, db_z_weight=const.DB_Z_WEIGHT, db_range_weight=const.DB_RANGE_WEIGHT, meta=_meta(), **data)
    print("  dbscan_small: clouds the BallTree rule would label differently, of", SMALL_CLOUDS, "per (size, min_samples):")
    for a, n in enumerate(SMALL_SIZES):
        print(f"    n={n:2d}: {list(map(int, differs[a]))}")


def gen_offline():
    """OfflineManager + offline_main dt logic (Utils.py:53-177, offline_main.py:40-62).
    offline_main.py itself cannot be imported (PyQt5/keras); its loop is 12 lines and
    is replayed here against the REAL OfflineManager/normalize_data/TrackBuffer."""
    const, utils, tracking = load_reference()
    rng = np.random.default_rng(77)
    n_frames, per_shard = 130, const.FB_EXPERIMENT_FILE_SIZE
    pts, cnt, _ = make_scene(909, n_frames, 48, 1)
    rows = []
    t_ms = 1_700_000_000_000
    for fr in range(1, n_frames + 1):
        t_ms += int(rng.integers(90, 115))
        m = int(rng.integers(20, 48))
        for r in range(m):
            p = pts[fr - 1, r]
            # raw sensor frame: undo the height so normalize_data keeps most rows
            rows.append((fr, float(p[0]), float(p[1]), float(p[2]) - 1.5, float(p[6]), int(p[7]), t_ms))
    shards = {1: [r for r in rows if r[0] <= 100], 2: [r for r in rows if r[0] > 100]}
    csv_text = {}
    with tempfile.TemporaryDirectory() as d:
        for k, rr in shards.items():
            txt = "".join(f"{a},{b!r},{c!r},{e!r},{g!r},{h},{i}\n" for (a, b, c, e, g, h, i) in rr)
            csv_text[k] = txt
            with open(os.path.join(d, f"{k}.csv"), "w") as fh:
                fh.write(txt)
        man = utils.OfflineManager(d)
        seq = []  # (ok, frame_count, n_points, posix0)
        tb, batch = tracking.TrackBuffer(), tracking.BatchedData()
        first = True
        ntr, dts, nnorm = [], [], []
        while not man.is_finished():
            ok, fc, det = man.get_data()
            if ok:
                seq.append((1, fc, len(det["x"]), det["posix"][0]))
                if first:
                    tb.dt = 0.1
                    first = False
                else:
                    tb.dt = det["posix"][0] / 1000 - tb.t
                tb.t = det["posix"][0] / 1000
                eff = utils.normalize_data(det)
                if eff.shape[0] != 0:
                    tb.track(eff, batch)
                ntr.append(len(tb.effective_tracks)); dts.append(tb.dt); nnorm.append(eff.shape[0])
            else:
                seq.append((0, fc, 0, 0))
    np.savez_compressed(os.path.join(GOLDEN_DIR, "offline.npz"), csv1=csv_text[1], csv2=csv_text[2],
                        seq=np.array(seq, dtype=np.int64), n_tracks=np.array(ntr, np.int32),
                        dt=np.array(dts), n_norm=np.array(nnorm, np.int32), meta=_meta())
    print(f"  offline: {len(seq)} get_data() calls, {sum(s[0] for s in seq)} frames delivered, "
          f"1-point frames at {[s[1] for s in seq if s[0] and s[2] == 1]}")


def gen_formatters():
    """Dataset-side formatters (Utils.py:523-574): format_batched_frames, format_single_frame_mode."""
    const, utils, _ = load_reference()
    rng = np.random.default_rng(77)
    out = {}
    for case, sizes in enumerate([(70, 64, 12), (5, 40), (0, 64, 3), (100,)]):
        frames = []
        for n in sizes:
            f = np.zeros((n, 8))
            f[:, 0:3] = rng.normal(0, 1, size=(n, 3))
            f[:, 3:6] = rng.normal(0, 0.3, size=(n, 3))
            f[:, 6] = rng.normal(0, 0.3, size=n)
            f[:, 7] = rng.gamma(1.0, 30.0, size=n)
            frames.append(f.astype(np.float32).astype(np.float64))
        blk = utils.format_batched_frames([fr.copy() for fr in frames])
        out[f"c{case}_n"] = np.array(sizes, dtype=np.int32)
        out[f"c{case}_in"] = np.concatenate(frames) if sum(sizes) else np.zeros((0, 8))
        out[f"c{case}_block"] = blk
        for bs in (1, 2, 3):
            out[f"c{case}_mode{bs}"] = utils.format_single_frame_mode(blk.copy(), 93.0, 40.0, bs, fuse=False)
            out[f"c{case}_fuse{bs}"] = utils.format_single_frame_mode(blk.copy(), 93.0, 40.0, bs, fuse=True)
    # calc_projection_points (Utils.py:180-219), incl. the x_dist == 0 / z_dist == 0 branches
    pp = rng.uniform(-3, 6, size=(40, 3))
    pp[0, 0] = const.M_X
    pp[1, 2] = const.M_Z
    pp[2] = [const.M_X, 2.0, const.M_Z]
    out["proj_in"] = pp
    out["proj_out"] = np.array([utils.calc_projection_points(*row) for row in pp])
    # calc_fade_square (Visualizer.py:14-29).  Visualizer.py needs Qt to import; the function is compiled on its
    # own and run against the real constants / calc_projection_points on tracks shaped like the reference's
    # (state.x a (9, 1) column, keypoints a float32 57-vector)
    import types
    fade = _reference_functions(os.path.join(os.path.dirname(const.__file__), "Visualizer.py"), {"calc_fade_square"},
                                {"const": const, "calc_projection_points": utils.calc_projection_points, "ClusterTrack": object})
    fx = rng.uniform(-2, 6, size=(24, 9))
    fx[0, 1] = 40.0   # far: the size clamps at the minimum
    fx[1, 1] = -3.0   # behind the screen plane: clamps at the maximum
    fk = rng.normal(0, 0.5, size=(24, 57)).astype(np.float32)
    res = []
    for t in range(24):
        tr = types.SimpleNamespace(state=types.SimpleNamespace(x=fx[t].reshape(9, 1)), keypoints=fk[t])
        (cx, cz), sz = fade["calc_fade_square"](tr)
        res.append([float(np.asarray(cx).reshape(-1)[0]), float(np.asarray(cz).reshape(-1)[0]), float(np.asarray(sz).reshape(-1)[0])])
    out["fade_x"], out["fade_kp"], out["fade_out"] = fx, fk, np.array(res)
    # the same with the keypoints widened first: no float32 scalar takes part in the arithmetic, so the values do
    # not depend on numpy's promotion rules (2.x keeps `float32 - python float` in float32, 1.26 -- the reference's
    # pinned version -- does not)
    res = []
    for t in range(24):
        tr = types.SimpleNamespace(state=types.SimpleNamespace(x=fx[t].reshape(9, 1)), keypoints=fk[t].astype(np.float64))
        (cx, cz), sz = fade["calc_fade_square"](tr)
        res.append([float(np.asarray(cx).reshape(-1)[0]), float(np.asarray(cz).reshape(-1)[0]), float(np.asarray(sz).reshape(-1)[0])])
    out["fade_out64"] = np.array(res)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "formatters.npz"), meta=_meta(), **out)
    print("  formatters: 4 cases + projection + fade squares")


def _uart_packet(frame, objs, qfmt=9, tlv_type=1, num_det=None):
    """One TI mmWave demo UART packet (header 36 B + one TLV) carrying `objs` = rows of int16
    (rangeIdx, dopplerIdx, peakVal, x, y, z)."""
    import struct
    body = struct.pack("<HH", len(objs), qfmt) + b"".join(struct.pack("<6h", *[int(v) for v in o]) for o in objs)
    tlv = struct.pack("<II", tlv_type, len(body)) + body
    total = 36 + len(tlv)
    hdr = bytes([2, 1, 4, 3, 6, 5, 8, 7]) + struct.pack("<IIIIIII", 0x01020304, total, 0xA1443, frame, 123456,
                                                        len(objs) if num_det is None else num_det, 1)
    return hdr + tlv


def gen_uart():
    """ReadIWR14xx.read (ReadDataIWR1443.py:27-201) on synthetic byte streams: the module is imported with the
    pyserial stand-in, __init__ (which opens ports) is bypassed, a fake Dataport delivers the chunks."""
    import importlib
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "serial_shim"))
    load_reference()
    mod = importlib.import_module("ReadDataIWR1443")

    class Port:
        def __init__(self):
            self.q = b""
        @property
        def in_waiting(self):
            return len(self.q)
        def read(self, n):
            d, self.q = self.q[:n], self.q[n:]
            return d
        def write(self, *_):
            pass
        def close(self):
            pass

    cfgp = {"rangeIdxToMeters": 0.0436, "dopplerResolutionMps": 0.1252, "numDopplerBins": 16.0}
    rd = object.__new__(mod.ReadIWR14xx)
    rd.MMWDEMO_UART_MSG_DETECTED_POINTS = 1
    rd.maxBufferSize = 2 ** 15
    rd.magicWord = [2, 1, 4, 3, 6, 5, 8, 7]
    rd.byteBuffer = np.zeros(2 ** 15, dtype="uint8")
    rd.byteBufferLength = 0
    rd.configParameters = cfgp
    rd.Dataport = Port()
    rd.CLIport = Port()
    rng = np.random.default_rng(11)

    def objs(n):
        o = np.zeros((n, 6), dtype=np.int64)
        o[:, 0] = rng.integers(0, 200, n)
        # (non-negative int16 only: under numpy >= 2 the reference raises OverflowError when it stores a u16 above
        #  32767 into its int16 arrays; with its pinned numpy 1.26 such values wrap, which is what the C parser does)
        o[:, 1] = rng.integers(0, 16, n)         # doppler bins on both sides of numDopplerBins/2 - 1
        o[:, 2] = rng.integers(0, 3000, n)
        o[:, 3:6] = rng.integers(0, 2500, size=(n, 3))
        return o

    # Only packets whose first TLV is NOT decoded (no objects, or another TLV type): under numpy >= 2 the
    # reference's decode branch raises OverflowError at `dopplerIdx[...] - 65535` (ReadDataIWR1443.py:150-157)
    # for every detected-points packet, so that branch cannot be recorded here.  What IS recorded: the byte
    # buffer discipline (garbage before the magic word, packets split over reads, several packets per read --
    # the LAST magic word wins --, the "remove processed data" rule) and the header fields.
    p1, p2, p3 = _uart_packet(7, objs(0)), _uart_packet(8, objs(3), tlv_type=2), _uart_packet(9, objs(0))
    p4, p5 = _uart_packet(10, objs(6), tlv_type=6), _uart_packet(11, objs(0), num_det=0)
    chunks = [b"\x00\x11\x02\x01garbage" + p1, p2, p3[:30], p3[30:], p4 + p5, b"\x05" * 10, _uart_packet(13, objs(2), tlv_type=3),
              b"\x02\x01\x04", p1[:20], p1[20:] + p2[:10], p2[10:]]
    out = {"n_chunks": np.int32(len(chunks)), "cfg": np.array([cfgp["rangeIdxToMeters"], cfgp["dopplerResolutionMps"], cfgp["numDopplerBins"]])}
    for i, ch in enumerate(chunks):
        rd.Dataport.q = ch
        ok, fn, det = rd.read()
        out[f"chunk{i}"] = np.frombuffer(ch, dtype=np.uint8)
        out[f"ok{i}"] = np.int32(ok)
        out[f"frame{i}"] = np.int64(fn)
        out[f"buflen{i}"] = np.int64(rd.byteBufferLength)
        if ok:
            out[f"det{i}"] = np.stack([np.asarray(det[k], dtype=np.float64) for k in ("x", "y", "z", "doppler", "peakVal", "range")], axis=1)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "uart.npz"), meta=_meta(), **out)
    print(f"  uart: {len(chunks)} chunks, ok = {[int(out[f'ok{i}']) for i in range(len(chunks))]}")


def gen_popframe():
    """BatchedData.pop_frame() (Tracking.py:66-71; the dataset pre-processing calls it between log shards) in the
    middle of a tracked sequence: the global ring after every frame, the association and the track count."""
    f, n, k = 14, 60, 2
    pts, cnt, dt = make_scene(4711, f, n, k)
    pops = (2, 3, 6, 7, 11)   # before these frames are tracked (frames 6+7: twice in a row; an empty ring is a no-op)
    ref = RefScene({})
    ring_n = np.zeros((f, 4), np.int32)
    ring_len = np.zeros(f, np.int32)
    after_pop = np.full((f, 4), -1, np.int32)
    assoc = np.full((f, n), -2, np.int16)
    n_tracks = np.zeros(f, np.int32)
    rings = []
    for i in range(f):
        if i in pops:
            ref.batch.pop_frame()
            br = ref.batch_ring()
            after_pop[i, : len(br)] = br
        a, _ = ref.track(pts[i, : cnt[i]].astype(np.float64), float(dt[i]))
        assoc[i, : cnt[i]] = a
        br = ref.batch_ring()
        ring_len[i] = len(br)
        ring_n[i, : len(br)] = br
        n_tracks[i] = ref.n_tracks
        eff = ref.batch.effective_data
        rings.append(np.asarray(eff, dtype=np.float64).reshape(-1, 8) if len(eff) else np.zeros((0, 8)))
    ref.close()
    rows = np.zeros((f, 3 * n, 8))
    for i, r in enumerate(rings):
        rows[i, : len(r)] = r
    np.savez_compressed(os.path.join(GOLDEN_DIR, "popframe.npz"), pts=pts, cnt=cnt, dt=dt, pops=np.array(pops), assoc=assoc,
                        n_tracks=n_tracks, ring_len=ring_len, ring_n=ring_n, after_pop=after_pop, ring_rows=rows, meta=_meta())
    print(f"  popframe: ring lengths {ring_len.tolist()}, tracks {n_tracks.tolist()}")


def _reference_functions(path, names, namespace):
    """Compile the named top-level functions / constants of a reference script that cannot be imported (it runs
    its pipeline at import time) and return them bound to `namespace`.  Nothing of the source is kept."""
    import ast
    with open(path, "r") as fh:
        tree = ast.parse(fh.read(), filename=path)
    keep = [n for n in tree.body
            if (isinstance(n, ast.FunctionDef) and n.name in names)
            or (isinstance(n, ast.Assign) and all(isinstance(t, ast.Name) and t.id in names for t in n.targets))]
    mod = ast.Module(body=keep, type_ignores=[])
    exec(compile(mod, path, "exec"), namespace)
    return namespace


def gen_preprocess():
    """The dataset pre-processing (src/preprocessing.py:27-145,148-275,298-384) on a synthetic experiment: the
    mmWave log of the offline golden plus a synthetic Kinect log.  preprocessing.py runs its whole pipeline when it is
    imported (and wants wakepy + the recorded data), so its functions are compiled one by one from the file and run
    against the REAL constants / Utils / Tracking with the paths pointed at a temporary tree."""
    import csv as _csv
    import shutil as _shutil
    import pandas as _pd
    const, utils, tracking = load_reference()
    z = np.load(os.path.join(GOLDEN_DIR, "offline.npz"))
    rng = np.random.default_rng(4242)
    saved = {k: getattr(const, k) for k in ("P_LOG_PATH", "P_PREPROCESS_PATH", "P_FORMATTED_PATH")}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        try:
            const.P_LOG_PATH, const.P_PREPROCESS_PATH, const.P_FORMATTED_PATH = d + "/log", d + "/pre", d + "/fmt"
            mm = f"{const.P_LOG_PATH}{const.P_MMWAVE_DIR}A1"
            os.makedirs(mm)
            os.makedirs(f"{const.P_LOG_PATH}{const.P_KINECT_DIR}")
            os.makedirs(f"{const.P_PREPROCESS_PATH}{const.P_KINECT_DIR}")
            os.makedirs(f"{const.P_PREPROCESS_PATH}{const.P_MMWAVE_DIR}")
            os.makedirs(d + "/centroids_final")
            for k in (1, 2):
                with open(os.path.join(mm, f"{k}.csv"), "w") as fh:
                    fh.write(str(z[f"csv{k}"]))
            # Kinect log: time stamp, frame number, 19 joints x (x, z, y), one trailing field.  Most mmWave frames get
            # a row within +-15 ms, every seventh none closer than 30 ms; a few rows match nothing.
            stamps = {}
            for k in (1, 2):
                for line in str(z[f"csv{k}"]).splitlines():
                    f = line.split(",")
                    stamps.setdefault(int(f[0]), int(f[6]))
            rows = []
            kf = 1000
            for fr, t in sorted(stamps.items()):
                off = int(rng.integers(-15, 16)) if fr % 7 else 30 + int(rng.integers(0, 10))
                base = np.array([0.1, 2.5, 0.9]) + rng.normal(0, 0.02, 3)
                joints = base + rng.normal(0, 0.25, (19, 3))
                rows.append([t + off, kf] + [repr(float(v)) for v in joints.reshape(-1)] + [0])
                kf += 1
                if fr % 11 == 0:
                    rows.append([t + 47, kf] + [repr(float(v)) for v in (joints + 0.01).reshape(-1)] + [0]); kf += 1
            kin_txt = "".join(",".join(str(v) for v in r) + "\n" for r in rows)
            with open(f"{const.P_LOG_PATH}{const.P_KINECT_DIR}A1.csv", "w") as fh:
                fh.write(kin_txt)
            names = {"pair", "filter_kinect_frames", "translate_kinect", "relative_kinect", "static_kinect", "preprocess_dataset",
                     "extract_parts", "format_mmwave_to_npy", "format_kinect_to_npy", "KINECT_Z", "KINECT_X", "RELATIVE_ENABLED"}
            ns = {"np": np, "pd": _pd, "os": os, "csv": _csv, "shutil": _shutil, "const": const, "tqdm": (lambda it: it),
                  "normalize_data": utils.normalize_data, "OfflineManager": utils.OfflineManager,
                  "format_single_frame_mode": utils.format_single_frame_mode, "relative_coordinates": utils.relative_coordinates,
                  "format_batched_frames": utils.format_batched_frames, "TrackBuffer": tracking.TrackBuffer,
                  "BatchedData": tracking.BatchedData, "print": (lambda *a, **k: None)}
            ref = _reference_functions(os.path.join(os.path.dirname(const.__file__), "preprocessing.py"), names, ns)
            os.chdir(d)
            pairs = ref["pair"]("A1")
            ref["preprocess_dataset"]()
            pre_dir = f"{const.P_PREPROCESS_PATH}{const.P_MMWAVE_DIR}/A1"
            pre_files = sorted(os.listdir(pre_dir), key=lambda x: int(os.path.splitext(x)[0]))
            pre_txt = [open(os.path.join(pre_dir, f)).read() for f in pre_files]
            kin_out = open(f"{const.P_PREPROCESS_PATH}{const.P_KINECT_DIR}A1.csv", newline="").read()
            cen = np.load(d + "/centroids_final/A1_centroid.npy")
            # split_sets would move the experiment under a mode directory; do that by hand, then format
            os.makedirs(f"{const.P_PREPROCESS_PATH}{const.P_MMWAVE_DIR}training")
            _shutil.copytree(pre_dir, f"{const.P_PREPROCESS_PATH}{const.P_MMWAVE_DIR}training/A1")
            os.makedirs(f"{const.P_PREPROCESS_PATH}{const.P_KINECT_DIR}training")
            _shutil.copy(f"{const.P_PREPROCESS_PATH}{const.P_KINECT_DIR}A1.csv", f"{const.P_PREPROCESS_PATH}{const.P_KINECT_DIR}training/A1.csv")
            os.makedirs(f"{const.P_FORMATTED_PATH}{const.P_MMWAVE_DIR}0")
            os.makedirs(f"{const.P_FORMATTED_PATH}{const.P_KINECT_DIR}0")
            ref["format_mmwave_to_npy"]("training", 0)
            ref["format_kinect_to_npy"]("training", 0)
            fm = np.load(f"{const.P_FORMATTED_PATH}{const.P_MMWAVE_DIR}0/training_mmWave.npy")
            fk = np.load(f"{const.P_FORMATTED_PATH}{const.P_KINECT_DIR}0/training_labels.npy")
            # row transforms on their own
            probe = next(_csv.reader(kin_txt.splitlines()))
            tr = ref["translate_kinect"](list(probe))
            st = ref["static_kinect"](list(tr))
            rel = ref["relative_kinect"](list(tr), [0.25, 2.0])
        finally:
            os.chdir(cwd)
            for k, v in saved.items():
                setattr(const, k, v)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "preprocess.npz"), kinect_in=kin_txt, pairs=np.array(pairs, dtype=np.int64),
                        pre_files=np.array(pre_files), pre_txt=np.array(pre_txt), kinect_out=kin_out, centroids=cen,
                        fmt_mmwave=fm, fmt_labels=fk, probe=np.array(probe), probe_translated=np.array(tr),
                        probe_static=np.array(st), probe_relative=np.array(rel), meta=_meta())
    print(f"  preprocess: {len(pairs)} pairs, {len(cen)} valid frames, files {pre_files}, mmWave {fm.shape}, labels {fk.shape}")


def _splitsets_tree(root):
    """A small pre-processed tree: experiments A1, A4, B2, B7 (mmWave: directories of two CSV shards; Kinect: one CSV each).
    Deterministic content, a few exact zeros in the coordinate columns (add_noise leaves them alone)."""
    rng = np.random.default_rng(99)
    for name in ("A1", "A4", "B2", "B7"):
        os.makedirs(os.path.join(root, "mmWave", name))
        for k in (1, 2):
            rows = []
            for r in range(6):
                v = rng.normal(0, 1, 3).round(4)
                if (r + k) % 4 == 0:
                    v[r % 3] = 0.0
                rows.append(f"{r},{v[0]},{v[1]},{v[2]},{float(rng.normal(20, 5)):.3f}")
            with open(os.path.join(root, "mmWave", name, f"{k}.csv"), "w") as fh:
                fh.write("\n".join(rows) + "\n")
        os.makedirs(os.path.join(root, "kinect"), exist_ok=True)
        with open(os.path.join(root, "kinect", f"{name}.csv"), "w") as fh:
            fh.write("".join(f"{i},{float(rng.normal()):.5f}\n" for i in range(4)))


def _tree_dump(root):
    out = {}
    for base, _, files in os.walk(root):
        for f in files:
            p = os.path.join(base, f)
            out[os.path.relpath(p, root)] = open(p, newline="").read()
    return out


def gen_splitsets():
    """split_sets / add_noise (src/preprocessing.py:406-509) on a synthetic pre-processed tree, numpy's global generator
    seeded: the resulting tree, file by file."""
    import csv as _csv
    import json as _json
    import shutil as _shutil
    const, utils, tracking = load_reference()
    saved = {k: getattr(const, k) for k in ("P_PREPROCESS_PATH", "P_KINECT_DIR", "P_MMWAVE_DIR")}
    prefixes = [["A4"], ["B7"]]
    with tempfile.TemporaryDirectory() as d:
        try:
            _splitsets_tree(d)
            for sub in ("kinect", "mmWave"):   # the reference removes the mode directories first and needs them to exist
                for mode in ("training", "validate", "testing"):
                    os.makedirs(os.path.join(d, sub, mode))
            const.P_PREPROCESS_PATH, const.P_KINECT_DIR, const.P_MMWAVE_DIR = d + "/", "kinect/", "mmWave/"
            ns = {"np": np, "os": os, "csv": _csv, "shutil": _shutil, "const": const}
            ref = _reference_functions(os.path.join(os.path.dirname(const.__file__), "preprocessing.py"), {"split_sets", "add_noise"}, ns)
            ref["split_sets"](prefixes)
            after_split = _tree_dump(d)
            np.random.seed(20241003)
            ref["add_noise"]()
            after_noise = _tree_dump(d)
        finally:
            for k, v in saved.items():
                setattr(const, k, v)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "splitsets.npz"), prefixes=_json.dumps(prefixes), after_split=_json.dumps(after_split, sort_keys=True),
                        after_noise=_json.dumps(after_noise, sort_keys=True), seed=20241003, meta=_meta())
    print(f"  splitsets: {len(after_split)} files after split_sets, {len(after_noise)} after add_noise")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    if not have_reference():
        print("reference not present: nothing generated")
        return 0
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name, sc in SCENARIOS.items():
        if args.only and args.only != name:
            continue
        gen_scenario(name, sc)
    if not args.only or args.only == "normalize":
        gen_normalize()
    if not args.only or args.only == "dbscan":
        gen_dbscan()
    if not args.only or args.only == "dbscan_huge":
        gen_dbscan_huge()
    if not args.only or args.only == "dbscan_small":
        gen_dbscan_small()
    if not args.only or args.only == "offline":
        gen_offline()
    if not args.only or args.only == "formatters":
        gen_formatters()
    if not args.only or args.only == "uart":
        gen_uart()
    if not args.only or args.only == "preprocess":
        gen_preprocess()
    if not args.only or args.only == "splitsets":
        gen_splitsets()
    if not args.only or args.only == "popframe":
        gen_popframe()
    return 0


if __name__ == "__main__":
    sys.exit(main())
