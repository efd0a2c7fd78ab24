"""Pin the CPU oracle (oracle/c) against the golden vectors recorded from the
reference (oracle/gen_golden.py).  CPU-only."""
import numpy as np
import pytest

from oracle import c_oracle as co
from tests._golden import (GOLDEN, assert_feat_equal, assert_tracks_match, load_scenario, overrides_to_cfg_kwargs,
                           scenario_names, scenario_tol)

import os


@pytest.mark.parametrize("name", scenario_names())
def test_tracker_matches_reference_golden(name):
    g = load_scenario(name)
    kw = overrides_to_cfg_kwargs(g["overrides"])
    n = g["pts"].shape[1]
    cfg = co.default_config(**kw)
    sc = co.OracleScene(cfg, n)
    ring = cfg.fb_frames_batch + 1
    if "batch_init" in g:   # BatchedData(init_data)
        sc.set_batch_frame(g["batch_init"])
    resize = {int(a): int(b) for a, b in g["overrides"].get("BATCH_RESIZE", [])}
    for f in range(g["pts"].shape[0]):
        c = int(g["cnt"][f])
        if f in resize:     # BatchedData.change_buffer_size before this frame's track()
            sc.set_batch_size(resize[f])
        if c == 0 and not g["overrides"].get("TRACK_EMPTY"):  # offline_main.py:56: empty frames never reach track()
            continue
        raised = int(g["raised"][f]) if "raised" in g else 0
        if raised:   # the reference raised ValueError out of apply_DBscan (a NaN / inf row in the ring, Utils.py:272-278)
            with pytest.raises(co.OracleNonFinite) as ei:
                sc.track(g["pts"][f, :c].astype(np.float64), float(g["dt"][f]))
            assert ei.value.kind == ("NaN", "infinity")[raised - 1], f"{name} f{f}: which ValueError"
            assert sc.last_db_n == co.DB_RAISED
            assoc, labels = sc.last_assoc, None
        else:
            assoc, labels = sc.track(g["pts"][f, :c].astype(np.float64), float(g["dt"][f]))
        assert np.array_equal(assoc, g["assoc"][f, :c]), f"{name} f{f}: association differs"
        dbn = int(g["db_n"][f])
        if dbn < 0:
            assert labels is None, f"{name} f{f}: DBSCAN ran but reference did not"
        else:
            assert labels is not None and len(labels) == dbn
            assert np.array_equal(labels, g["labels"][f, :dbn]), f"{name} f{f}: DBSCAN labels differ"
        nt = int(g["n_tracks"][f])
        assert sc.n_tracks == nt
        assert_tracks_match(sc.tracks(), g["tracks"][f, :nt], ctx=f"{name} f{f}", tol=scenario_tol(g))
        if kw.get("seek_inner"):   # ClusterTrack.seek_inner_clusters (Tracking.py:409-448) with its call site active
            calls = sc.inner_calls()
            assert len(calls) == int(g["inner_calls"][f]), f"{name} f{f}: seek_inner_clusters call count"
            for q, (tpos, lab_in) in enumerate(calls):
                assert tpos == g["inner_track"][f, q] and len(lab_in) == g["inner_n"][f, q]
                assert np.array_equal(lab_in, g["inner_labels"][f, q, : len(lab_in)]), f"{name} f{f}: inner labels of call {q}"
            assert [sc.track_ring_size(t) for t in range(nt)] == list(g["ring_size"][f, :nt]), f"{name} f{f}: batch.size"
        br = sc.batch_ring()
        assert len(br) == g["ring_len"][f] and np.array_equal(br, g["ring_n"][f, : len(br)])
        feat, owner = sc.features()
        nf = int(g["n_feat"][f])
        assert len(owner) == nf and np.array_equal(owner, g["owner"][f, :nf])
        if nf:
            assert_feat_equal(feat, g["feat"][f, :nf], ctx=f"{name} f{f}")
    assert ring >= 1


def test_normalize_matches_reference_golden():
    z = np.load(os.path.join(GOLDEN, "normalize.npz"))
    cfg = co.default_config(s_height=float(z["s_height"]), s_tilt=float(z["s_tilt"]))
    out = co.normalize(cfg, z["raw"])
    assert out.shape == z["out"].shape
    assert np.allclose(out, z["out"], rtol=0, atol=1e-12)
    # pass-through columns are exact
    assert np.array_equal(out[:, [0, 6, 7]], z["out"][:, [0, 6, 7]])


def test_dbscan_matches_sklearn_golden():
    z = np.load(os.path.join(GOLDEN, "dbscan.npz"))
    cfg = co.default_config()
    for n in z["sizes"]:
        pts = z[f"pts_{n}"].astype(np.float64)
        for ms in (35, 8):
            lab = co.dbscan(cfg, pts, min_samples=ms)
            assert np.array_equal(lab, z[f"labels_{n}_{ms}"]), f"n={n} min_samples={ms}"
        lab = co.dbscan(cfg, pts)
        sizes = np.array([(lab == k).sum() for k in range(lab.max() + 1)], dtype=np.int32)
        assert np.array_equal(sizes, z[f"sizes_{n}"])


def test_dbscan_of_at_most_13_points_matches_sklearn_golden():
    """Clouds of 1 .. 13 points, DB_MIN_SAMPLES_MIN 1 .. 10: sklearn answers n // 2 <= 5 by BRUTE FORCE (NearestNeighbors'
    default n_neighbors = 5 >= n_samples // 2, sklearn/neighbors/_base.py:622-633), the BallTree starts at 12 points.  The
    file would fail an implementation that ran the tree's take-all / prune shortcuts on the small clouds (what oracle/c and the
    kernels did up to round 5): `tree_differs` -- recorded from DBSCAN(algorithm="ball_tree") -- says how many clouds per
    (size, min_samples) that rule labels differently, and the test insists the file has teeth."""
    import json
    z = np.load(os.path.join(GOLDEN, "dbscan_small.npz"))
    cfg = co.default_config(db_eps=float(z["db_eps"]), db_z_weight=float(z["db_z_weight"]), db_range_weight=float(z["db_range_weight"]))
    sizes, mss = [int(v) for v in z["sizes"]], [int(v) for v in z["min_samples"]]
    assert sizes == list(range(1, 14))
    for n in sizes:
        pts = z[f"pts_{n}"].astype(np.float64)
        for ms in mss:
            want = z[f"labels_{n}_{ms}"]
            for c in range(len(pts)):
                assert np.array_equal(co.dbscan(cfg, pts[c], min_samples=ms), want[c]), f"n={n} min_samples={ms} cloud {c}"
    td = z["tree_differs"]
    per = len(z["pts_3"])
    for n in range(2, 12):     # every brute-force size has a min_samples at which >= 10 % of its clouds tell the two rules apart
        assert td[sizes.index(n)].max() >= per // 10, (n, td[sizes.index(n)])
    assert td[sizes.index(12):].sum() == 0   # (12 points up: the tree IS the reference)
    kw = json.loads(str(z["named_cfg"]))
    ms = kw.pop("db_min_samples")
    got = co.dbscan(co.default_config(**kw), z["named_pts"].astype(np.float64), min_samples=ms)
    assert np.array_equal(got, z["named_labels"]) and list(got) == [-1, -1, -1]   # the round-5 review's 3-point cloud: no track


def test_dbscan_of_more_than_1920_points_matches_sklearn_golden():
    """64 .. 128 BallTree leaves (Utils.py:250-291 has no size limit): the oracle the GPU's global-memory path is checked
    against, pinned on sklearn's labels for the reference's own metric."""
    z = np.load(os.path.join(GOLDEN, "dbscan_huge.npz"))
    cfg = co.default_config()
    for n in z["sizes"]:
        pts = z[f"pts_{n}"].astype(np.float64)
        for ms in (35, 8):
            assert np.array_equal(co.dbscan(cfg, pts, min_samples=ms), z[f"labels_{n}_{ms}"]), f"n={n} min_samples={ms}"


def test_log_and_pairwise_sum_helpers():
    import math

    rng = np.random.default_rng(5)
    L = co.lib()
    xs = np.concatenate([rng.uniform(1e-300, 1e300, 2000), rng.uniform(0.5, 2.0, 4000), 10.0 ** rng.uniform(-20, 20, 4000)])
    for x in xs:
        got, want = L.orc_log(float(x)), math.log(float(x))
        assert abs(got - want) <= 2.3e-16 * max(1.0, abs(want)), (x, got, want)
    for n in [0, 1, 5, 7, 8, 9, 57, 64, 127, 128, 129, 200, 256, 257, 460, 1000, 1536]:
        a = rng.normal(size=max(n, 1))[:n]
        got = L.orc_np_pairwise_sum(np.ascontiguousarray(a).ctypes.data_as(co.C.POINTER(co.C.c_double)), n)
        want = float(np.add.reduce(a)) if n else 0.0
        assert got == want, (n, got, want)


def test_pop_frame_matches_reference_recording():
    """BatchedData.pop_frame() (Tracking.py:66-71) in the C restatement: ring sizes, association, track count."""
    from oracle import c_oracle as co
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "popframe.npz"))
    pops = set(int(p) for p in g["pops"])
    sc = co.OracleScene(co.default_config(), 64)
    for f in range(len(g["cnt"])):
        c = int(g["cnt"][f])
        if f in pops:
            sc.pop_frame()
            assert sc.batch_ring().tolist() == [int(v) for v in g["after_pop"][f] if v >= 0], f
        a, _ = sc.track(g["pts"][f, :c].astype(np.float64), float(g["dt"][f]))
        assert np.array_equal(a.astype(np.int16), g["assoc"][f, :c]), f
        assert sc.n_tracks == int(g["n_tracks"][f]), f
        assert sc.batch_ring().tolist() == [int(v) for v in g["ring_n"][f, : int(g["ring_len"][f])]], f
