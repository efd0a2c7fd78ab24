"""Pin the oracle's BallTree emulation against the INSTALLED scikit-learn: neighbour sets of
BallTree(leaf_size=30, metric=<callable>).query_radius and DBSCAN labels on fresh random
clouds (not only the committed goldens).  CPU-only; sizes straddle the node-count thresholds."""
import numpy as np
import pytest

sklearn = pytest.importorskip("sklearn")
from sklearn.cluster import DBSCAN  # noqa: E402
from sklearn.neighbors import BallTree  # noqa: E402

from mmwave_msc_amd.synth import make_scene  # noqa: E402
from oracle import c_oracle as co  # noqa: E402

W, ZW, EPS = 0.03, 0.4, 0.3


def metric(p1, p2):
    w = 1 - ((p1[1] + p2[1]) / 2) * W
    return w * ((p1[0] - p2[0]) ** 2 + (p1[1] - p2[1]) ** 2 + ZW * ((p1[2] - p2[2]) ** 2))


@pytest.mark.parametrize("seed,n,k,frames", [(1, 59, 1, 1), (2, 61, 1, 1), (3, 200, 2, 2), (4, 256, 4, 3), (5, 350, 3, 3), (6, 512, 8, 2)])
def test_neighbour_sets_equal_sklearn_balltree(seed, n, k, frames):
    pts, _, _ = make_scene(9000 + seed, frames, -(-n // frames), k)
    x = pts.reshape(-1, 8)[:n].astype(np.float64)
    cfg = co.default_config()
    adj = co.dbscan_neighbors(cfg, x)
    ref = BallTree(x, leaf_size=30, metric=metric).query_radius(x, EPS)
    brute_diff = 0
    for i in range(n):
        mine = np.nonzero(adj[i])[0]
        assert np.array_equal(mine, np.sort(ref[i])), f"query {i}"
        d = np.array([metric(x[i], x[j]) for j in range(n)]) if i < 8 else None
        if d is not None:
            brute_diff += int(not np.array_equal(np.nonzero(d <= EPS)[0], mine))
    for ms in (35, 10):
        lab = DBSCAN(eps=EPS, min_samples=ms, metric=metric).fit_predict(x)
        assert np.array_equal(co.dbscan(cfg, x, min_samples=ms), lab)


def _window_base():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "fuzz_window.json")) as fh:
        return int(os.environ.get("MMW_REF_FUZZ_BASE", json.load(fh)["base"]))


@pytest.mark.parametrize("n", list(range(1, 14)) + [59, 60, 61, 62, 119, 120, 121, 122])
def test_labels_equal_sklearn_around_its_size_switches(n):
    """DBSCAN(metric=<callable>) as apply_DBscan builds it (Utils.py:272-278) against oracle/c on random clouds of the sizes at
    which scikit-learn changes what it does -- 1 .. 11 points: brute force (NearestNeighbors' default n_neighbors = 5 >=
    n_samples // 2, sklearn/neighbors/_base.py:622-633); 12 up: BallTree; 61, 121: one more tree level -- with random eps,
    weights and min_samples, duplicate rows, and clouds tight enough for the tree's take-all rule to fire.  The seeds move
    with the round (tests/fuzz_window.json)."""
    base = _window_base()
    reps = 48 if n <= 13 else 6
    n_core = 0
    for r in range(reps):
        rng = np.random.default_rng([base, n, r])
        w, zw, eps = float(np.round(rng.uniform(0.0, 0.06), 3)), float(np.round(rng.uniform(0.0, 1.0), 2)), float(np.round(rng.uniform(0.1, 0.6), 3))
        sig = float(rng.choice([0.12, 0.2, 0.3, 0.45]))
        x = np.zeros((n, 8))
        x[:, 0] = rng.uniform(-2, 2) + sig * rng.standard_normal(n)
        x[:, 1] = rng.uniform(1.5, 6) + sig * rng.standard_normal(n)
        x[:, 2] = rng.uniform(0.05, 1.8, n) if r % 2 else 0.9 + sig * rng.standard_normal(n)
        x[:, 3:] = rng.standard_normal((n, 5)) * np.array([0.3, 0.3, 0.3, 0.3, 30.0])
        if n > 1 and r % 4 == 3:
            x[int(rng.integers(1, n))] = x[0]
        x = x.astype(np.float32).astype(np.float64)

        def metric(p1, p2, w=w, zw=zw):
            wt = 1 - ((p1[1] + p2[1]) / 2) * w
            return wt * ((p1[0] - p2[0]) ** 2 + (p1[1] - p2[1]) ** 2 + zw * ((p1[2] - p2[2]) ** 2))

        cfg = co.default_config(db_eps=eps, db_z_weight=zw, db_range_weight=w)
        for ms in sorted({1, 2, int(rng.integers(2, 7)), int(rng.integers(2, max(3, n // 2 + 2)))}):
            lab = DBSCAN(eps=eps, min_samples=ms, metric=metric).fit_predict(x)
            got = co.dbscan(cfg, x, min_samples=ms)
            assert np.array_equal(got, lab), (n, r, ms, eps, w, zw)
            n_core += int(lab.max() >= 0)
    assert n_core >= reps     # (clusters do form: the comparison is not all-noise against all-noise)
