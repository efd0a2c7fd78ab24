"""Pin the reference-faithful Python restatement (oracle/py_tracker.py, the
cpu_baseline "port") against the golden vectors: integer outputs bit-exact.
Short prefixes of the scenarios keep the CPU suite fast (the path is as slow as
the reference by construction)."""
import numpy as np
import pytest

from oracle.py_tracker import Params, PyScene
from tests._golden import close64, load_scenario

CASES = [("n200_k2", 12), ("n256_k3_ragged", 8), ("expiry", 60), ("const_vel", 10), ("var_dt", 16), ("fb0", 8)]


def _params(over):
    kw = {}
    for k, v in over.items():
        if k == "MOTION_MODEL":
            kw["DIM_X"] = 6 if v == "CONST_VEL_MODEL" else 9
        else:
            kw[k] = v
    return Params(**kw)


@pytest.mark.parametrize("name,frames", CASES)
def test_py_restatement_matches_golden(name, frames):
    g = load_scenario(name)
    sc = PyScene(_params(g["overrides"]))
    for f in range(min(frames, g["pts"].shape[0])):
        c = int(g["cnt"][f])
        if c == 0:
            continue
        assoc, labels = sc.track(g["pts"][f, :c].astype(np.float64), float(g["dt"][f]))
        assert np.array_equal(assoc, g["assoc"][f, :c]), f"{name} f{f}"
        dbn = int(g["db_n"][f])
        assert (labels is None) == (dbn < 0)
        if labels is not None:
            assert np.array_equal(labels, g["labels"][f, :dbn])
        nt = int(g["n_tracks"][f])
        assert sc.n_tracks == nt
        for j, t in enumerate(sc.tracks):
            w = g["tracks"][f, j]
            dx = t.x.shape[0]
            assert close64(t.x[:, 0], w["x"][:dx]) and close64(t.P, w["P"][:dx, :dx])
            assert close64(t.group_disp_est, w["group_disp_est"]) and close64(t.spread_est, w["spread_est"])
            assert t.point_num == w["point_num"] and int(t.is_static) == w["is_static"]
