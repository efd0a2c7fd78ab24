"""Randomised differential test of the C oracle (oracle/c) against the LIVE reference (/root/reference/src imported in
this container through oracle/ref_import.py; filterpy from oracle/filterpy_shim): what the golden fixtures pin on 22
recorded scenarios, checked again on fresh seeds with random constant overrides -- the configurations of
tests/_fuzz.py (the same generator drives the GPU fuzz test), smaller.

The reference has no tests, so this IS the pin -- and a pin that never moves only ever re-proves what it
proved when it was written.  The window therefore MOVES: tests/fuzz_window.json holds the seed base of the round and the
size of each arm; every round bumps the base past the last window (scripts/bump_fuzz_window.py), and every seed that ever
failed stays in the file's regression list.  Arms (tests/_fuzz.py: draw_case):

  wide        the whole configuration surface (<= 2 scenes, <= 300 points, 10 frames: the reference's DBSCAN metric is a
              Python callable)
  small       N <= 16 points per frame, DB_MIN_SAMPLES_MIN 1 .. 4, ring 1 .. 2: clouds sklearn answers by BRUTE FORCE
              (NearestNeighbors: n_neighbors 5 >= n_samples // 2) or right behind that switch -- the window that found
              the round-5 divergence
  threshold   cloud sizes around 11 | 12, 60 | 61, 120 | 121, 240 | 241 (brute force -> BallTree, one more tree level)
  nonfinite   NaN / +-inf planted in random columns of random rows, all three draws in turn

`-m reference`: skipped where /root/reference is absent (the GPU box).  Integers -- association vectors, DBSCAN labels,
track count and order, point counts, ring lengths, static flags, feature owners -- must be equal; fp64 state within
tests/_golden.py's tolerance (BLAS / LAPACK summation order differs from the oracle's fixed order by a few ULP).

DB_EPS and DB_MIN_SAMPLES_MIN are bound as default arguments of Utils.apply_DBscan when the reference is imported
(Utils.py:250); a run with other values re-binds `apply_DBscan.__defaults__`, which is what editing constants.py does."""
import json
import os

import numpy as np
import pytest

from tests._fuzz import ARMS, draw_case, reference_overrides, scene_inputs
from tests._golden import assert_feat_equal, assert_tracks_match

pytestmark = pytest.mark.reference

with open(os.path.join(os.path.dirname(__file__), "fuzz_window.json")) as _fh:
    WINDOW = json.load(_fh)

_SIZES = {"wide": dict(max_pts=300, max_scenes=2, frames=10), "small": dict(max_pts=300, max_scenes=2, frames=12),
          "threshold": dict(max_pts=300, max_scenes=2, frames=8), "nonfinite": dict(max_pts=260, max_scenes=2, frames=12)}


def _window():
    """(arm, draw_case seed) of this round: `counts[arm]` seeds from `base` on (MMW_REF_FUZZ_BASE / MMW_REF_FUZZ_SCALE move and
    widen it for a one-off run), the legacy windows of rounds 3-5, the regression list."""
    base = int(os.environ.get("MMW_REF_FUZZ_BASE", WINDOW["base"]))
    scale = float(os.environ.get("MMW_REF_FUZZ_SCALE", "1"))
    out = []
    for arm, n in WINDOW["counts"].items():   # (the non-finite arm plants into all three draws in turn)
        out += [(arm, base + i, ARMS[(base + i) % 3] if arm == "nonfinite" else arm) for i in range(int(round(n * scale)))]
    if "MMW_REF_FUZZ_BASE" not in os.environ:
        for arm, (a, b) in WINDOW["legacy"].items():
            out += [(arm, sd, "wide") for sd in range(a, b)]
        out += [(r["arm"], int(r["seed"]), r.get("draw", "wide")) for r in WINDOW["regression"]]
    return out


def run_case(arm: str, seed: int, draw_arm: str = None):
    """One configuration, every scene, every frame: the live reference and oracle/c side by side.  Returns what was seen.
    `draw_arm`: which draw of tests/_fuzz.py the non-finite arm plants into (default: the arm itself, "wide" for nonfinite)."""
    from oracle import c_oracle as co
    from oracle.ref_import import load_reference
    from oracle.ref_runner import RefScene, _Recorder
    from tests._fuzz import plant_nonfinite

    nonfinite = arm == "nonfinite"
    draw_arm = draw_arm or ("wide" if nonfinite else arm)
    case = draw_case(seed, arm=draw_arm, **_SIZES[arm])
    kw, S, N, F = case["cfg"], case["S"], case["N"], case["F"]
    pts, cnt, dts = scene_inputs(case)
    planted = plant_nonfinite(case, pts, cnt, rate=0.3) if nonfinite else []
    const, utils, _ = load_reference()
    over = reference_overrides(kw)
    if "MOTION_MODEL" in over:
        over["MOTION_MODEL"] = getattr(const, over["MOTION_MODEL"])
    saved_defaults = utils.apply_DBscan.__defaults__
    utils.apply_DBscan.__defaults__ = (kw["db_eps"], kw["db_min_samples"])
    refs = []
    seen = dict(ints=0, dbscan=0, clustered=0, small_clouds=0, raised=0, planted=len(planted))
    try:
        cfg = co.default_config(**kw)
        for s in range(S):
            ref = RefScene(dict(over))
            refs.append(ref)
            orc = co.OracleScene(cfg, N)
            for f in range(F):
                c = int(cnt[f, s])
                if c == 0:
                    continue
                rows = pts[f, s, : max(c, 0)].astype(np.float64)
                ctx = (arm, seed, draw_arm, s, f)
                try:
                    with np.errstate(all="ignore"):
                        ra, rl = ref.track(rows, float(dts[f, s]))
                    raised = None
                except ValueError as e:
                    assert nonfinite, (ctx, e)
                    raised = str(e)
                except (ZeroDivisionError, np.linalg.LinAlgError) as e:
                    with pytest.raises(RuntimeError) as ei:     # the oracle's return code for the same exception
                        orc.track(rows, float(dts[f, s]))
                    assert ("rc=-3" if isinstance(e, ZeroDivisionError) else "rc=-2") in str(ei.value), (ctx, e)
                    break
                if raised is not None:
                    # sklearn's input validation raised out of apply_DBscan (Utils.py:272-278): the oracle returns ORC_E_NONFINITE_*
                    # with the same message kind (NaN when any value of the cloud is NaN, else infinity), and BOTH are stepped on --
                    # the state the exception leaves behind (frame in the ring, nothing clustered, nothing cleared) and every later
                    # frame must still agree; the row raises again on each frame it stays in the ring while the trigger holds
                    seen["raised"] += 1
                    assert raised.startswith("Input X contains"), (ctx, raised)
                    with pytest.raises(co.OracleNonFinite) as ei:
                        orc.track(rows, float(dts[f, s]))
                    assert raised.startswith(co.sklearn_message(ei.value.kind)), (ctx, raised, ei.value.kind)
                    if not kw.get("seek_inner"):   # (the frame's own call raised: the association came first and is comparable)
                        assert orc.last_db_n == co.DB_RAISED, ctx
                        assert np.array_equal(orc.last_assoc, _Recorder.assoc), (ctx, "association of a raising frame")
                else:
                    oa, ol = orc.track(rows, float(dts[f, s]))
                    assert np.array_equal(oa, ra), (ctx, "association")
                    assert (ol is None) == (rl is None), (ctx, "apply_DBscan call pattern")
                    if ol is not None:
                        seen["dbscan"] += 1
                        seen["clustered"] += int(len(rl) > 0 and rl.max() >= 0)
                        seen["small_clouds"] += int(len(rl) // 2 <= 5)
                        assert np.array_equal(ol, rl), (ctx, "DBSCAN labels", len(rl))
                    if kw.get("seek_inner") and not nonfinite:
                        oc, rc = orc.inner_calls(), ref.inner
                        assert len(oc) == len(rc), (ctx, "seek_inner_clusters calls")
                        for (tp, lab), (rtp, rlab) in zip(oc, rc):
                            assert tp == rtp and np.array_equal(lab, rlab), ctx
                    seen["ints"] += len(oa)
                assert orc.n_tracks == ref.n_tracks, ctx
                assert_tracks_match(orc.tracks(), ref.tracks(), ctx=f"{arm} seed {seed} s{s} f{f}", tol=1e-7)
                assert np.array_equal(orc.batch_ring(), ref.batch_ring()), ctx
                if not nonfinite and raised is None and f in (F // 2, F - 1):
                    of, oo = orc.features()
                    rf, ro = ref.features()
                    assert np.array_equal(oo, ro), (ctx, "feature owners")
                    if len(oo):
                        assert_feat_equal(of, rf.reshape(of.shape), ctx=f"{arm} seed {seed} s{s} f{f}")
            ref.close()
    finally:
        utils.apply_DBscan.__defaults__ = saved_defaults
        for r in refs:
            r.close()
    if nonfinite:
        assert planted
    return seen


_CHUNK = 16
_CASES = _window()
_TALLY = {}


@pytest.mark.parametrize("chunk", range(-(-len(_CASES) // _CHUNK)))
def test_live_reference_vs_c_oracle_on_this_rounds_window(chunk):
    """Sixteen (arm, seed) cases of the round's window per test id; a failure names its arm and seed in the assertion context
    (re-run one with `tests.test_reference_fuzz.run_case(arm, seed, draw_arm)`)."""
    for arm, seed, draw_arm in _CASES[chunk * _CHUNK: (chunk + 1) * _CHUNK]:
        seen = run_case(arm, seed, draw_arm)
        t = _TALLY.setdefault(arm, dict(cases=0))
        t["cases"] += 1
        for k, v in seen.items():
            t[k] = t.get(k, 0) + v


def test_the_window_exercised_what_it_is_for():
    """Runs after the chunks (file order).  The arms must have reached the decisions they exist for: DBSCAN calls on clouds of
    n // 2 <= 5 points WITH clusters in the small arm, raised ValueErrors in the non-finite arm.  The tally goes to
    gpurun_out/reference_fuzz_tally.json (copied to profiles/ by the round)."""
    if sum(t["cases"] for t in _TALLY.values()) < len(_CASES):
        pytest.skip("the window did not run in full (-k / -x)")
    assert _TALLY["small"]["small_clouds"] >= 200 and _TALLY["small"]["clustered"] >= 50, _TALLY["small"]
    assert _TALLY["threshold"]["dbscan"] >= 100, _TALLY["threshold"]
    assert _TALLY["nonfinite"]["raised"] >= 20, _TALLY["nonfinite"]
    assert _TALLY["wide"]["ints"] > 10000, _TALLY["wide"]
    out = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "reference_fuzz_tally.json"), "w") as fh:
            json.dump(dict(round=WINDOW["round"], base=WINDOW["base"], counts=WINDOW["counts"], tally=_TALLY), fh, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.mark.parametrize("seed", range(4))
def test_live_reference_normalize_data_with_nonfinite_rows(seed):
    """normalize_data (Utils.py:342-434) on raw rows with NaN / +-inf planted in every column: the full 4x4 homogeneous products
    of point_transform_to_standard_axis turn ONE non-finite coordinate into three NaN coordinates (0 * inf), which the scene
    filter drops; a non-finite doppler gives three NaN velocities on a row that is kept; peakVal passes through."""
    from oracle import c_oracle as co
    from oracle.ref_import import load_reference
    const, utils, _ = load_reference()
    from tests._fuzz import nonfinite_raw_rows
    raw = nonfinite_raw_rows(seed)
    det = {"x": list(raw[:, 0]), "y": list(raw[:, 1]), "z": list(raw[:, 2]), "doppler": list(raw[:, 3]), "peakVal": list(raw[:, 4])}
    with np.errstate(all="ignore"):
        want = utils.normalize_data(det)
    cfg = co.default_config(s_height=float(const.S_HEIGHT), s_tilt=float(const.S_TILT))
    got = co.normalize(cfg, raw)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want))
    assert np.allclose(got, want, rtol=0, atol=1e-12, equal_nan=True)
    assert np.isnan(want[:, 3:6]).any() and not np.isnan(want[:, :3]).any() and len(want) < 140
