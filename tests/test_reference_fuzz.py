"""Randomised differential test of the C oracle (oracle/c) against the LIVE reference (/root/reference/src imported in
this container through oracle/ref_import.py; filterpy from oracle/filterpy_shim): what the golden fixtures pin on 21
recorded scenarios, checked again on fresh seeds with random constant overrides -- the configurations of
tests/_fuzz.py (the same generator drives the GPU fuzz test), smaller.

`-m reference`: skipped where /root/reference is absent (the GPU box).  Integers -- association vectors, DBSCAN labels,
track count and order, point counts, ring lengths, static flags, feature owners -- must be equal; fp64 state within
tests/_golden.py's tolerance (BLAS / LAPACK summation order differs from the oracle's fixed order by a few ULP).

DB_EPS and DB_MIN_SAMPLES_MIN are bound as default arguments of Utils.apply_DBscan when the reference is imported
(Utils.py:250); a run with other values re-binds `apply_DBscan.__defaults__`, which is what editing constants.py does."""
import numpy as np
import pytest

from tests._fuzz import draw_case, reference_overrides, scene_inputs
from tests._golden import assert_feat_equal, assert_tracks_match

pytestmark = pytest.mark.reference

N_REF_CASES = 24


@pytest.mark.parametrize("seed", range(N_REF_CASES))
def test_live_reference_vs_c_oracle_on_fresh_seeds(seed):
    from oracle import c_oracle as co
    from oracle.ref_import import load_reference
    from oracle.ref_runner import RefScene

    # the reference is Python + a Python DBSCAN metric: small contexts (<= 2 scenes, <= 300 points, 10 frames)
    case = draw_case(1000 + seed, max_pts=300, max_scenes=2, frames=10)
    kw, S, N, F = case["cfg"], case["S"], case["N"], case["F"]
    pts, cnt, dts = scene_inputs(case)
    const, utils, _ = load_reference()
    over = reference_overrides(kw)
    if "MOTION_MODEL" in over:
        over["MOTION_MODEL"] = getattr(const, over["MOTION_MODEL"])
    saved_defaults = utils.apply_DBscan.__defaults__
    utils.apply_DBscan.__defaults__ = (kw["db_eps"], kw["db_min_samples"])
    refs = []
    try:
        cfg = co.default_config(**kw)
        n_int = n_db = 0
        for s in range(S):
            ref = RefScene(dict(over))
            refs.append(ref)
            orc = co.OracleScene(cfg, N)
            for f in range(F):
                c = int(cnt[f, s])
                if c == 0:
                    continue
                rows = pts[f, s, : max(c, 0)].astype(np.float64)
                try:
                    ra, rl = ref.track(rows, float(dts[f, s]))
                except (ZeroDivisionError, np.linalg.LinAlgError) as e:
                    with pytest.raises(RuntimeError) as ei:     # the oracle's return code for the same exception
                        orc.track(rows, float(dts[f, s]))
                    assert ("rc=-3" if isinstance(e, ZeroDivisionError) else "rc=-2") in str(ei.value), (seed, s, f, e)
                    break
                oa, ol = orc.track(rows, float(dts[f, s]))
                assert np.array_equal(oa, ra), (seed, s, f, "association")
                assert (ol is None) == (rl is None), (seed, s, f, "apply_DBscan call pattern")
                if ol is not None:
                    n_db += 1
                    assert np.array_equal(ol, rl), (seed, s, f, "DBSCAN labels")
                if kw.get("seek_inner"):
                    oc, rc = orc.inner_calls(), ref.inner
                    assert len(oc) == len(rc), (seed, s, f, "seek_inner_clusters calls")
                    for (tp, lab), (rtp, rlab) in zip(oc, rc):
                        assert tp == rtp and np.array_equal(lab, rlab), (seed, s, f)
                assert orc.n_tracks == ref.n_tracks, (seed, s, f)
                assert_tracks_match(orc.tracks(), ref.tracks(), ctx=f"seed {seed} s{s} f{f}", tol=1e-7)
                assert np.array_equal(orc.batch_ring(), ref.batch_ring()), (seed, s, f)
                n_int += len(oa)
                if f in (F // 2, F - 1):
                    of, oo = orc.features()
                    rf, ro = ref.features()
                    assert np.array_equal(oo, ro), (seed, s, f, "feature owners")
                    if len(oo):
                        assert_feat_equal(of, rf.reshape(of.shape), ctx=f"seed {seed} s{s} f{f}")
            ref.close()
        assert n_int > 0
    finally:
        utils.apply_DBscan.__defaults__ = saved_defaults
        for r in refs:
            r.close()


N_NONFINITE_CASES = 16


@pytest.mark.parametrize("seed", range(N_NONFINITE_CASES))
def test_live_reference_vs_c_oracle_with_nonfinite_rows(seed):
    """NaN / +-inf planted in random columns of random rows (tests/_fuzz.py: plant_nonfinite).  Where the reference raises
    ValueError out of apply_DBscan (sklearn's input validation, Utils.py:272-278) the oracle returns ORC_E_NONFINITE_* with the
    same message kind (NaN when any value of the cloud is NaN, else infinity), and BOTH are stepped on: the state the
    exception leaves behind (frame in the ring, nothing clustered, nothing cleared) and every later frame must still agree --
    the row raises again on each frame it stays in the ring while the trigger holds."""
    from oracle import c_oracle as co
    from oracle.ref_import import load_reference
    from oracle.ref_runner import RefScene
    from tests._fuzz import plant_nonfinite

    case = draw_case(3000 + seed, max_pts=260, max_scenes=2, frames=12)
    kw, S, N, F = case["cfg"], case["S"], case["N"], case["F"]
    pts, cnt, dts = scene_inputs(case)
    planted = plant_nonfinite(case, pts, cnt, rate=0.3)
    const, utils, _ = load_reference()
    over = reference_overrides(kw)
    if "MOTION_MODEL" in over:
        over["MOTION_MODEL"] = getattr(const, over["MOTION_MODEL"])
    saved_defaults = utils.apply_DBscan.__defaults__
    utils.apply_DBscan.__defaults__ = (kw["db_eps"], kw["db_min_samples"])
    refs = []
    n_raised = 0
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
                ctx = (seed, s, f)
                try:
                    with np.errstate(all="ignore"):
                        ra, rl = ref.track(rows, float(dts[f, s]))
                    raised = None
                except ValueError as e:
                    raised = str(e)
                except (ZeroDivisionError, np.linalg.LinAlgError) as e:
                    with pytest.raises(RuntimeError) as ei:
                        orc.track(rows, float(dts[f, s]))
                    assert ("rc=-3" if isinstance(e, ZeroDivisionError) else "rc=-2") in str(ei.value), (ctx, e)
                    break
                if raised is not None:
                    n_raised += 1
                    assert raised.startswith("Input X contains"), (ctx, raised)
                    with pytest.raises(co.OracleNonFinite) as ei:
                        orc.track(rows, float(dts[f, s]))
                    assert raised.startswith(co.sklearn_message(ei.value.kind)), (ctx, raised, ei.value.kind)
                    from oracle.ref_runner import _Recorder
                    if not kw.get("seek_inner"):   # (the frame's own call raised: the association came first and is comparable)
                        assert orc.last_db_n == co.DB_RAISED, ctx
                        assert np.array_equal(orc.last_assoc, _Recorder.assoc), (ctx, "association of a raising frame")
                else:
                    oa, ol = orc.track(rows, float(dts[f, s]))
                    assert np.array_equal(oa, ra), (ctx, "association")
                    assert (ol is None) == (rl is None), (ctx, "apply_DBscan call pattern")
                    if ol is not None:
                        assert np.array_equal(ol, rl), (ctx, "DBSCAN labels")
                assert orc.n_tracks == ref.n_tracks, ctx
                assert_tracks_match(orc.tracks(), ref.tracks(), ctx=f"seed {seed} s{s} f{f}", tol=1e-7)
                assert np.array_equal(orc.batch_ring(), ref.batch_ring()), ctx
            ref.close()
    finally:
        utils.apply_DBscan.__defaults__ = saved_defaults
        for r in refs:
            r.close()
    assert planted


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
