"""Seeded random differential test of the HIP path against the C oracle (oracle/c; Tracking.py:664-703, Utils.py:250-291,
437-520): 64 random configurations x 12 frames x the four kernel layouts (tests/_layouts.py).

What a configuration draws (tests/_fuzz.py: draw_case): 1..8 scenes; points per frame anywhere in [1, 1024] (multiples of
64 are the exception); ring length 1..4; DB_EPS, DB_MIN_SAMPLES_MIN, DB_Z_WEIGHT, DB_RANGE_WEIGHT; TR_GATE,
TR_MAX_TRACKS, TR_VEL_THRES, the two lifetimes; CONST_ACC / CONST_VEL model; KF_ENABLE_EST and the estimator constants;
0..12 targets per scene (some appear or vanish half-way); ragged point counts; a different dt every frame; frames that
are skipped (n = 0, offline_main.py:56) and frames on which track() is called with an empty cloud (MMW_EMPTY_FRAME); one
configuration in eight runs ClusterTrack.seek_inner_clusters (Tracking.py:409-448).

Everything is compared bit for bit, every frame: association vectors, DBSCAN labels and call pattern, track count and
order, every fp64 field of every track, ring lengths, the global ring, inner-cluster labels, and the feature tensors."""
import numpy as np
import pytest

from tests._fuzz import N_CASES, SEED0, draw_case, scene_inputs
from tests._golden import assert_tracks_match
from tests._layouts import LAYOUTS, make_checked

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("seed", range(SEED0, SEED0 + N_CASES))
def test_random_configuration_vs_oracle(seed, layout):
    run_case(draw_case(seed), layout)


import json as _json
import os as _os
with open(_os.path.join(_os.path.dirname(__file__), "fuzz_window.json")) as _fh:
    _WINDOW = _json.load(_fh)
# the arms aimed at scikit-learn's size switches (tests/_fuzz.py: "small" = clouds answered by brute force, n // 2 <= 5, and right
# behind; "threshold" = 11 | 12, 60 | 61, 120 | 121, 240 | 241 points), on the seed window of the ROUND (tests/fuzz_window.json:
# the live-reference fuzz runs the same seeds against oracle/c in the build container)
ARM_SEED0 = int(_os.environ.get("MMW_FUZZ_ARM_SEED0", _WINDOW["base"]))
N_ARM = int(_os.environ.get("MMW_FUZZ_ARM_CASES", "32"))


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("arm", ["small", "threshold"])
@pytest.mark.parametrize("seed", range(ARM_SEED0, ARM_SEED0 + N_ARM))
def test_random_small_and_threshold_clouds_vs_oracle(seed, arm, layout):
    seen = run_case(draw_case(seed, arm=arm), layout)
    assert seen is not None


def run_case(case, layout):
    from mmwave_msc_amd import _lib
    from oracle import c_oracle as co
    seed = case["seed"]
    kw, S, N, F = case["cfg"], case["S"], case["N"], case["F"]
    pts, cnt, dts = scene_inputs(case)                       # [F,S,N,8] float32, [F,S] (0 = skipped, -1 = empty cloud), [F,S]
    sb = make_checked(S, N, layout, **kw)
    cfg = co.default_config(**kw)
    scenes = [co.OracleScene(cfg, N) for _ in range(S)]
    seen = dict(assigned=0, dbscan=0, clusters=0, empty=0, skipped=0)
    for f in range(F):
        # the oracle first: a frame on which the reference raises (numpy's LinAlgError for a singular 6 x 6 matrix,
        # ZeroDivisionError in _get_Rc, Tracking.py:299-312) must be a loud error of the same kind here, and ends the case
        want, failed = [None] * S, {}
        for s in range(S):
            c = int(cnt[f, s])
            if c != 0:
                try:
                    want[s] = scenes[s].track(pts[f, s, : max(c, 0)].astype(np.float64), float(dts[f, s]))
                except RuntimeError as e:
                    failed[s] = int(str(e).rsplit("rc=", 1)[1])
        if failed:
            codes = {-2: _lib.E_SINGULAR, -3: _lib.E_DIVZERO, -4: _lib.E_CAPACITY}   # (-4: a limit of include/mmw.h, the oracle keeps the same ones)
            with pytest.raises(_lib.MmwError) as ei:
                sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
            assert ei.value.code in {codes[v] for v in failed.values()}, (seed, f, failed, str(ei.value))
            err = sb.errors()
            for s in range(S):
                assert bool(err[s] & 7) == (s in failed), (seed, f, s, err[s], failed)
            seen["raised"] = seen.get("raised", 0) + 1
            break
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
        ntr = sb.num_tracks()
        trk = sb.tracks(cap=max(int(ntr.max()), 1))
        ln, rn = sb.batch_ring()
        inner = sb.inner_calls() if kw.get("seek_inner") else None
        feat, owner = sb.features_host() if f in (F // 2, F - 1) else (None, None)
        row = 0
        for s in range(S):
            c = int(cnt[f, s])
            if c == 0:                                     # the frame never reaches track()
                seen["skipped"] += 1
                assert dbn[s] == -1, (seed, f, s)
            else:
                n = max(c, 0)
                seen["empty"] += int(c < 0)
                oa, ol = want[s]
                assert np.array_equal(assoc[s, :n], oa), (seed, f, s)
                seen["assigned"] += int((oa >= 0).sum())
                assert (ol is None) == (dbn[s] < 0), (seed, f, s, dbn[s])
                if ol is not None:
                    seen["dbscan"] += 1
                    seen["clusters"] += int(ol.max() + 1) if len(ol) else 0
                    assert dbn[s] == len(ol) and np.array_equal(labels[s, : dbn[s]], ol), (seed, f, s)
                if inner is not None:
                    oc = scenes[s].inner_calls()
                    assert len(inner[s]) == len(oc), (seed, f, s)
                    for a, (_, b) in zip(inner[s], oc):
                        assert np.array_equal(a, b), (seed, f, s)
            assert ntr[s] == scenes[s].n_tracks, (seed, f, s)
            assert_tracks_match(trk[s, : ntr[s]], scenes[s].tracks(), ctx=f"seed {seed} f{f} s{s}", exact=True)
            assert np.array_equal(rn[s, : ln[s]], scenes[s].batch_ring()), (seed, f, s)
            if feat is not None:
                of, oo = scenes[s].features()
                k = len(oo)
                assert np.all(owner[row: row + k, 0] == s) and np.array_equal(owner[row: row + k, 1], oo), (seed, f, s)
                if k:
                    assert np.array_equal(feat[row: row + k], of), (seed, f, s)
                row += k
        if feat is not None:
            assert row == len(owner), (seed, f)
    else:
        sb.check()
        # the per-track rings of one scene, oldest frame first
        s = seed % S
        recs = scenes[s].tracks()
        for t in range(scenes[s].n_tracks):
            for k in range(int(recs[t]["ring_len"])):
                assert np.array_equal(sb.track_ring_frame(s, t, k), scenes[s].track_ring_frame(t, k)[: sb.ring_rows]), (seed, s, t, k)
    sb.close()
    case["seen"] = seen
    return seen


@pytest.mark.parametrize("layout", ["track_wise", "per_scene", "one_workgroup"])
@pytest.mark.parametrize("seed", [5, 56])
def test_scene_reset_after_a_reference_exception_vs_oracle(seed, layout):
    """Two of the random configurations run into the reference's ZeroDivisionError (`_get_Rc`, Tracking.py:299-312: KF_ENABLE_EST
    with a one-point cluster).  The scene gets its sticky error bit; `mmw_reset_scenes` must give exactly that scene a fresh
    TrackBuffer / BatchedData -- in the track-wise Kalman layout too, where the update lists of the frame before the reset
    still name the scene's (now stale, inf / NaN) records: they must not be predicted again, or the error bit comes straight
    back on the fresh scene -- while every other scene carries on; all scenes equal their oracles to the end."""
    from mmwave_msc_amd import _lib
    from oracle import c_oracle as co
    case = draw_case(seed)
    kw, S, N, F = case["cfg"], case["S"], case["N"], case["F"]
    pts, cnt, dts = scene_inputs(case)
    sb = make_checked(S, N, layout, **kw)
    cfg = co.default_config(**kw)
    scenes = [co.OracleScene(cfg, N) for _ in range(S)]
    resets = 0
    for f in range(F):
        want, failed = [None] * S, []
        for s in range(S):
            c = int(cnt[f, s])
            if c != 0:
                try:
                    want[s] = scenes[s].track(pts[f, s, : max(c, 0)].astype(np.float64), float(dts[f, s]))
                except RuntimeError:
                    failed.append(s)
        if failed:
            if seed == 56:   # the read-out form: no scene's error raises, the results of the others are there, the caller reads errors()
                assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f], check=False)
                for s in range(S):
                    if want[s] is not None:
                        assert np.array_equal(assoc[s, : max(int(cnt[f, s]), 0)], want[s][0]), (seed, f, s)
            else:
                with pytest.raises(_lib.MmwError):
                    sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
            mask = np.zeros(S, bool)
            mask[failed] = True
            assert np.array_equal(sb.errors() != 0, mask), (seed, f)
            sb.reset_scenes(mask)
            assert not sb.errors().any()
            for s in failed:
                scenes[s] = co.OracleScene(cfg, N)
            resets += len(failed)
        else:
            assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
            for s in range(S):
                if want[s] is not None:
                    n = max(int(cnt[f, s]), 0)
                    assert np.array_equal(assoc[s, :n], want[s][0]), (seed, f, s)
        assert not sb.errors().any(), (seed, f, sb.errors())
        ntr = sb.num_tracks()
        trk = sb.tracks(cap=max(int(ntr.max()), 1))
        for s in range(S):
            assert ntr[s] == scenes[s].n_tracks, (seed, f, s)
            assert_tracks_match(trk[s, : ntr[s]], scenes[s].tracks(), ctx=f"seed {seed} f{f} s{s}", exact=True)
    assert resets >= 1
    sb.check()
    sb.close()


N_NONFINITE = int(_os.environ.get("MMW_FUZZ_NF_CASES", "24"))      # (a one-off wider window: MMW_FUZZ_NF_CASES=512, as MMW_FUZZ_CASES)
NF_SEED0 = int(_os.environ.get("MMW_FUZZ_NF_SEED0", "5000"))


@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("seed", range(NF_SEED0, NF_SEED0 + N_NONFINITE))
def test_random_configuration_with_nonfinite_rows_vs_oracle(seed, layout):
    """The non-finite arm (tests/_fuzz.py: plant_nonfinite): NaN / +inf / -inf in random columns of random rows -- x, y, z, the
    velocities, doppler, peakVal; rows that stay unassigned and rows a track would have taken.  The reference raises ValueError
    out of apply_DBscan on every frame such a row is in the global ring while the trigger holds (sklearn's input validation,
    Utils.py:272-278; pinned on the live reference by tests/test_reference_fuzz.py and the `nonfinite` golden).  Here: db_n =
    MMW_DB_RAISED for exactly those scenes and frames, the sticky bit names sklearn's message (NaN wins over infinity), the
    frame's association and every scene's state equal the oracle's bit for bit, and after mmw_clear_errors the scene carries
    on as the reference does after the exception.  seek_inner configurations can raise INSIDE _associate_points_to_tracks (an
    assigned point's NaN doppler reaches the inner apply_DBscan): the scene is flagged, then reset on both sides."""
    from mmwave_msc_amd import _lib
    from oracle import c_oracle as co
    from tests._fuzz import plant_nonfinite
    case = draw_case(seed)
    kw, S, N, F = case["cfg"], case["S"], case["N"], case["F"]
    pts, cnt, dts = scene_inputs(case)
    planted = plant_nonfinite(case, pts, cnt, rate=0.3)
    sb = make_checked(S, N, layout, **kw)
    cfg = co.default_config(**kw)
    scenes = [co.OracleScene(cfg, N) for _ in range(S)]
    NF = _lib.ERRBIT_NONFINITE_NAN | _lib.ERRBIT_NONFINITE_INF
    n_raised = n_inner = 0
    for f in range(F):
        want, raised, inner_raised, failed = [None] * S, {}, {}, {}
        for s in range(S):
            c = int(cnt[f, s])
            if c == 0:
                continue
            try:
                want[s] = scenes[s].track(pts[f, s, : max(c, 0)].astype(np.float64), float(dts[f, s]))
            except co.OracleNonFinite as e:
                bit = _lib.ERRBIT_NONFINITE_NAN if e.kind == "NaN" else _lib.ERRBIT_NONFINITE_INF
                if scenes[s].last_db_n == co.DB_RAISED:
                    raised[s] = bit
                    want[s] = (scenes[s].last_assoc, None)
                else:
                    inner_raised[s] = bit      # inside seek_inner_clusters: the reference's track() stopped half-way
            except RuntimeError as e:
                failed[s] = int(str(e).rsplit("rc=", 1)[1])
        if failed:     # LinAlgError / ZeroDivisionError / a capacity limit: covered by test_random_configuration_vs_oracle
            break
        # (a scene that goes on past the reference's MID-frame exception may meet the ZeroDivisionError / LinAlgError the reference
        #  never got to -- seed 41205 of a wide window: that must not end the step's read-out; its bits are looked at below)
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f], raise_nonfinite=False, check=not inner_raised)
        err = sb.errors()
        for s in range(S):
            want_bit = raised.get(s, 0) | inner_raised.get(s, 0)
            assert (err[s] & NF) == want_bit or (s in inner_raised and (err[s] & NF)), (seed, f, s, err[s], raised, inner_raised)
            if s not in inner_raised:   # (a scene that went on past the reference's MID-frame exception may raise at its trigger too, or
                #  meet the ZeroDivisionError / LinAlgError the reference never got to: it is reset below)
                assert (err[s] & ~NF) == 0, (seed, f, s, err[s])
                assert (dbn[s] == _lib.DB_RAISED) == (s in raised), (seed, f, s, dbn[s])
        if raised or inner_raised:
            with pytest.raises((ValueError, ZeroDivisionError, np.linalg.LinAlgError) if inner_raised else ValueError):
                sb.check()
            sb.clear_errors(NF)
        if inner_raised:
            mask = np.zeros(S, bool)
            mask[list(inner_raised)] = True
            sb.reset_scenes(mask)
            for s in inner_raised:
                scenes[s] = co.OracleScene(cfg, N)
            n_inner += len(inner_raised)
        n_raised += len(raised)
        ntr = sb.num_tracks()
        trk = sb.tracks(cap=max(int(ntr.max()), 1))
        ln, rn = sb.batch_ring()
        feat, owner = sb.features_host() if f in (F // 2, F - 1) else (None, None)
        row = 0
        for s in range(S):
            c = int(cnt[f, s])
            if c != 0 and s not in inner_raised:
                n = max(c, 0)
                oa, ol = want[s]
                assert np.array_equal(assoc[s, :n], oa), (seed, f, s)
                assert (ol is None) == (dbn[s] < 0), (seed, f, s, dbn[s])
                if ol is not None:
                    assert dbn[s] == len(ol) and np.array_equal(labels[s, : dbn[s]], ol), (seed, f, s)
            assert ntr[s] == scenes[s].n_tracks, (seed, f, s)
            assert_tracks_match(trk[s, : ntr[s]], scenes[s].tracks(), ctx=f"seed {seed} f{f} s{s}", exact=True)
            assert np.array_equal(rn[s, : ln[s]], scenes[s].batch_ring()), (seed, f, s)
            if feat is not None:
                of, oo = scenes[s].features()
                k = len(oo)
                assert np.all(owner[row: row + k, 0] == s) and np.array_equal(owner[row: row + k, 1], oo), (seed, f, s)
                if k:
                    assert np.array_equal(feat[row: row + k], of, equal_nan=True), (seed, f, s)
                row += k
        if feat is not None:
            assert row == len(owner), (seed, f)
    sb.check()
    sb.close()
    assert planted
    case["seen"] = dict(raised=n_raised, inner=n_inner)
