"""GPU parity: the HIP path (through the C-ABI) against (a) the golden vectors
recorded from the reference and (b) the CPU oracle on the same inputs.

(a) ints bit-exact, fp64 within tests/_golden.py's stated tolerance.
(b) everything bit-exact: the oracle and the kernels share one fixed operation order.
"""
import os

import numpy as np
import pytest

from tests._golden import (GOLDEN, assert_feat_equal, assert_tracks_match, load_scenario,
                           overrides_to_cfg_kwargs, scenario_names, scenario_tol)

pytestmark = pytest.mark.gpu


from tests._layouts import LAYOUTS, make_checked

_LAYOUT = {"name": "per_scene"}


@pytest.fixture(params=LAYOUTS, autouse=True)
def kalman_layout(request):
    """Every test of this file runs under the four kernel layouts (tests/_layouts.py): the bulk kernels with the Kalman
    kernels per scene, the same laid out over tracks (by default only for contexts with more than 1024 four-track waves,
    mmw_config.kalman_dense_min_units), a third time with the small-cloud DBSCAN workers on the side stream (k_chain beside
    k_track, mmw_config.chain_side_stream: by default only for contexts of more than 512 scenes), and a fourth time with the
    one-workgroup step (k_scene.hip, mmw_config.fused_step: by default only for contexts of 257..512 scenes).  A context
    whose configuration forbids the layout is SKIPPED with the reason, never run as a duplicate of another layout."""
    _LAYOUT["name"] = request.param
    yield
    _LAYOUT["name"] = "per_scene"


def _mk(n_scenes, max_pts, **kw):
    return make_checked(n_scenes, max_pts, _LAYOUT["name"], **kw)


@pytest.mark.parametrize("name", scenario_names())
def test_golden_scenario(name):
    from oracle import c_oracle as co
    g = load_scenario(name)
    kw = overrides_to_cfg_kwargs(g["overrides"])
    n = g["pts"].shape[1]
    sb = _mk(1, n, **kw)
    orc = co.OracleScene(co.default_config(**kw), n)
    if "batch_init" in g:   # BatchedData(init_data) (Tracking.py:38-41)
        sb.set_batch_frame(0, g["batch_init"])
        orc.set_batch_frame(g["batch_init"])
    resize = {int(a): int(b) for a, b in g["overrides"].get("BATCH_RESIZE", [])}
    for f in range(g["pts"].shape[0]):
        c = int(g["cnt"][f])
        if f in resize:     # BatchedData.change_buffer_size (Tracking.py:60-64) before this frame's track()
            sb.set_batch_size(resize[f])
            orc.set_batch_size(resize[f])
        pts = np.zeros((1, n, 8))
        pts[0, :c] = g["pts"][f, :c]
        track_empty = bool(g["overrides"].get("TRACK_EMPTY"))   # the scenario calls track() on its empty frames
        n_arg = c if (c > 0 or not track_empty) else -1             # MMW_EMPTY_FRAME
        raised = int(g["raised"][f]) if "raised" in g else 0
        assoc, labels, dbn = sb.step_host(pts, np.array([n_arg], np.int32), np.array([g["dt"][f]]), raise_nonfinite=False)
        if c == 0 and not track_empty:
            assert dbn[0] == -1
            assert sb.num_tracks()[0] == g["n_tracks"][f]
            continue
        if raised:
            # the reference raised ValueError out of apply_DBscan (a NaN / an infinite value in the ring, Utils.py:272-278):
            # db_n = MMW_DB_RAISED, the scene's sticky bit says which message, mmw_check reports MMW_E_NONFINITE -- a ValueError
            # in the Python binding -- and after mmw_clear_errors the scene carries on in the state the exception left
            from mmwave_msc_amd import _lib
            assert dbn[0] == _lib.DB_RAISED, f"{name} f{f}: db_n {dbn[0]}"
            assert sb.errors()[0] == (_lib.ERRBIT_NONFINITE_NAN, _lib.ERRBIT_NONFINITE_INF)[raised - 1], f"{name} f{f}: which ValueError"
            with pytest.raises(ValueError) as ei:
                sb.check()
            assert ei.value.code == _lib.E_NONFINITE and ("NaN" if raised == 1 else "infinity") in str(ei.value)
            sb.clear_errors(_lib.ERRBIT_NONFINITE_NAN | _lib.ERRBIT_NONFINITE_INF)
            with pytest.raises(co.OracleNonFinite):
                orc.track(pts[0, :c], float(g["dt"][f]))
            o_assoc, o_lab = orc.last_assoc, None
            dbn[0] = -1   # (the recording's db_n of a raising frame: no labels)
        else:
            assert sb.errors()[0] == 0, f"{name} f{f}"
            o_assoc, o_lab = orc.track(pts[0, :c], float(g["dt"][f]))
        # (a) golden
        assert np.array_equal(assoc[0, :c], g["assoc"][f, :c]), f"{name} f{f}: association vs golden"
        assert dbn[0] == g["db_n"][f], f"{name} f{f}: db_n {dbn[0]} vs {g['db_n'][f]}"
        if dbn[0] >= 0:
            assert np.array_equal(labels[0, : dbn[0]], g["labels"][f, : dbn[0]]), f"{name} f{f}: labels vs golden"
        nt = int(g["n_tracks"][f])
        assert sb.num_tracks()[0] == nt
        trk = sb.tracks(cap=max(nt, 1))[0, :nt]
        assert_tracks_match(trk, g["tracks"][f, :nt], ctx=f"{name} f{f} vs golden", tol=scenario_tol(g))
        ln, rn = sb.batch_ring()
        assert ln[0] == g["ring_len"][f] and np.array_equal(rn[0, : ln[0]], g["ring_n"][f, : ln[0]])
        # (b) oracle, bit-exact
        assert np.array_equal(assoc[0, :c], o_assoc)
        assert (o_lab is None) == (dbn[0] < 0)
        if o_lab is not None:
            assert np.array_equal(labels[0, : dbn[0]], o_lab)
        assert_tracks_match(trk, orc.tracks(), ctx=f"{name} f{f} vs oracle", exact=True)
        if kw.get("seek_inner"):   # ClusterTrack.seek_inner_clusters: the calls of this frame, golden and oracle
            calls = sb.inner_calls()[0]
            assert len(calls) == int(g["inner_calls"][f]) == len(orc.inner_calls()), f"{name} f{f}: seek_inner_clusters calls"
            for q, lab_in in enumerate(calls):
                assert np.array_equal(lab_in, g["inner_labels"][f, q, : g["inner_n"][f, q]]), f"{name} f{f}: inner labels of call {q}"
                assert np.array_equal(lab_in, orc.inner_calls()[q][1])
        feat, owner = sb.features_host()
        nf = int(g["n_feat"][f])
        assert len(owner) == nf
        if nf:
            assert np.array_equal(owner[:, 1], g["owner"][f, :nf]) and np.all(owner[:, 0] == 0)
            assert_feat_equal(feat, g["feat"][f, :nf], ctx=f"{name} f{f}")
            o_feat, o_own = orc.features()
            assert np.array_equal(feat, o_feat, equal_nan=True), f"{name} f{f}: features vs oracle"
    sb.close()


def test_multi_scene_vs_oracle():
    """32 scenes with different target counts stepped together; every scene must equal
    its own oracle run bit for bit (ints, fp64 state, ring contents, features)."""
    from mmwave_msc_amd.synth import make_batch
    from oracle import c_oracle as co
    S, N, F = 32, 256, 14
    kw = dict(tr_max_tracks=4)
    sb = _mk(S, N, **kw)
    cfg = co.default_config(**kw)
    scenes = [co.OracleScene(cfg, N) for _ in range(S)]
    pts = np.zeros((F, S, N, 8), np.float32)
    cnt = np.zeros((F, S), np.int32)
    dts = np.zeros((F, S))
    for s in range(S):
        p, c, d = make_batch([700 + s], F, N, 1 + s % 5, ragged=(s % 3 == 0))
        pts[:, s], cnt[:, s], dts[:, s] = p[:, 0], c[:, 0], d[:, 0]
    for f in range(F):
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
        ntr = sb.num_tracks()
        trk = sb.tracks(cap=8)
        feat, owner = sb.features_host()
        row = 0
        for s in range(S):
            c = cnt[f, s]
            oa, ol = scenes[s].track(pts[f, s, :c].astype(np.float64), dts[f, s])
            assert np.array_equal(assoc[s, :c], oa), (f, s)
            assert (ol is None) == (dbn[s] < 0), (f, s)
            if ol is not None:
                assert np.array_equal(labels[s, : dbn[s]], ol), (f, s)
            assert ntr[s] == scenes[s].n_tracks
            assert_tracks_match(trk[s, : ntr[s]], scenes[s].tracks(), ctx=f"f{f} s{s}", exact=True)
            of, oo = scenes[s].features()
            k = len(oo)
            assert np.all(owner[row: row + k, 0] == s) and np.array_equal(owner[row: row + k, 1], oo)
            if k:
                assert np.array_equal(feat[row: row + k], of), (f, s)
            row += k
        assert row == len(owner)
    # ring contents of one scene
    s = 5
    for t in range(scenes[s].n_tracks):
        rec = scenes[s].tracks()[t]
        for k in range(rec["ring_len"]):
            # (the rows both sides store: the oracle keeps ring_rows = 64 of a frame, a context that forms the cluster statistics per
            #  track -- mmw_config.split_stats -- keeps every row)
            want = scenes[s].track_ring_frame(t, k)
            assert np.array_equal(sb.track_ring_frame(s, t, k)[: len(want)], want[: sb.ring_rows])
    sb.close()


@pytest.mark.parametrize("case", [
    dict(db_min_samples=4, db_eps=0.05),                       # cores appear and vanish inside plain clutter
    dict(db_min_samples=6, db_eps=0.12),
    dict(db_min_samples=12, db_eps=0.3, db_z_weight=0.0),      # z ignored by the metric
    dict(db_min_samples=8, db_eps=0.2, db_range_weight=-0.02), # the weight grows with range
    dict(db_min_samples=5, db_eps=0.1, scale=6.0),             # clutter spread over tens of metres: the cell torus wraps
    dict(db_min_samples=3, db_eps=0.02, tr_max_tracks=2),
])
def test_dbscan_screen_decisions_vs_oracle(case):
    """The no-core-point screens (cell count in k_track, pair count in k_post) only ever skip a BallTree run
    whose result is "all noise".  Configurations that put many scenes right at the boundary between "a core
    point exists" and "none does": labels, cluster spawns and every later frame must equal the oracle."""
    from mmwave_msc_amd.synth import make_batch
    from oracle import c_oracle as co
    case = dict(case)
    scale = case.pop("scale", 1.0)
    S, N, F = 48, 128, 10
    kw = dict(tr_max_tracks=3)
    kw.update(case)
    sb = _mk(S, N, **kw)
    cfg = co.default_config(**kw)
    scenes = [co.OracleScene(cfg, N) for _ in range(S)]
    pts = np.zeros((F, S, N, 8), np.float32)
    cnt = np.zeros((F, S), np.int32)
    dts = np.zeros((F, S))
    for s in range(S):
        p, c, d = make_batch([9100 + s], F, N, s % 3, ragged=(s % 4 == 1))
        pts[:, s], cnt[:, s], dts[:, s] = p[:, 0], c[:, 0], d[:, 0]
    if scale != 1.0:
        pts[..., 0] *= scale
        pts[..., 1] = np.minimum(pts[..., 1] * scale, 30.0).astype(np.float32)  # keep 1 - range_w * y positive
    n_db = n_clustered = 0
    for f in range(F):
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
        ntr = sb.num_tracks()
        trk = sb.tracks(cap=16)
        for s in range(S):
            c = cnt[f, s]
            oa, ol = scenes[s].track(pts[f, s, :c].astype(np.float64), dts[f, s])
            assert np.array_equal(assoc[s, :c], oa), (f, s)
            assert (ol is None) == (dbn[s] < 0), (f, s)
            if ol is not None:
                n_db += 1
                n_clustered += int(ol.max() >= 0)
                assert dbn[s] == len(ol), (f, s)
                assert np.array_equal(labels[s, : dbn[s]], ol), (f, s, case)
            assert ntr[s] == scenes[s].n_tracks, (f, s)
            assert_tracks_match(trk[s, : ntr[s]], scenes[s].tracks(), ctx=f"f{f} s{s}", exact=True)
    sb.check()
    sb.close()
    # the case is only meaningful if both outcomes occur
    assert n_db > 50 and 0 < n_clustered < n_db, (n_db, n_clustered)


def _grid_scene(seed, n_frames, n_pts, n_targets):
    """Targets on a 1.3 m grid (4 rows in y), 24 points each per frame, the rest clutter; fp32-representable."""
    rng = np.random.default_rng(seed)
    cols = (n_targets + 3) // 4
    centres = np.array([[-0.65 * (cols - 1) + 1.3 * (k // 4), 1.5 + 1.3 * (k % 4)] for k in range(n_targets)])
    vel = rng.normal(0.0, 0.08, size=(n_targets, 2))
    out = np.zeros((n_frames, n_pts, 8))
    per = 24
    for f in range(n_frames):
        c = centres + vel * 0.1 * f
        rows = []
        for k in range(n_targets):
            r = np.zeros((per, 8))
            r[:, 0:2] = c[k] + rng.normal(0.0, 0.08, size=(per, 2))
            r[:, 2] = rng.uniform(0.3, 1.5, size=per)
            r[:, 3:5] = vel[k] + rng.normal(0.0, 0.03, size=(per, 2))
            r[:, 5] = rng.normal(0.0, 0.03, size=per)
            rows.append(r)
        ncl = n_pts - per * n_targets
        cl = np.zeros((ncl, 8))
        cl[:, 0] = rng.uniform(-6.0, 6.0, size=ncl); cl[:, 1] = rng.uniform(0.3, 7.5, size=ncl); cl[:, 2] = rng.uniform(0.05, 2.4, size=ncl)
        cl[:, 3:6] = rng.normal(0.0, 0.05, size=(ncl, 3))
        fr = np.concatenate(rows + [cl])
        fr[:, 6] = rng.normal(0.0, 0.3, size=n_pts); fr[:, 7] = rng.gamma(1.0, 30.0, size=n_pts)
        out[f] = fr[rng.permutation(n_pts)]
    return out.astype(np.float32)


def test_many_tracks_vs_oracle():
    """More than 16 tracks per scene: the gate-record chunks of k_track, the second statistics round, the track
    loops of the batched Kalman kernels (tracks beyond 4 * waves-per-scene) and the per-track ring bookkeeping."""
    from oracle import c_oracle as co
    S, N, F = 4, 640, 10
    kw = dict(tr_max_tracks=28, db_min_samples=12)
    sb = _mk(S, N, **kw)
    cfg = co.default_config(**kw)
    scenes = [co.OracleScene(cfg, N) for _ in range(S)]
    pts = np.stack([_grid_scene(4300 + s, F, N, 18 + 2 * s) for s in range(S)], axis=1)  # [F][S][N][8]
    cnt = np.full((F, S), N, np.int32)
    dts = np.full((F, S), 0.1)
    most = 0
    for f in range(F):
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
        ntr = sb.num_tracks()
        trk = sb.tracks(cap=64)
        for s in range(S):
            oa, ol = scenes[s].track(pts[f, s].astype(np.float64), dts[f, s])
            assert np.array_equal(assoc[s], oa), (f, s)
            assert (ol is None) == (dbn[s] < 0), (f, s)
            if ol is not None:
                assert np.array_equal(labels[s, : dbn[s]], ol), (f, s)
            assert ntr[s] == scenes[s].n_tracks, (f, s)
            assert_tracks_match(trk[s, : ntr[s]], scenes[s].tracks(), ctx=f"f{f} s{s}", exact=True)
            most = max(most, int(ntr[s]))
    sb.check()
    sb.close()
    assert most > 16, most


def test_more_than_16_tracks_below_the_one_workgroup_cap_vs_oracle():
    """test_many_tracks_vs_oracle's scenes with track_cap = 40: below 64, so the one-workgroup step (k_scene.hip) and the
    track-wise Kalman layout are available and every layout runs its OWN kernels -- k_scene's slow path for the tracks past
    the kRes = 16 records it keeps in LDS (prediction, ring bookkeeping, statistics and update through global memory), the
    dense update lists with more than 16 tracks per scene.  Every frame against the oracle (Tracking.py:664-703)."""
    from oracle import c_oracle as co
    S, N, F = 8, 640, 10
    kw = dict(tr_max_tracks=28, db_min_samples=12, track_cap=40)
    sb = _mk(S, N, **kw)
    assert sb.track_cap == 40
    if _LAYOUT["name"] == "one_workgroup":
        assert sb.step_kind() == 1
    cfg = co.default_config(**kw)
    scenes = [co.OracleScene(cfg, N) for _ in range(S)]
    pts = np.stack([_grid_scene(4400 + s, F, N, 17 + s) for s in range(S)], axis=1)  # [F][S][N][8]: 17 .. 24 targets
    cnt = np.full((F, S), N, np.int32)
    dts = np.full((F, S), 0.1)
    most = 0
    for f in range(F):
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
        ntr = sb.num_tracks()
        trk = sb.tracks(cap=40)
        feat, owner = sb.features_host()
        row = 0
        for s in range(S):
            oa, ol = scenes[s].track(pts[f, s].astype(np.float64), dts[f, s])
            assert np.array_equal(assoc[s], oa), (f, s)
            assert (ol is None) == (dbn[s] < 0), (f, s)
            if ol is not None:
                assert np.array_equal(labels[s, : dbn[s]], ol), (f, s)
            assert ntr[s] == scenes[s].n_tracks, (f, s)
            assert_tracks_match(trk[s, : ntr[s]], scenes[s].tracks(), ctx=f"f{f} s{s}", exact=True)
            of, oo = scenes[s].features()
            k = len(oo)
            assert np.all(owner[row: row + k, 0] == s) and np.array_equal(owner[row: row + k, 1], oo), (f, s)
            if k:
                assert np.array_equal(feat[row: row + k], of), (f, s)
            row += k
            most = max(most, int(ntr[s]))
        assert row == len(owner)
    if _LAYOUT["name"] == "one_workgroup":
        assert sb.step_kind() == 1   # ... to the end
    sb.check()
    sb.close()
    assert most > 16, most


def test_normalize_golden_and_oracle():
    from oracle import c_oracle as co
    z = np.load(os.path.join(GOLDEN, "normalize.npz"))
    raw = z["raw"]
    n = raw.shape[0]
    sb = _mk(2, n)
    rawb = np.zeros((2, n, 5))
    rawb[0] = raw
    rawb[1, : n // 2] = raw[n // 2: n // 2 * 2]
    pts, n_out = sb.normalize_host(rawb, np.array([n, n // 2], np.int32))
    assert n_out[0] == z["out"].shape[0]
    assert np.allclose(pts[0, : n_out[0]], z["out"], rtol=0, atol=1e-12)
    cfg = co.default_config()
    assert np.array_equal(pts[0, : n_out[0]], co.normalize(cfg, raw))
    assert np.array_equal(pts[1, : n_out[1]], co.normalize(cfg, raw[n // 2: n // 2 * 2]))
    sb.close()


def test_fp32_row_entry_is_bit_equal_to_the_fp64_entry():
    """mmw_step_f32 (32-byte rows promoted to fp64 as the association kernel loads them, include/mmw.h) against mmw_step on the
    promoted rows: every output and every track field bit-equal, every frame, in every layout; and against the oracle on the
    last frame.  mmw_normalize_f32 likewise against mmw_normalize (Utils.py:342-434).  Ragged counts, skipped and empty frames."""
    import torch
    from mmwave_msc_amd.synth import make_batch
    from oracle import c_oracle as co
    S, N, F, T = 40, 200, 12, 5
    kw = dict(tr_max_tracks=T)
    pts = np.zeros((F, S, N, 8), np.float32); cnt = np.zeros((F, S), np.int32); dts = np.zeros((F, S))
    for s in range(S):
        p, c, d = make_batch([6100 + s], F, N, s % (T + 1), ragged=(s % 2 == 0))
        pts[:, s], cnt[:, s], dts[:, s] = p[:, 0], c[:, 0], d[:, 0]
    cnt[3, 5] = 0; cnt[4, 6] = -1; cnt[7, ::9] = 0
    dev = torch.device("cuda:0")
    a, b = _mk(S, N, **kw), _mk(S, N, **kw)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        a.follow_torch_stream(st); b.follow_torch_stream(st)
        d_cnt = torch.from_numpy(cnt).to(dev); d_dt = torch.from_numpy(dts).to(dev)
        out = [dict(a=torch.full((S, N), -7, dtype=torch.int32, device=dev), l=torch.full((S, a.UM), -7, dtype=torch.int32, device=dev),
                    n=torch.full((S,), -7, dtype=torch.int32, device=dev)) for _ in range(2)]
        for f in range(F):
            p32 = torch.from_numpy(pts[f]).to(dev)
            p64 = p32.double()
            a.step_dev(p64.data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr(), out[0]["a"].data_ptr(), out[0]["l"].data_ptr(), out[0]["n"].data_ptr())
            b.step_dev_f32(p32.data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr(), out[1]["a"].data_ptr(), out[1]["l"].data_ptr(), out[1]["n"].data_ptr())
            st.synchronize()
            for k in ("a", "l", "n"):
                assert torch.equal(out[0][k], out[1][k]), (f, k)
            na, nb = a.num_tracks(), b.num_tracks()
            assert np.array_equal(na, nb), f
            assert a.tracks(cap=max(int(na.max()), 1)).tobytes() == b.tracks(cap=max(int(na.max()), 1)).tobytes(), f
    a.check(); b.check()
    fa, oa = a.features_host(); fb, ob_ = b.features_host()
    assert np.array_equal(oa, ob_) and np.array_equal(fa, fb)
    ocfg = co.default_config(**kw)
    ntr = b.num_tracks(); trk = b.tracks(cap=max(int(ntr.max()), 1))
    for s in range(S):
        sc = co.OracleScene(ocfg, N)
        for f in range(F):
            if cnt[f, s] != 0:   # (0: the frame never reaches track(); -1: track() on an empty cloud)
                sc.track(pts[f, s, : max(int(cnt[f, s]), 0)].astype(np.float64), float(dts[f, s]))
        assert ntr[s] == sc.n_tracks, s
        assert_tracks_match(trk[s, : ntr[s]], sc.tracks(), ctx=f"scene {s}", exact=True)
    # mmw_normalize_f32 against mmw_normalize on the promoted rows
    rng = np.random.default_rng(12)
    raw = np.zeros((S, N, 5), np.float32)
    raw[..., 0] = rng.uniform(-4, 4, (S, N)); raw[..., 1] = rng.uniform(-0.5, 8, (S, N)); raw[..., 2] = rng.uniform(-2.5, 1.5, (S, N))
    raw[..., 3] = rng.normal(0, 0.5, (S, N)); raw[..., 4] = rng.gamma(1.0, 30.0, (S, N))
    raw[0, 0, :3] = 0.0                                       # r == 0 (Utils.py:387-390)
    nraw = rng.integers(0, N + 1, S).astype(np.int32)
    with torch.cuda.stream(st):
        r32 = torch.from_numpy(raw).to(dev); r64 = r32.double(); dn = torch.from_numpy(nraw).to(dev)
        o = [torch.zeros((S, N, 8), dtype=torch.float64, device=dev) for _ in range(2)]
        no = [torch.zeros((S,), dtype=torch.int32, device=dev) for _ in range(2)]
        a.normalize_dev(r64.data_ptr(), dn.data_ptr(), o[0].data_ptr(), no[0].data_ptr())
        a.normalize_dev(r32.data_ptr(), dn.data_ptr(), o[1].data_ptr(), no[1].data_ptr(), f32=True)
        st.synchronize()
    assert torch.equal(no[0], no[1])
    h0, h1, hn = o[0].cpu().numpy(), o[1].cpu().numpy(), no[0].cpu().numpy()
    cfg = co.default_config()
    for s in range(S):
        assert np.array_equal(h0[s, : hn[s]], h1[s, : hn[s]]), s
        assert np.array_equal(h1[s, : hn[s]], co.normalize(cfg, raw[s, : nraw[s]].astype(np.float64))), s
    assert 0 < int(hn.sum()) < int(nraw.sum())                  # the scene filter kept some rows and dropped some
    a.close(); b.close()


def test_dbscan_golden():
    z = np.load(os.path.join(GOLDEN, "dbscan.npz"))
    sizes = [int(v) for v in z["sizes"]]
    mx = max(sizes)
    sb = _mk(len(sizes), 512)  # ring*max_pts = 1536 >= every size
    pts = np.zeros((len(sizes), mx, 8))
    n = np.array(sizes, np.int32)
    for i, sz in enumerate(sizes):
        pts[i, :sz] = z[f"pts_{sz}"]
    for ms in (35, 8):
        labels, ncl = sb.dbscan_host(pts, n, min_samples=ms)
        for i, sz in enumerate(sizes):
            want = z[f"labels_{sz}_{ms}"]
            assert np.array_equal(labels[i, :sz], want), f"n={sz} min_samples={ms}"
            assert ncl[i] == want.max() + 1
    sb.close()


def test_dbscan_of_at_most_13_points_golden():
    """tests/golden/dbscan_small.npz (sklearn through the reference's metric: 1 .. 13 points, DB_MIN_SAMPLES_MIN 1 .. 10, clouds on
    which the BallTree's take-all rule and the brute force sklearn really runs below 12 points DISAGREE -- `tree_differs`) through
    mmw_dbscan, and the round-5 review's 3-point cloud with its own constants."""
    import json
    z = np.load(os.path.join(GOLDEN, "dbscan_small.npz"))
    sizes, mss = [int(v) for v in z["sizes"]], [int(v) for v in z["min_samples"]]
    per = len(z["pts_1"])
    S = len(sizes) * per
    sb = _mk(S, 16, fb_frames_batch=0, db_eps=float(z["db_eps"]), db_z_weight=float(z["db_z_weight"]), db_range_weight=float(z["db_range_weight"]))
    pts = np.zeros((S, 13, 8))
    n = np.zeros(S, np.int32)
    for a, sz in enumerate(sizes):
        pts[a * per: (a + 1) * per, :sz] = z[f"pts_{sz}"]
        n[a * per: (a + 1) * per] = sz
    for ms in mss:
        labels, ncl = sb.dbscan_host(pts, n, min_samples=ms)
        for a, sz in enumerate(sizes):
            want = z[f"labels_{sz}_{ms}"]
            got = labels[a * per: (a + 1) * per, :sz]
            assert np.array_equal(got, want), (sz, ms, np.flatnonzero((got != want).any(axis=1)))
            assert np.array_equal(ncl[a * per: (a + 1) * per], want.max(axis=1) + 1)
    sb.close()
    kw = json.loads(str(z["named_cfg"]))
    ms = kw.pop("db_min_samples")
    sb = _mk(1, 16, fb_frames_batch=0, **kw)
    labels, ncl = sb.dbscan_host(z["named_pts"][None].astype(np.float64), np.array([3], np.int32), min_samples=ms)
    assert list(labels[0, :3]) == [-1, -1, -1] and ncl[0] == 0
    sb.close()


@pytest.mark.parametrize("ms", [1, 2, 3, 4, 5, 6, 8, 10])
def test_dbscan_of_at_most_13_points_through_the_step(ms):
    """The same clouds as FRAMES: a scene per cloud, ring of one frame, no tracks yet -- TrackBuffer.track hands apply_DBscan
    exactly the cloud (Tracking.py:689-697) --, in whichever kernels the layout runs the start-up and the small-cloud DBSCAN in
    (k_dbscan_startup / k_post's workers / k_chain / k_scene's hand-over).  Frame 0 against the golden labels; two more
    frames (the clouds shifted by 13 scenes: new tracks gate, the rest clusters again) against the oracle."""
    from oracle import c_oracle as co
    z = np.load(os.path.join(GOLDEN, "dbscan_small.npz"))
    sizes = [int(v) for v in z["sizes"]]
    per = min(len(z["pts_1"]), 512 // len(sizes))    # (<= 512 scenes: the one-workgroup step takes no more)
    S = len(sizes) * per
    kw = dict(fb_frames_batch=0, db_min_samples=ms, tr_max_tracks=12, db_eps=float(z["db_eps"]), db_z_weight=float(z["db_z_weight"]),
              db_range_weight=float(z["db_range_weight"]))
    sb = _mk(S, 16, **kw)
    cfg = co.default_config(**kw)
    scenes = [co.OracleScene(cfg, 16) for _ in range(S)]
    pts = np.zeros((S, 16, 8))
    n = np.zeros(S, np.int32)
    for a, sz in enumerate(sizes):
        pts[a * per: (a + 1) * per, :sz] = z[f"pts_{sz}"][:per]
        n[a * per: (a + 1) * per] = sz
    dts = np.full(S, 0.1)
    n_small = 0
    for f in range(3):
        P, C = np.roll(pts, 13 * f, axis=0), np.roll(n, 13 * f)
        assoc, labels, dbn = sb.step_host(P, C, dts)
        ntr = sb.num_tracks()
        trk = sb.tracks(cap=16)
        for s in range(S):
            oa, ol = scenes[s].track(P[s, : C[s]], 0.1)
            assert np.array_equal(assoc[s, : C[s]], oa), (f, s)
            assert (ol is None) == (dbn[s] < 0), (f, s)
            if ol is not None:
                assert dbn[s] == len(ol) and np.array_equal(labels[s, : dbn[s]], ol), (f, s, ms, len(ol))
                n_small += int(len(ol) // 2 <= 5)
            if f == 0:
                sz = sizes[s // per]
                assert np.array_equal(labels[s, :sz], z[f"labels_{sz}_{ms}"][s % per]), (s, sz, ms)
            assert ntr[s] == scenes[s].n_tracks, (f, s)
            assert_tracks_match(trk[s, : ntr[s]], scenes[s].tracks(), ctx=f"f{f} s{s}", exact=True)
    sb.check()
    sb.close()
    assert n_small > S


def test_dbscan_pairs_at_the_threshold_vs_oracle():
    """query_radius decides most leaf tests from an fp32 value of the metric and only the pairs within its error bound of eps in
    fp64 (k_dbscan.hip, leaf_screen).  Clouds built so that MANY pairs sit within 1e-7 .. 1e-3 of eps (lattices at the
    threshold spacing, jittered), the same far from the origin (a wider bound), and at coordinates where fp32 resolves nothing
    (1e9: the screen must switch itself off): labels equal the fp64 oracle's (Utils.py:250-291 through sklearn's BallTree)."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(5)
    cfg = co.default_config()
    eps, rw = cfg.db_eps, cfg.db_range_weight
    clouds = []
    for off, jit in ((0.0, 1e-7), (0.0, 1e-5), (40.0, 1e-6), (3000.0, 1e-4), (1e9, 1e-3), (0.0, 0.0)):
        y0 = 2.0
        h = float(np.sqrt(eps / (1.0 - y0 * rw)))          # metric(p, p + (h, 0, 0)) == eps up to rounding
        rows = []
        for r in range(6):   # chains along x, 10 m apart in z: with min_samples = 2 a chain breaks wherever a step is "out"
            k = 60
            pts = np.zeros((k, 8))
            pts[:, 0] = off + np.cumsum(h * (1.0 + jit * rng.standard_normal(k)))
            pts[:, 1] = y0
            pts[:, 2] = 10.0 * r
            rows.append(pts)
        c = np.concatenate(rows)
        clouds.append(c[rng.permutation(len(c))])
    # clusters a hundred times the usual size: every difference is large, the far-pair argument of the bound is what holds
    big = np.zeros((300, 8))
    big[:, 0] = 100.0 * rng.normal(0, 0.4, 300) + np.repeat([0.0, 500.0, -800.0], 100)
    big[:, 1] = np.abs(rng.normal(3.0, 1.0, 300))
    big[:, 2] = rng.uniform(0, 2, 300)
    clouds.append(big)
    mx = max(len(c) for c in clouds)
    sb = _mk(len(clouds), 512)
    pts = np.zeros((len(clouds), mx, 8))
    n = np.array([len(c) for c in clouds], np.int32)
    for i, c in enumerate(clouds):
        pts[i, : len(c)] = c
    seen = set()
    for ms in (2, 3, 5):
        labels, ncl = sb.dbscan_host(pts, n, min_samples=ms)
        for i, c in enumerate(clouds):
            want = co.dbscan(cfg, c, min_samples=ms)
            assert np.array_equal(labels[i, : len(c)], want), (i, ms)
            seen.add(int(want.max()))
    sb.close()
    assert len(seen) > 2   # the chains fall apart differently with min_samples: the threshold pairs matter


def test_full_size_config_against_oracle():
    """BASELINE.json configs[2] shape (512 pts, TR_MAX_TRACKS 8, mixed target counts) on 384 scenes:
    final track state after 10 frames bit-equal to the oracle for EVERY scene, plus the
    size-independent invariants (labels in range, association indices valid, ring lengths)."""
    from mmwave_msc_amd.synth import make_batch
    from oracle import c_oracle as co
    import bench
    S, N, T, F = 384, 512, 8, 10
    pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=1)
    sb = _mk(S, N, tr_max_tracks=T)
    ob = co.OracleBatch(co.default_config(tr_max_tracks=T), S, N)
    co.batch_run_f32(ob, pts, cnt, dts, 0)
    for f in range(F):
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
        ntr_before = sb.num_tracks()
        for s in range(0, S, 37):
            a = assoc[s, : cnt[f, s]]
            assert a.min() >= -1
            if dbn[s] >= 0:
                lab = labels[s, : dbn[s]]
                assert lab.min() >= -1 and (lab.max() < 0 or set(range(lab.max() + 1)) <= set(lab.tolist()))
    ntr = sb.num_tracks()
    trk = sb.tracks(cap=int(ntr.max()))
    ln, rn = sb.batch_ring()
    for s in range(S):
        want = ob.scenes[s].tracks()
        assert ntr[s] == len(want), s
        assert_tracks_match(trk[s, : ntr[s]], want, ctx=f"scene {s}", exact=True)
        assert np.array_equal(rn[s, : ln[s]], ob.scenes[s].batch_ring())
    sb.close()


def test_error_paths_are_loud():
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.synth import make_batch
    # capacity: a track list that cannot hold the clusters of frame 0
    sb = SceneBatch(_lib.default_config(track_cap=1), 1, 256)
    p, c, d = make_batch([3], 1, 256, 3)
    with pytest.raises(_lib.MmwError) as ei:
        sb.step_host(p[0].astype(np.float64), c[0], d[0])
    assert ei.value.code == _lib.E_CAPACITY
    sb.close()
    # a point count the context was not sized for is an error, not a silently skipped frame
    sb = SceneBatch(_lib.default_config(), 2, 64)
    pts = np.zeros((2, 64, 8)); pts[..., 1] = 1.0; pts[..., 2] = 1.0
    with pytest.raises(_lib.MmwError) as ei:
        sb.step_host(pts, np.array([10, 65], np.int32), np.array([0.1, 0.1]))
    assert ei.value.code == _lib.E_ARG
    sb.close()
    # misaligned device pointer
    sb = SceneBatch(_lib.default_config(), 1, 64)
    buf = sb.alloc(64 * 64 + 64)
    nb = sb.alloc(4).upload(np.array([8], np.int32))
    db = sb.alloc(8).upload(np.array([0.1]))
    with pytest.raises(_lib.MmwError) as ei:
        sb.step_dev(buf.ptr + 8, nb.ptr, db.ptr)
    assert ei.value.code == _lib.E_ARG
    # what sklearn's DBSCAN refuses on every call (InvalidParameterError: eps <= 0 or NaN, min_samples < 1) is refused at creation
    labels_of = lambda **k: sb.dbscan_host(np.zeros((1, 8, 8)), np.array([4], np.int32), **k)   # noqa: E731
    for bad in (dict(eps=0.0), dict(eps=-0.3), dict(eps=float("nan")), dict(min_samples=0)):
        with pytest.raises(_lib.MmwError) as ei:
            labels_of(**bad)
        assert ei.value.code == _lib.E_ARG, bad
    assert list(labels_of(min_samples=1)[0][0, :4]) == [0, 0, 0, 0]     # (min_samples = 1: every point a core point, four equal rows one cluster)
    sb.close()
    for bad in (dict(db_eps=0.0), dict(db_min_samples=0), dict(db_eps=float("nan")), dict(seek_inner=1, db_inner_eps=0.0)):
        with pytest.raises(_lib.MmwError) as ei:
            SceneBatch(_lib.default_config(**bad), 1, 64)
        assert ei.value.code == _lib.E_ARG, bad


def test_capacity_overflow_is_per_scene_and_recoverable():
    """A scene whose clusters do not fit its track list (Tracking.py:576-589 has no cap; here track_cap) gets the sticky capacity
    bit and MMW_E_CAPACITY names it; the OTHER scenes of the context are not touched -- they keep matching their oracles -- and
    mmw_reset_scenes gives the one scene a fresh TrackBuffer / BatchedData from which it matches a fresh oracle again."""
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.synth import make_batch
    from oracle import c_oracle as co
    S, N, F = 6, 256, 12
    kw = dict(tr_max_tracks=4, track_cap=2)
    sb = _mk(S, N, **kw)
    cfg = co.default_config(**kw)
    pts = np.zeros((F, S, N, 8), np.float32); cnt = np.zeros((F, S), np.int32); dts = np.zeros((F, S))
    targets = [1, 2, 3, 1, 2, 1]                      # scene 2 holds three targets: one more than track_cap
    for s in range(S):
        p, c, d = make_batch([900 + s], F, N, targets[s])
        pts[:, s], cnt[:, s], dts[:, s] = p[:, 0], c[:, 0], d[:, 0]
    orc = [co.OracleScene(cfg, N) for _ in range(S)]
    good = [s for s in range(S) if s != 2]
    raised = False
    for f in range(6):
        try:
            sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
        except _lib.MmwError as e:
            assert e.code == _lib.E_CAPACITY and "scene 2" in str(e)
            raised = True
        for s in good:
            orc[s].track(pts[f, s, : cnt[f, s]].astype(np.float64), dts[f, s])
    assert raised
    err = sb.errors()
    assert err[2] & 4 and not err[good].any()
    ntr, trk = sb.num_tracks(), sb.tracks(cap=4)
    for s in good:
        want = orc[s].tracks()
        assert ntr[s] == len(want)
        for name in ("x", "P", "centroid", "lifetime", "point_num", "ring_n"):
            assert np.array_equal(trk[s, : ntr[s]][name], want[name]), (s, name)
    # recover scene 2 alone: from here on it is a fresh scene fed ONE target (which fits)
    mask = np.zeros(S, bool); mask[2] = True
    sb.reset_scenes(mask)
    assert not sb.errors().any()
    sb.check()
    orc[2] = co.OracleScene(cfg, N)
    p2, c2, d2 = make_batch([990], F, N, 1)
    pts[6:, 2], cnt[6:, 2], dts[6:, 2] = p2[6:, 0], c2[6:, 0], d2[6:, 0]
    for f in range(6, F):
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
        for s in range(S):
            oa, ol = orc[s].track(pts[f, s, : cnt[f, s]].astype(np.float64), dts[f, s])
            assert np.array_equal(assoc[s, : cnt[f, s]], oa), (f, s)
    ntr, trk = sb.num_tracks(), sb.tracks(cap=4)
    for s in range(S):
        want = orc[s].tracks()
        assert ntr[s] == len(want), s
        for name in ("x", "P", "centroid", "lifetime", "point_num", "ring_n"):
            assert np.array_equal(trk[s, : ntr[s]][name], want[name]), (s, name)
    sb.close()


def test_seek_inner_many_scenes_vs_oracle():
    """ClusterTrack.seek_inner_clusters (Tracking.py:409-448, call site Tracking.py:656 active) over a batch: pairs of
    people one outer cluster wide split into two tracks, standing pairs take the FB_FRAMES_BATCH_STATIC ring size, scenes
    whose track list the inner tracks fill skip the frame's own apply_DBscan.  Ints, fp64 state, rings, inner labels
    and features bit-equal to the oracle, which tests/test_oracle_golden.py pins on recordings of the reference."""
    from mmwave_msc_amd.synth import make_pair_scene, make_scene
    from oracle import c_oracle as co
    S, N, F = 12, 256, 10
    kw = dict(seek_inner=1, tr_max_tracks=3, fb_frames_batch_static=3)
    ps, cs, ds = [], [], []
    for s in range(S):
        if s % 4 == 3:
            p, c, d = make_scene(4000 + s, F, N, 2)                       # no pairs: calls that find one cluster
        else:
            p, c, d = make_pair_scene(4000 + s, F, N, 1 + s % 2, sep=0.7 + 0.1 * (s % 3), static=(s % 4 == 1))
        ps.append(p); cs.append(c); ds.append(d)
    pts, cnt, dts = np.stack(ps, 1), np.stack(cs, 1), np.stack(ds, 1)
    sb = _mk(S, N, **kw)
    cfg = co.default_config(**kw)
    scenes = [co.OracleScene(cfg, N) for _ in range(S)]
    split = 0
    for f in range(F):
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
        calls = sb.inner_calls()
        ntr = sb.num_tracks()
        trk = sb.tracks(cap=max(int(ntr.max()), 1))
        for s in range(S):
            oa, ol = scenes[s].track(pts[f, s].astype(np.float64), float(dts[f, s]))
            assert np.array_equal(assoc[s], oa), (f, s)
            assert (ol is None) == (dbn[s] < 0), (f, s, dbn[s])
            if ol is not None:
                assert np.array_equal(labels[s, : dbn[s]], ol), (f, s)
            oc = scenes[s].inner_calls()
            assert len(calls[s]) == len(oc), (f, s)
            for a, (_, b) in zip(calls[s], oc):
                assert np.array_equal(a, b), (f, s)
                split += int(b.max() >= 1)
            assert ntr[s] == scenes[s].n_tracks, (f, s)
            assert_tracks_match(trk[s, : ntr[s]], scenes[s].tracks(), ctx=f"f{f} s{s}", exact=True)
        feat, owner = sb.features_host()
        ofeat = [scenes[s].features()[0] for s in range(S)]
        ofeat = np.concatenate([x for x in ofeat if len(x)], axis=0)
        assert np.array_equal(feat, ofeat), f
    assert split >= 4, "no inner split happened: the scenario does not exercise the spawn"
    sb.check()
    sb.close()


def test_side_workers_are_probed_and_switchable():
    """The DBSCAN chain workers run on a side stream only after the first step has checked that this stream does not
    share a hardware queue with the context's stream (mmw_side_workers: 2 = unchecked, 1 = in use, 0 = off); large and
    small clouds give the oracle's labels with the workers on and off."""
    from mmwave_msc_amd.synth import make_batch
    from oracle import c_oracle as co
    S, N, F = 24, 256, 8
    pts = np.zeros((F, S, N, 8), np.float32)
    cnt = np.zeros((F, S), np.int32)
    dts = np.zeros((F, S))
    for s in range(S):   # every third scene is clutter only: it clusters its whole ring (up to 768 points) every frame
        p, c, d = make_batch([5200 + s], F, N, s % 3)
        pts[:, s], cnt[:, s], dts[:, s] = p[:, 0], c[:, 0], d[:, 0]
    results = []
    for on in (1, -1):
        sb = _mk(S, N, tr_max_tracks=2, chain_side_stream=on)
        assert sb.side_workers() == (2 if on == 1 else 0)
        out = []
        for f in range(F):
            if on == 1 and f == 4:
                sb.set_chain_side_stream(False)
                assert sb.side_workers() == 0
            if on == 1 and f == 6:
                sb.set_chain_side_stream(True)
                assert sb.side_workers() == 2
            assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
            if on == 1 and f == 0:
                assert sb.side_workers() in (0, 1)   # checked by the first step
            out.append((assoc.copy(), labels.copy(), dbn.copy()))
        sb.check()
        results.append((out, sb.tracks(cap=8).copy(), sb.num_tracks().copy()))
        sb.close()
    for (a0, l0, d0), (a1, l1, d1) in zip(results[0][0], results[1][0]):
        assert np.array_equal(a0, a1) and np.array_equal(d0, d1)
        for s in range(S):
            assert np.array_equal(l0[s, : max(d0[s], 0)], l1[s, : max(d1[s], 0)]), s
    assert np.array_equal(results[0][2], results[1][2])
    for s in range(S):
        k = int(results[0][2][s])
        for name in ("x", "P", "centroid", "min_vals", "max_vals", "spread_est", "group_disp_est", "n_est", "lifetime",
                     "point_num", "is_static", "ring_len", "ring_n", "uid"):
            assert np.array_equal(results[0][1][s, :k][name], results[1][1][s, :k][name]), (s, name)
    # ... and both equal the oracle (large clouds occur: frame 0 clusters 256 points, frames 1-2 up to 768)
    ob = co.OracleBatch(co.default_config(tr_max_tracks=2), S, N)
    saw_big = False
    for f in range(F):
        oa, ol, od = ob.step(pts[f].astype(np.float64), cnt[f], dts[f])
        a, l, d = results[0][0][f]
        assert np.array_equal(d, od) and np.array_equal(a, oa)
        for s in range(S):
            if od[s] > 0:
                assert np.array_equal(l[s, : od[s]], ol[s, : od[s]]), (f, s)
                saw_big |= od[s] > 256
    assert saw_big


def test_mid_size_context_takes_the_four_launch_step():
    """Contexts of <= 768 scenes run a two-launch step (_predict_all at the head of k_track, the large clouds in k_post);
    everything above runs k_predict / k_track / k_post / k_dbscan_big.  832 scenes -- the smallest kind of context that takes
    the second form, in whichever Kalman layout the fixture selects -- against the oracle, start-up frames (large clouds)
    included."""
    from oracle import c_oracle as co
    import bench
    S, N, T, F = 832, 128, 3, 7
    pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=4)
    sb = _mk(S, N, tr_max_tracks=T)
    ob = co.OracleBatch(co.default_config(tr_max_tracks=T), S, N)
    for f in range(F):
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
        oa, ol, od = ob.step(pts[f].astype(np.float64), cnt[f], dts[f])
        assert np.array_equal(dbn, od), f
        assert np.array_equal(assoc, oa), f
        for s in range(S):
            if od[s] > 0:
                assert np.array_equal(labels[s, : od[s]], ol[s, : od[s]]), (f, s)
    ntr = sb.num_tracks()
    trk = sb.tracks(cap=max(int(ntr.max()), 1))
    for s in range(S):
        want = ob.scenes[s].tracks()
        assert ntr[s] == len(want), s
        for name in ("x", "P", "centroid", "spread_est", "group_disp_est", "n_est", "lifetime", "point_num", "ring_n"):
            assert np.array_equal(trk[s, : ntr[s]][name], want[name]), (s, name)
    sb.check()
    sb.close()


def test_largest_capacity_context_vs_oracle():
    """ring * max_pts = 1920, the largest cloud the BallTree emulation holds (63 nodes): the LDS carve-up of that capacity
    keeps 256-point bit rows only, k_track runs its four-points-per-thread build; start-up frames cluster 640, 1280 and
    1920 points."""
    from oracle import c_oracle as co
    import bench
    S, N, T, F = 24, 640, 4, 6
    pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=4)
    for s in range(0, S, 3):   # every third scene clutter only: it keeps clustering its whole ring
        rng = np.random.default_rng(9000 + s)
        pts[:, s, :, 0] = rng.uniform(-6, 6, size=(F, N)).astype(np.float32)
        pts[:, s, :, 1] = rng.uniform(0.3, 7.5, size=(F, N)).astype(np.float32)
        pts[:, s, :, 2] = rng.uniform(0.05, 2.4, size=(F, N)).astype(np.float32)
    sb = _mk(S, N, tr_max_tracks=T)
    assert sb.UM == 1920
    ob = co.OracleBatch(co.default_config(tr_max_tracks=T), S, N)
    biggest = 0
    for f in range(F):
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
        oa, ol, od = ob.step(pts[f].astype(np.float64), cnt[f], dts[f])
        assert np.array_equal(dbn, od), f
        assert np.array_equal(assoc, oa), f
        for s in range(S):
            if od[s] > 0:
                assert np.array_equal(labels[s, : od[s]], ol[s, : od[s]]), (f, s)
        biggest = max(biggest, int(od.max()))
    assert biggest > 1536
    ntr = sb.num_tracks()
    trk = sb.tracks(cap=max(int(ntr.max()), 1))
    for s in range(S):
        want = ob.scenes[s].tracks()
        assert ntr[s] == len(want), s
        for name in ("x", "P", "centroid", "spread_est", "n_est", "lifetime", "point_num", "ring_n"):
            assert np.array_equal(trk[s, : ntr[s]][name], want[name]), (s, name)
    sb.check()
    sb.close()


def test_dbscan_of_more_than_1920_points_golden():
    """apply_DBscan has no size limit (Utils.py:250-291); a context's largest cloud is its ring, MMW_RING_MAX x
    MMW_MAX_PTS_LIMIT = 4096 points.  Above 1920 points (64 .. 128 BallTree leaves) the carve-up lives in global memory
    (k_dbscan_only_huge / k_dbscan_huge, four mask words per position): sklearn's labels on the reference's metric for
    1921 .. 4096 points, and the smaller golden clouds through the same context (they take the LDS kernel)."""
    z = np.load(os.path.join(GOLDEN, "dbscan_huge.npz"))
    sizes = [int(v) for v in z["sizes"]]
    mx = max(sizes)
    sb = _mk(len(sizes), 1024, fb_frames_batch=3)
    assert sb.UM == 4096
    pts = np.zeros((len(sizes), mx, 8))
    n = np.array(sizes, np.int32)
    for i, sz in enumerate(sizes):
        pts[i, :sz] = z[f"pts_{sz}"]
    for ms in (35, 8):
        labels, ncl = sb.dbscan_host(pts, n, min_samples=ms)
        for i, sz in enumerate(sizes):
            want = z[f"labels_{sz}_{ms}"]
            assert np.array_equal(labels[i, :sz], want), f"n={sz} min_samples={ms}"
            assert ncl[i] == want.max() + 1
    zs = np.load(os.path.join(GOLDEN, "dbscan.npz"))
    small = [int(v) for v in zs["sizes"]][-len(sizes):]
    pts2 = np.zeros((len(sizes), max(small), 8))
    for i, sz in enumerate(small):
        pts2[i, :sz] = zs[f"pts_{sz}"]
    labels, ncl = sb.dbscan_host(pts2, np.array(small, np.int32), min_samples=35)
    for i, sz in enumerate(small):
        assert np.array_equal(labels[i, :sz], zs[f"labels_{sz}_35"]), sz
    sb.check()
    sb.close()


def test_rings_of_more_than_1920_points_vs_oracle():
    """ring * max_pts = 4 x 960: scenes of clutter keep clustering their whole ring -- 960, 1920, 2880, 3840 points: the last
    two on k_dbscan_huge (work list 2) --, and scenes whose targets return 10 points a frame get their clusters only once
    the ring holds four frames of them (min_samples 35), i.e. from a cloud of > 1920 points: labels, spawned tracks and every
    later frame equal the oracle."""
    from oracle import c_oracle as co
    from mmwave_msc_amd.synth import make_batch
    S, N, T, F = 9, 960, 4, 8
    rng = np.random.default_rng(77)
    pts = np.zeros((F, S, N, 8), np.float32)
    pts[..., 0] = rng.uniform(-6, 6, size=(F, S, N))
    pts[..., 1] = rng.uniform(0.3, 7.5, size=(F, S, N))
    pts[..., 2] = rng.uniform(0.05, 2.4, size=(F, S, N))
    pts[..., 3:6] = rng.normal(0, 0.05, size=(F, S, N, 3))
    pts[..., 6] = rng.normal(0, 0.3, size=(F, S, N))
    pts[..., 7] = rng.gamma(1.0, 30.0, size=(F, S, N))
    for s in range(S):
        if s % 3 == 0:
            continue   # clutter only
        for k in range(1 + s % 2):   # one or two faint targets: 10 points a frame each
            c = np.array([-2.0 + 3.0 * k + 0.3 * s, 2.0 + 0.5 * s])
            for f in range(F):
                idx = rng.choice(N, size=10, replace=False)
                pts[f, s, idx, 0:2] = (c + rng.normal(0, 0.06, size=(10, 2))).astype(np.float32)
                pts[f, s, idx, 2] = rng.uniform(0.6, 1.4, size=10).astype(np.float32)
    cnt = np.full((F, S), N, np.int32)
    dts = np.full((F, S), 0.1)
    kw = dict(tr_max_tracks=T, fb_frames_batch=3)
    sb = _mk(S, N, **kw)
    assert sb.UM == 3840
    ob = co.OracleBatch(co.default_config(**kw), S, N)
    biggest = clustered_big = 0
    for f in range(F):
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
        oa, ol, od = ob.step(pts[f].astype(np.float64), cnt[f], dts[f])
        assert np.array_equal(dbn, od), f
        assert np.array_equal(assoc, oa), f
        for s in range(S):
            if od[s] > 0:
                assert np.array_equal(labels[s, : od[s]], ol[s, : od[s]]), (f, s)
                if od[s] > 1920 and ol[s, : od[s]].max() >= 0:
                    clustered_big += 1
        biggest = max(biggest, int(od.max()))
    assert biggest > 3000 and clustered_big > 0, (biggest, clustered_big)
    ntr = sb.num_tracks()
    trk = sb.tracks(cap=max(int(ntr.max()), 1))
    for s in range(S):
        want = ob.scenes[s].tracks()
        assert ntr[s] == len(want), s
        for name in ("x", "P", "centroid", "spread_est", "n_est", "lifetime", "point_num", "ring_n"):
            assert np.array_equal(trk[s, : ntr[s]][name], want[name]), (s, name)
    sb.check()
    sb.close()


def test_two_side_worker_contexts_in_one_process():
    """Two contexts with DBSCAN chain workers (each its own side stream) stepped alternately on their own streams: streams are
    multiplexed onto a few hardware queues, a polling worker must never keep a context's own kernels waiting for its
    bounded wait (the probe of the first step turns the workers off where the queues are shared; idle workers leave after
    a few ms).  Results equal the oracle, and no step stalls."""
    import time
    import torch
    from oracle import c_oracle as co
    import bench
    S, N, T, F = 1536, 64, 3, 10
    ctxs = []
    for c in range(2):
        pts, cnt, dts = bench.generate(np.arange(c * S, (c + 1) * S), F, N, T, workers=4)
        sb = _mk(S, N, tr_max_tracks=T, chain_side_stream=1)
        st = torch.cuda.Stream()
        sb.follow_torch_stream(st)
        dev = torch.device("cuda:0")
        bufs = dict(p=[torch.from_numpy(pts[f]).to(dev).double() for f in range(F)], n=torch.from_numpy(cnt).to(dev),
                    d=torch.from_numpy(dts).to(dev), a=torch.empty((S, N), dtype=torch.int32, device=dev),
                    l=torch.empty((S, sb.UM), dtype=torch.int32, device=dev), b=torch.empty((S,), dtype=torch.int32, device=dev))
        ctxs.append((sb, st, bufs, (pts, cnt, dts)))
    torch.cuda.synchronize()
    times = []
    for f in range(F):
        t0 = time.perf_counter()
        for sb, st, b, _ in ctxs:
            sb.step_dev(b["p"][f].data_ptr(), b["n"][f].data_ptr(), b["d"][f].data_ptr(), b["a"].data_ptr(), b["l"].data_ptr(), b["b"].data_ptr())
        torch.cuda.synchronize()
        if f >= 3:   # (the first frames hold the start-up DBSCAN and the stream probe)
            times.append(time.perf_counter() - t0)
    # a worker that holds a stream back costs its bounded wait: 0.2 s and more (k_dbscan.hip: kMustWaitTicks) -- or, for idle
    # workers on a crosswise-shared queue, a few ms per step.  Wall-clock on a shared box: the typical step is judged at 20 ms,
    # a single one at 150 ms (one hiccup of the host must not fail the suite); give-ups are checked below (mmw_check)
    times.sort()
    assert times[len(times) // 2] < 0.02 and times[-1] < 0.15, f"steps of two contexts took {[round(t * 1e3, 1) for t in times]} ms: a polling worker held a stream back"
    for sb, st, b, (pts, cnt, dts) in ctxs:
        assert sb.side_workers() in (0, 1)
        sb.check()
        assert int(sb.diag_queue()[4]) == 0   # no bounded wait was given up
        ob = co.OracleBatch(co.default_config(tr_max_tracks=T), S, N)
        co.batch_run_f32(ob, pts, cnt, dts, 0)
        ntr = sb.num_tracks()
        trk = sb.tracks(cap=max(int(ntr.max()), 1))
        for s in range(0, S, 7):
            want = ob.scenes[s].tracks()
            assert ntr[s] == len(want), s
            for name in ("x", "P", "centroid", "lifetime", "point_num", "ring_n"):
                assert np.array_equal(trk[s, : ntr[s]][name], want[name]), (s, name)
        sb.close()
