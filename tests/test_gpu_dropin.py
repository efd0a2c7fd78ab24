"""GPU tests of the reference-shaped host surface: TrackBuffer/BatchedData views, the
headless offline loop, the Utils helpers, and the MARS CNN (keypoints vs the fp64
numpy oracle within 1e-4 m, the tolerance stated in SURVEY.md §8c)."""
import os

import numpy as np
import pytest

from tests._golden import assert_tracks_match, GOLDEN, assert_feat_equal, close64, load_scenario

pytestmark = pytest.mark.gpu
KP_TOL = 1e-4


def test_trackbuffer_dropin_matches_golden():
    from mmwave_msc_amd.tracking import BatchedData, TrackBuffer
    g = load_scenario("n200_k2")
    tb, batch = TrackBuffer(max_pts=256), BatchedData()
    assert tb.effective_tracks == [] and len(batch.effective_data) == 0
    for f in range(16):
        c = int(g["cnt"][f])
        tb.dt = float(g["dt"][f])
        tb.track(g["pts"][f, :c].astype(np.float64), batch)
        assert np.array_equal(tb.last_assoc, g["assoc"][f, :c])
        nt = int(g["n_tracks"][f])
        tracks = tb.effective_tracks
        assert len(tracks) == nt
        for j, t in enumerate(tracks):
            w = g["tracks"][f, j]
            assert t.state.x.shape == (9, 1) and close64(t.state.x[:, 0], w["x"]) and close64(t.state.P, w["P"])
            assert np.array_equal(t.cluster.centroid, w["centroid"]) and t.cluster.point_num == w["point_num"]
            assert t.cluster.status == bool(w["is_static"]) and t.lifetime == w["lifetime"]
            assert len(t.batch.buffer) == w["ring_len"]
            assert t.batch.effective_data.shape == (int(w["ring_n"].sum()), 8)
            assert t.color.shape == (3,)
        assert len(batch.effective_data) == int(g["ring_n"][f].sum())
        assert [len(fr) for fr in batch.buffer] == list(g["ring_n"][f, : g["ring_len"][f]])
    # full-fidelity rings in the single-scene face: first spawned track keeps every cluster row
    assert tb.effective_tracks[0].batch.effective_data.shape[0] > 64
    tb.close()


def test_offline_loop_matches_reference_run(tmp_path):
    from mmwave_msc_amd.offline_main import offline_main
    z = np.load(os.path.join(GOLDEN, "offline.npz"))
    (tmp_path / "1.csv").write_text(str(z["csv1"]))
    (tmp_path / "2.csv").write_text(str(z["csv2"]))
    ntr, dts = [], []
    tb = offline_main(str(tmp_path), on_frame=lambda t, det, k: (ntr.append(len(t.effective_tracks)), dts.append(t.dt)), max_pts=64)
    assert np.array_equal(np.array(ntr), z["n_tracks"])
    assert np.allclose(np.array(dts), z["dt"], rtol=0, atol=0)
    tb.close()


def test_utils_helpers_against_golden_and_oracle():
    from mmwave_msc_amd import utils
    from oracle import c_oracle as co
    z = np.load(os.path.join(GOLDEN, "dbscan.npz"))
    for n in (61, 241, 700, 1536):
        pts = z[f"pts_{n}"].astype(np.float64)
        clusters = utils.apply_DBscan(pts)
        assert [len(c) for c in clusters] == list(z[f"sizes_{n}"])
        lab = utils.dbscan_labels(pts, min_samples=8)
        assert np.array_equal(lab, z[f"labels_{n}_8"])
        if clusters:
            first = np.nonzero(z[f"labels_{n}_35"] == 0)[0]
            assert np.array_equal(np.array(clusters[0]), pts[first])
    zn = np.load(os.path.join(GOLDEN, "normalize.npz"))
    raw = zn["raw"]
    det = {"x": list(raw[:, 0]), "y": list(raw[:, 1]), "z": list(raw[:, 2]), "doppler": list(raw[:, 3]), "peakVal": list(raw[:, 4])}
    out = utils.normalize_data(det)
    assert out.shape == zn["out"].shape and np.allclose(out, zn["out"], rtol=0, atol=1e-12)
    # format_single_frame(relative_coordinates(...)) vs the oracle's feature map of a live track
    g = load_scenario("n256_k4")
    sc = co.OracleScene(co.default_config(), 256)
    for f in range(6):
        sc.track(g["pts"][f, : g["cnt"][f]].astype(np.float64), float(g["dt"][f]))
    feat, owner = sc.features()
    rec = sc.tracks()[0]
    frames = [sc.track_ring_frame(0, k) for k in range(rec["ring_len"])]
    mine = utils.format_single_frame(utils.relative_coordinates(frames, rec["centroid"][:2]))
    assert mine.shape == (3, 8, 8, 5)
    assert_feat_equal(mine[None], feat[:1])


@pytest.mark.parametrize("frames", [3, 1])
def test_mars_cnn_keypoints_vs_fp64_oracle(frames):
    import torch
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    from oracle.mars_np import mars_forward_np
    w = random_keras_weights(seed=3, frames=frames)
    model = MarsCNN.from_keras_weights(w).to("cuda:0")
    rng = np.random.default_rng(1)
    shape = (257, 3, 8, 8, 5) if frames == 3 else (257, 8, 8, 5)
    x = rng.normal(0, 0.4, size=shape).astype(np.float32)
    x[:, ..., 40:, :] = 0 if frames == 1 else x[:, ..., 40:, :]
    with torch.no_grad():
        y = model(torch.from_numpy(x).to("cuda:0")).float().cpu().numpy()
    ref = mars_forward_np(w, x)
    assert y.shape == (257, 57)
    assert np.abs(y - ref).max() <= KP_TOL, np.abs(y - ref).max()
    assert np.abs(y - ref).max() <= KP_TOL * max(1.0, np.abs(ref).max())


def test_estimate_posture_end_to_end():
    """track -> features (GPU) -> CNN (GPU) -> track.keypoints, against oracle features + fp64 CNN."""
    import torch
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    from mmwave_msc_amd.tracking import BatchedData, TrackBuffer
    from oracle import c_oracle as co
    from oracle.mars_np import mars_forward_np
    g = load_scenario("n256_k4")
    w = random_keras_weights(seed=5, frames=3)
    model = MarsCNN.from_keras_weights(w).to("cuda:0")
    tb, batch = TrackBuffer(max_pts=256), BatchedData()
    sc = co.OracleScene(co.default_config(), 256)
    for f in range(8):
        c = int(g["cnt"][f])
        tb.dt = float(g["dt"][f])
        tb.track(g["pts"][f, :c].astype(np.float64), batch)
        tb.estimate_posture(model)
        sc.track(g["pts"][f, :c].astype(np.float64), float(g["dt"][f]))
        feat, owner = sc.features()
        ref = mars_forward_np(w, feat)
        tracks = tb.effective_tracks
        assert len(tracks) == len(owner)
        for i, j in enumerate(owner):
            assert np.abs(tracks[j].keypoints - ref[i]).max() <= KP_TOL
    tb.close()


def test_frame_host_one_round_trip_equals_the_separate_calls():
    """mmw_frame_host (normalize_data + track in one round trip; include/mmw.h) for a batch of scenes against mmw_normalize
    followed by mmw_step_host, and against the oracle: rows, counts, association, labels, track counts and state bit-equal;
    scenes whose rows are all filtered out are skipped (offline_main.py:56), errors come back with the same call."""
    import bench_ingest
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.synth import make_batch
    from oracle import c_oracle as co
    S, N, F, T = 12, 160, 10, 3
    pts = np.zeros((F, S, N, 8), np.float32); cnt = np.zeros((F, S), np.int32); dts = np.zeros((F, S))
    for s in range(S):
        p, c, d = make_batch([7300 + s], F, N, s % (T + 1), ragged=(s % 2 == 1))
        pts[:, s], cnt[:, s], dts[:, s] = p[:, 0], c[:, 0], d[:, 0]
    cfg = _lib.default_config(tr_max_tracks=T)
    raw = bench_ingest.raw_rows_from_normalised(pts, float(cfg.tilt_cos), float(cfg.tilt_sin), float(cfg.s_height)).astype(np.float64)
    raw[2, 4, :, 2] = 9.0                               # frame 2 of scene 4: every row above the scene filter -> the frame is skipped
    cnt[5, 7] = 0
    a, b = SceneBatch(cfg, S, N), SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
    ocfg = co.default_config(tr_max_tracks=T)
    scenes = [co.OracleScene(ocfg, N) for _ in range(S)]
    for f in range(F):
        r = a.frame_host(cnt[f], dts[f], raw=raw[f], want_rows=True)
        rows, n_out = b.normalize_host(raw[f], cnt[f])
        assoc, labels, dbn = b.step_host(rows, n_out, dts[f])
        assert np.array_equal(r["n_out"], n_out) and np.array_equal(r["db_n"], dbn), f
        assert np.array_equal(r["n_tracks"], b.num_tracks()), f
        for s in range(S):
            k = int(n_out[s])
            assert np.array_equal(r["rows"][s, :k], rows[s, :k]), (f, s)
            assert np.array_equal(r["assoc"][s, :k], assoc[s, :k]), (f, s)
            if dbn[s] > 0:
                assert np.array_equal(r["labels"][s, : dbn[s]], labels[s, : dbn[s]]), (f, s)
            o_rows = co.normalize(ocfg, raw[f, s, : max(int(cnt[f, s]), 0)])
            assert len(o_rows) == k
            if k:
                oa, ol = scenes[s].track(o_rows, float(dts[f, s]))
                assert np.array_equal(r["assoc"][s, :k], oa), (f, s)
        if f == 2:
            assert int(r["n_out"][4]) == 0 and int(r["db_n"][4]) == -1      # every row filtered out: the frame never reached track()
    ntr = a.num_tracks(); trk = a.tracks(cap=max(int(ntr.max()), 1))
    for s in range(S):
        assert_tracks_match(trk[s, : ntr[s]], scenes[s].tracks(), ctx=f"scene {s}", exact=True)
    # a loud error in the same call: a count the context was not sized for
    bad = cnt[0].copy(); bad[3] = N + 1
    with pytest.raises(_lib.MmwError) as ei:
        a.frame_host(bad, dts[0], pts=pts[0].astype(np.float64))
    assert ei.value.code == _lib.E_ARG and "scene 3" in str(ei.value)
    a.close(); b.close()


def test_offline_loop_fused_calls_and_on_device_posture():
    """TrackBuffer.track_raw (normalize_data + track, one round trip) + estimate_posture with a MarsCNN on the GPU (features, CNN
    and keypoint scatter stay on the device) against the separate reference-shaped calls -- utils.normalize_data, TrackBuffer.track,
    estimate_posture through a Keras-style .predict object -- frame by frame: same rows, association, tracks, keypoints within
    the CNN tolerance of the fp64 oracle."""
    import bench_ingest
    from mmwave_msc_amd import constants as const
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    from mmwave_msc_amd.synth import make_scene
    from mmwave_msc_amd.tracking import BatchedData, TrackBuffer
    from mmwave_msc_amd.utils import normalize_data
    from oracle.mars_np import mars_forward_np
    w = random_keras_weights(seed=8, frames=3)
    model = MarsCNN.from_keras_weights(w).to("cuda:0")

    class KerasLike:   # the reference's model object: .predict(ndarray) -> ndarray (Tracking.py:732), here the fp64 oracle CNN
        def predict(self, x, verbose=0):
            return mars_forward_np(w, np.asarray(x, dtype=np.float64)).astype(np.float32)

    p, c, d = make_scene(77, 14, 220, 2, ragged=True)
    ang = np.radians(const.S_TILT)
    raw = bench_ingest.raw_rows_from_normalised(p, float(np.cos(ang)), float(np.sin(ang)), float(const.S_HEIGHT)).astype(np.float64)
    fused, fb = TrackBuffer(max_pts=256), BatchedData()
    plain, pb = TrackBuffer(max_pts=256), BatchedData()
    for f in range(14):
        n = int(c[f])
        det = {k: list(raw[f, :n, i]) for i, k in enumerate(("x", "y", "z", "doppler", "peakVal"))}
        fused.dt = plain.dt = float(d[f])
        kept, rows = fused.track_raw(det, fb, want_rows=True)
        eff = normalize_data(det)
        assert kept == eff.shape[0] and np.array_equal(rows, eff), f
        if eff.shape[0]:
            plain.track(eff, pb)
            assert np.array_equal(fused.last_assoc, plain.last_assoc), f
        fused.estimate_posture(model)
        plain.estimate_posture(KerasLike())
        ta, tb_ = fused.effective_tracks, plain.effective_tracks
        assert len(ta) == len(tb_), f
        for x, y in zip(ta, tb_):
            assert np.array_equal(x.state.x, y.state.x) and np.array_equal(x.state.P, y.state.P) and x.uid == y.uid, f
            assert np.abs(x.keypoints - y.keypoints).max() <= KP_TOL * max(1.0, float(np.abs(y.keypoints).max())), f
    assert len(fused.effective_tracks) >= 1
    fused.close(); plain.close()


def test_attached_posture_model_one_round_trip_per_frame():
    """TrackBuffer.attach_posture_model: track_raw() queues estimate_posture behind the step (mmw_frame_posture_host: feature
    tensors, fp32 CNN and keypoint assignment with a row count only the device knows) and the loop's own estimate_posture() call
    is a no-op.  Against the unattached loop (the same kernels launched call by call): every keypoint BIT-equal; against the
    fp64 oracle CNN: within the CNN tolerance; frames that are skipped (no row passes the scene filter) and frames without
    tracks leave the keypoints alone."""
    import bench_ingest
    from mmwave_msc_amd import constants as const
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    from mmwave_msc_amd.synth import make_scene
    from mmwave_msc_amd.tracking import BatchedData, TrackBuffer
    from oracle.mars_np import mars_forward_np
    w = random_keras_weights(seed=9, frames=3)
    model = MarsCNN.from_keras_weights(w).to("cuda:0")

    class KerasLike:
        def predict(self, x, verbose=0):
            return mars_forward_np(w, np.asarray(x, dtype=np.float64)).astype(np.float32)

    p, c, d = make_scene(91, 16, 220, 3, ragged=True)
    ang = np.radians(const.S_TILT)
    raw = bench_ingest.raw_rows_from_normalised(p, float(np.cos(ang)), float(np.sin(ang)), float(const.S_HEIGHT)).astype(np.float64)
    from mmwave_msc_amd.utils import normalize_data
    att, ab = TrackBuffer(max_pts=256), BatchedData()
    att2, ab2 = TrackBuffer(max_pts=256), BatchedData()   # attached too, driven through normalize_data + track() (the rows form)
    dev, db = TrackBuffer(max_pts=256), BatchedData()
    ora, ob = TrackBuffer(max_pts=256), BatchedData()
    assert att.attach_posture_model(model) is True and att2.attach_posture_model(model) is True
    assert att.attach_posture_model(KerasLike()) is False and att._fused_model is model   # (not a MarsCNN: refused, nothing changes)
    estimated = 0
    for f in range(16):
        n = int(c[f])
        if f == 9:   # a frame none of whose rows pass the scene filter: skipped by the loop (offline_main.py:56), nothing estimated
            det = {k: [100.0] * 5 for k in ("x", "y", "z", "doppler", "peakVal")}
        else:
            det = {k: list(raw[f, :n, i]) for i, k in enumerate(("x", "y", "z", "doppler", "peakVal"))}
        for tb, bb, m in ((att, ab, model), (dev, db, model), (ora, ob, KerasLike())):
            tb.dt = float(d[f])
            kept = tb.track_raw(det, bb)
            assert (kept == 0) == (f == 9)
            if kept:
                if tb is att:
                    assert tb._posture_done
                tb.estimate_posture(m)
                if tb is att:
                    assert not tb._posture_done
        att2.dt = float(d[f])
        eff = normalize_data(det)
        if eff.shape[0]:
            att2.track(eff, ab2)
            assert att2._posture_done
            att2.estimate_posture(model)
        t2 = att2.effective_tracks
        ta, td, to = att.effective_tracks, dev.effective_tracks, ora.effective_tracks
        assert len(ta) == len(td) == len(to) == len(t2), f
        for x, y in zip(ta, t2):
            assert np.array_equal(x.state.x, y.state.x) and np.array_equal(x.keypoints, y.keypoints), f
        for x, y, z in zip(ta, td, to):
            assert np.array_equal(x.state.x, y.state.x) and np.array_equal(x.state.P, y.state.P) and x.uid == y.uid, f
            assert np.array_equal(x.keypoints, y.keypoints), f
            assert np.abs(x.keypoints - z.keypoints).max() <= KP_TOL * max(1.0, float(np.abs(z.keypoints).max())), f
            estimated += int(not np.array_equal(x.keypoints, np.asarray(const.MODEL_DEFAULT_POSTURE, dtype=np.float32)))
    assert len(att.effective_tracks) >= 2 and estimated >= 10
    att.attach_posture_model(None)
    assert att._fused_model is None
    att.close(); att2.close(); dev.close(); ora.close()


def test_attach_posture_refuses_what_it_cannot_run():
    """mmw_attach_posture / mmw_frame_posture_host argument errors are loud: a context of several scenes, another ring length, a
    frame before any model was attached, null weight pointers."""
    import ctypes as C
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    model = MarsCNN.from_keras_weights(random_keras_weights(seed=2, frames=3)).to("cuda:0")
    two = SceneBatch(_lib.default_config(), 2, 64)
    with pytest.raises(_lib.MmwError) as e:
        two.attach_posture(model)
    assert e.value.code == _lib.E_ARG and "one-scene" in str(e.value)
    two.close()
    short = SceneBatch(_lib.default_config(fb_frames_batch=1), 1, 64)   # ring of 2 frames: not the 3-frame model's
    with pytest.raises(_lib.MmwError):
        short.attach_posture(model)
    short.close()
    one = SceneBatch(_lib.default_config(), 1, 64)
    n, dt, pts = np.array([4], np.int32), np.array([0.1]), np.zeros((1, 64, 8))
    with pytest.raises(_lib.MmwError) as e:
        one.frame_host(n, dt, pts=pts, posture=True)
    assert e.value.code == _lib.E_ARG and "no model attached" in str(e.value)
    m = _lib.MmwPostureModel()   # all pointers null
    assert one.L.mmw_attach_posture(one.h, C.byref(m)) == _lib.E_ARG
    one.attach_posture(model)
    r = one.frame_host(n, dt, pts=pts, posture=True)
    assert r["posture_rows"] == 0   # (four points: no track yet)
    one.attach_posture(None)
    with pytest.raises(_lib.MmwError):
        one.frame_host(n, dt, pts=pts, posture=True)
    one.close()


def test_device_pointer_posture_path_is_ordered_with_torch():
    """features_dev -> torch CNN -> set_keypoints_dev with NO host synchronisation in between, torch on its default
    stream: the context must run on that very stream (SceneBatch.follow_torch_stream; include/mmw.h
    MMW_STREAM_LEGACY), otherwise set_keypoints reads the CNN's output buffer before the CNN has written it.  Two
    models back to back into the same (re-used) output allocation: the keypoints must be the second model's."""
    import torch
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    from mmwave_msc_amd.synth import make_batch
    S, N, F = 64, 256, 6
    pts, cnt, dts = make_batch(range(S), F, N, 3)
    sb = SceneBatch(_lib.default_config(tr_max_tracks=4), S, N)
    for f in range(F):
        sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
    dev = torch.device("cuda:0")
    sb.follow_torch_stream()  # torch's current stream: the legacy default one here
    cap = S * sb.track_cap
    d_feat = torch.empty((cap, sb.ring, 8, 8, 5), dtype=torch.float32, device=dev)
    d_owner = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    models = [MarsCNN.from_keras_weights(random_keras_weights(seed, sb.ring)).to(dev) for seed in (1, 2)]
    nrow = 0
    for m in models:
        nrow = sb.features_dev(d_feat.data_ptr(), d_owner.data_ptr(), cap)
        with torch.no_grad():
            kp = m(d_feat[:nrow])
        sb.set_keypoints_dev(kp.data_ptr(), d_owner.data_ptr(), nrow)
        del kp  # the caching allocator hands the same block to the next model's output
    assert nrow > 0
    torch.cuda.synchronize()
    with torch.no_grad():
        want = models[1](d_feat[:nrow]).cpu().numpy()
    owner = d_owner[:nrow].cpu().numpy()
    table = sb.tracks()
    for i, (s, j) in enumerate(owner):
        got = table[s][j]["keypoints"]
        assert np.array_equal(np.asarray(got, dtype=np.float32), want[i]), (s, j)
    sb.close()


def test_batcheddata_pop_frame_matches_reference_recording():
    """BatchedData.pop_frame() (Tracking.py:66-71) between tracked frames -- empty ring, one frame, a full ring,
    twice in a row: ring sizes and contents, association and track count as the reference recorded them."""
    from mmwave_msc_amd.tracking import BatchedData, TrackBuffer
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "popframe.npz"))
    pops = set(int(p) for p in g["pops"])
    tb, batch = TrackBuffer(max_pts=64), BatchedData()
    for f in range(len(g["cnt"])):
        c = int(g["cnt"][f])
        if f in pops:
            batch.pop_frame()
            if tb._sb is not None:
                want = [int(v) for v in g["after_pop"][f] if v >= 0]
                assert [len(fr) for fr in batch.buffer] == want, f
        tb.dt = float(g["dt"][f])
        tb.track(g["pts"][f, :c].astype(np.float64), batch)
        assert np.array_equal(np.asarray(tb.last_assoc, dtype=np.int16), g["assoc"][f, :c]), f
        assert len(tb.effective_tracks) == int(g["n_tracks"][f]), f
        sizes = [len(fr) for fr in batch.buffer]
        assert sizes == [int(v) for v in g["ring_n"][f, : int(g["ring_len"][f])]], f
        eff = batch.effective_data
        n = sum(sizes)
        got = np.asarray(eff, dtype=np.float64).reshape(-1, 8) if n else np.zeros((0, 8))
        assert np.array_equal(got, g["ring_rows"][f, :n]), f
    tb.close()


def test_trackbuffer_track_on_an_empty_cloud_is_a_real_call():
    """TrackBuffer.track(empty) is not a no-op in the reference (Tracking.py:664-703 on an empty array: predict, lifetime
    += dt, expiry, _update_all, an empty frame pushed into the ring); the golden `empty_tracked` holds such a run."""
    from mmwave_msc_amd.tracking import BatchedData, TrackBuffer
    g = load_scenario("empty_tracked")
    tb, batch = TrackBuffer(max_pts=128), BatchedData()
    for f in range(52):
        c = int(g["cnt"][f])
        tb.dt = float(g["dt"][f])
        tb.track(g["pts"][f, :c].astype(np.float64) if c else np.empty((0, 8)), batch)
        nt = int(g["n_tracks"][f])
        tracks = tb.effective_tracks
        assert len(tracks) == nt, f
        for j, t in enumerate(tracks):
            assert t.lifetime == g["tracks"][f, j]["lifetime"] and t.cluster.point_num == g["tracks"][f, j]["point_num"]
        assert [len(fr) for fr in batch.buffer] == list(g["ring_n"][f, : g["ring_len"][f]])
    assert len(tb.effective_tracks) == 0 and g["n_tracks"][48] == 2   # both tracks expired during the empty run
    tb.close()


def test_batcheddata_init_data_and_change_buffer_size():
    """BatchedData(init_data) (Tracking.py:38-41) and BatchedData.change_buffer_size (Tracking.py:60-64) through the
    reference-shaped objects, against the recording `batch_init_resize` of the reference doing the same."""
    from mmwave_msc_amd.tracking import BatchedData, TrackBuffer
    g = load_scenario("batch_init_resize")
    resize = {int(a): int(b) for a, b in g["overrides"]["BATCH_RESIZE"]}
    tb, batch = TrackBuffer(max_pts=160), BatchedData(g["batch_init"])
    assert [len(fr) for fr in batch.buffer] == [40] and batch.effective_data.shape == (40, 8)   # before any track()
    for f in range(g["pts"].shape[0]):
        c = int(g["cnt"][f])
        if f in resize:
            batch.change_buffer_size(resize[f])
            assert batch.size == resize[f]
        tb.dt = float(g["dt"][f])
        tb.track(g["pts"][f, :c].astype(np.float64), batch)
        assert np.array_equal(tb.last_assoc, g["assoc"][f, :c]), f
        dbn = int(g["db_n"][f])
        assert (tb.last_db_labels is None) == (dbn < 0)
        if dbn >= 0:
            assert np.array_equal(tb.last_db_labels, g["labels"][f, :dbn]), f
        assert len(tb.effective_tracks) == g["n_tracks"][f]
        assert [len(fr) for fr in batch.buffer] == list(g["ring_n"][f, : g["ring_len"][f]]), f
    tb.close()
    # a size change BEFORE the first track() call is applied when the TrackBuffer binds the object
    tb, batch = TrackBuffer(max_pts=160), BatchedData()
    batch.change_buffer_size(1)
    for f in range(4):
        tb.dt = 0.1
        tb.track(g["pts"][f, : int(g["cnt"][f])].astype(np.float64), batch)
        assert len(batch.buffer) <= 1
    tb.close()
    with pytest.raises(Exception):
        batch2 = BatchedData()
        tb2 = TrackBuffer(max_pts=64)
        tb2.dt = 0.1
        tb2.track(g["pts"][0, :32].astype(np.float64), batch2)
        batch2.change_buffer_size(0)   # the reference's add_frame would spin forever: MMW_E_ARG here


def test_trackbuffer_raises_the_references_valueerror_on_nonfinite_rows():
    """The `nonfinite` recording (oracle/gen_golden.py): NaN / +-inf rows enter the global ring and sklearn's input validation
    raises ValueError out of the reference's track() on every frame apply_DBscan is reached while one is there
    (Utils.py:272-278, Tracking.py:693-697).  The drop-in TrackBuffer raises a ValueError on exactly those frames -- after the
    frame's results are in place --, keeps the state the exception leaves behind, and carries on like the reference when the
    caller catches it.  apply_DBscan / normalize_data of the Utils mirror behave as theirs on such rows too."""
    from mmwave_msc_amd import _lib, utils
    from mmwave_msc_amd import constants as const
    from mmwave_msc_amd.tracking import BatchedData, TrackBuffer
    g = load_scenario("nonfinite")
    saved = const.TR_MAX_TRACKS
    const.TR_MAX_TRACKS = int(g["overrides"]["TR_MAX_TRACKS"])
    try:
        tb, batch = TrackBuffer(max_pts=g["pts"].shape[1]), BatchedData()
        n_raised = 0
        for f in range(g["pts"].shape[0]):
            c = int(g["cnt"][f])
            tb.dt = float(g["dt"][f])
            raised = int(g["raised"][f])
            if raised:
                with pytest.raises(ValueError) as ei:
                    tb.track(g["pts"][f, :c].astype(np.float64), batch)
                assert isinstance(ei.value, _lib.MmwNonFinite) and ("NaN" if raised == 1 else "infinity") in str(ei.value), f
                assert tb.last_db_labels is None
                n_raised += 1
            else:
                tb.track(g["pts"][f, :c].astype(np.float64), batch)
                dbn = int(g["db_n"][f])
                assert (tb.last_db_labels is None) == (dbn < 0), f
                if dbn >= 0:
                    assert np.array_equal(tb.last_db_labels, g["labels"][f, :dbn]), f
            assert np.array_equal(tb.last_assoc, g["assoc"][f, :c]), f
            nt = int(g["n_tracks"][f])
            tracks = tb.effective_tracks
            assert len(tracks) == nt, f
            for j, t in enumerate(tracks):
                w = g["tracks"][f, j]
                assert close64(t.state.x[:, 0], w["x"]) and close64(t.state.P, w["P"]) and t.cluster.point_num == w["point_num"], (f, j)
            assert [len(fr) for fr in batch.buffer] == list(g["ring_n"][f, : g["ring_len"][f]]), f
        assert n_raised == int((g["raised"] > 0).sum()) >= 6
        tb._sb.check()   # (the sticky bits were cleared frame by frame)
        tb.close()
    finally:
        const.TR_MAX_TRACKS = saved
    # Utils.apply_DBscan on a cloud with a non-finite value: sklearn's ValueError; which message: NaN wins over infinity
    z = np.load(os.path.join(GOLDEN, "dbscan.npz"))
    for n in (61, 700, 1536):
        for col, val, word in ((0, np.nan, "NaN"), (7, np.inf, "infinity"), (2, -np.inf, "infinity")):
            pts = z[f"pts_{n}"].astype(np.float64)
            pts[n // 2, col] = val
            with pytest.raises(ValueError) as ei:
                utils.apply_DBscan(pts)
            assert word in str(ei.value), (n, col)
        pts = z[f"pts_{n}"].astype(np.float64)
        pts[0, 6], pts[n - 1, 1] = np.inf, np.nan
        with pytest.raises(ValueError) as ei:
            utils.dbscan_labels(pts)
        assert "NaN" in str(ei.value)
    # ... and sklearn's other refusals: parameters outside DBSCAN's constraints, no samples, a 1-D array (np.array([]) is what
    # BatchedData.clear() leaves in effective_data, Tracking.py:57-58)
    ok = z["pts_61"].astype(np.float64)
    for args, word in (((ok, 0.0, 5), "eps"), ((ok, float("nan"), 5), "eps"), ((ok, 0.3, 0), "min_samples"),
                       ((np.zeros((0, 8)), 0.3, 5), "0 sample"), ((np.array([]), 0.3, 5), "2D array")):
        with pytest.raises(ValueError) as ei:
            utils.apply_DBscan(*args)
        assert word in str(ei.value), (word, str(ei.value))
    # twelve points: the first cloud size sklearn hands to the BallTree; eleven: brute force (both golden: dbscan_small.npz)
    zs = np.load(os.path.join(GOLDEN, "dbscan_small.npz"))
    for n in (3, 8, 11, 12, 13):
        for c in range(0, 40, 7):
            for ms in (2, 5, 8):
                assert np.array_equal(utils.dbscan_labels(zs[f"pts_{n}"][c], min_samples=ms), zs[f"labels_{n}_{ms}"][c]), (n, c, ms)
    # normalize_data: one non-finite coordinate turns all three into NaN (the reference's full 4 x 4 products: 0 * inf) and the
    # scene filter drops the row; a non-finite doppler leaves three NaN velocities on a kept row -- against the oracle's
    # restatement, which tests/test_reference_fuzz.py pins on the live reference with the same rows
    from oracle import c_oracle as co
    from tests._fuzz import nonfinite_raw_rows
    zn = np.load(os.path.join(GOLDEN, "normalize.npz"))
    for seed in range(4):
        raw = nonfinite_raw_rows(seed)
        det = {"x": list(raw[:, 0]), "y": list(raw[:, 1]), "z": list(raw[:, 2]), "doppler": list(raw[:, 3]), "peakVal": list(raw[:, 4])}
        out = utils.normalize_data(det)
        want = co.normalize(co.default_config(s_height=float(zn["s_height"]), s_tilt=float(zn["s_tilt"])), raw)
        assert out.shape == want.shape and np.array_equal(out, want, equal_nan=True), seed
        assert np.isnan(out[:, 3:6]).any() and not np.isnan(out[:, :3]).any() and len(out) < 140
