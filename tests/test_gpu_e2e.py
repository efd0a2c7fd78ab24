"""GPU tests of the end-to-end path (BASELINE.json configs[3]) and of the N > 1 exchange step on real contexts:

* track -> features -> MARS CNN -> keypoints every frame for 256 scenes x 256 points x 4 tracks, pipelined on two
  streams (mmwave_msc_amd/posture.py), against the C oracle's tracker / feature map and the fp64 numpy CNN;
* the pipelined loop against the plain, frame-synchronous one (same final keypoints);
* `mmw_features_async` / `mmw_set_keypoints_uid` after the track list has been re-ordered;
* two ranks, two contexts: `mmw_track_table` with `scene_base`, gathered, equals the table of one context over all scenes.

CNN oracle = oracle/mars_np.py: parity unpinned against Keras (absent from the image); tolerance 1e-4."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KP_TOL = 1e-4


def test_e2e_256_scenes_every_frame_vs_oracle():
    import bench_e2e
    ref = bench_e2e.oracle_reference(workers=4)
    assert ref["samples_cnn"] > 256  # most scenes hold tracks with keypoints from the CNN
    res = bench_e2e.e2e_parity_leg(ref, 0)
    assert res["tracker_state_bit_equal_vs_oracle"], res
    assert res["tracks_checked"] >= ref["samples_cnn"]
    assert res["keypoint_max_err"] <= KP_TOL, res


def test_e2e_one_rank_of_configs4_vs_oracle():
    """BASELINE.json configs[4] is configs[3]'s loop on 4096 scenes x 512 points x 8 tracks over 8 GPUs: one rank's shard
    (512 scenes) end to end against the oracle -- tracker state bit-equal, keypoints within 1e-4 of the fp64 CNN."""
    import bench_e2e
    ref = bench_e2e.oracle_reference(workers=4, par=dict(S=512, N=512, T=8, F=6, seed0=512 * 3, label="configs[4], rank 3 of 8"))
    assert ref["samples_cnn"] > 1024
    res = bench_e2e.e2e_parity_leg(ref, 0)
    assert res["tracker_state_bit_equal_vs_oracle"], res
    assert res["tracks_checked"] >= ref["samples_cnn"]
    assert res["keypoint_max_err"] <= KP_TOL, res


def test_e2e_configs4_at_full_size_vs_oracle():
    """BASELINE.json configs[4] at its stated size, on the one GPU: 4096 scenes x 512 points x TR_MAX_TRACKS 8, cluster -> track ->
    feature map -> MARS CNN -> keypoints EVERY frame (offline_main.py:57-60) for 6 frames through the pipelined PosturePipeline --
    the tracker state of ALL 4096 scenes bit-equal to the C oracle's, the keypoints of every live track within 1e-4 of the fp64
    CNN (oracle/mars_np.py).  The oracle's side runs in child processes, one block of scenes per core (scenes are independent)."""
    import bench
    import bench_e2e
    ref = bench_e2e.oracle_reference(workers=1, par=dict(S=4096, N=512, T=8, F=6, seed0=0, label="configs[4] at 4096 scenes, one GPU"),
                                     procs=max(2, min(16, bench.effective_cores())))
    assert len(ref["finals"]) == 4096 and ref["samples_cnn"] > 8192
    res = bench_e2e.e2e_parity_leg(ref, 0)
    assert res["tracker_state_bit_equal_vs_oracle"], res
    assert res["tracks_checked"] >= ref["samples_cnn"] > 8192
    assert res["keypoint_max_err"] <= KP_TOL, res


def _run_loop(sb, pts, cnt, dts, model, pipelined):
    import torch
    from mmwave_msc_amd.posture import PosturePipeline
    dev = torch.device("cuda", 0)
    F, S = cnt.shape
    cap = S * sb.track_cap
    pipe = PosturePipeline(sb, model, cap, overlap=pipelined)
    with torch.cuda.stream(pipe.A):
        d_pts = torch.from_numpy(pts).to(dev).double()
        d_cnt = torch.from_numpy(cnt).to(dev)
        d_dt = torch.from_numpy(dts).to(dev)
    pipe.A.synchronize()
    sb.reset()
    if pipelined:
        for f in range(F):
            sb.step_dev(d_pts[f].data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr())
            pipe.after_step()
        pipe.close()
    else:  # the reference's order: estimate_posture completes before the next track()
        for f in range(F):
            sb.step_dev(d_pts[f].data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr())
            feat, owner = sb.features_host()
            if len(owner):
                sb.set_keypoints_host(model.predict_numpy(feat), owner)
    sb.check()
    ntr = sb.num_tracks()
    return ntr, sb.tracks(cap=max(int(ntr.max()), 1))


def test_pipelined_posture_equals_frame_synchronous_loop():
    """Tracks expire and the list is compacted while a frame's CNN is still in flight (presence gaps of > 3 s)."""
    import torch
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    from mmwave_msc_amd.synth import make_scene
    S, N, F = 24, 256, 48
    ps, cs, ds = [], [], []
    for s in range(S):
        presence = np.ones((F, 3), dtype=bool)
        presence[6 + (s % 5): 6 + (s % 5) + 34, s % 3] = False   # one target vanishes for 3.4 s: its track expires mid-run
        p, c, d = make_scene(700 + s, F, N, 3, ragged=(s % 2 == 0), presence=presence)
        ps.append(p); cs.append(c); ds.append(d)
    pts, cnt, dts = np.stack(ps, 1), np.stack(cs, 1), np.stack(ds, 1)
    model = MarsCNN.from_keras_weights(random_keras_weights(3, 3)).to("cuda:0")
    sb = SceneBatch(_lib.default_config(tr_max_tracks=4), S, N, device=0)
    n1, t1 = _run_loop(sb, pts, cnt, dts, model, pipelined=True)
    n2, t2 = _run_loop(sb, pts, cnt, dts, model, pipelined=False)
    assert np.array_equal(n1, n2) and int(n1.sum()) > S
    uids1 = [set(t1[s, : n1[s]]["uid"]) for s in range(S)]
    assert any(max(u) >= 3 for u in uids1 if u), "no scene re-spawned a track: the scenario does not exercise list compaction"
    for s in range(S):
        a, b = t1[s, : n1[s]], t2[s, : n2[s]]
        assert np.array_equal(a["uid"], b["uid"]) and np.array_equal(a["x"], b["x"])
        assert np.allclose(a["keypoints"], b["keypoints"], rtol=0, atol=1e-6), s
    sb.close()
    torch.cuda.synchronize()


def test_pipeline_repairs_out_of_range_samples_on_the_tracker_stream():
    """The split-fp16 CNN's fp32 repair (samples whose inputs or activations leave fp16's range: mars.MarsCNN.range_fixup) runs on
    the TRACKER stream in the pipelined schedule, in front of the frame's scatter, off a fix-up list of the frame's own -- not on
    the CNN stream it used to lengthen by five launches a frame.  A feature tensor of the last frame is pushed out of fp16's range
    behind the feature kernel: the track it belongs to ends up with the keypoints of Keras' fp32 arithmetic, the others with
    the split arithmetic's, and the run reports one repaired batch and nothing left meaningless."""
    import torch
    import bench
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    from mmwave_msc_amd.posture import PosturePipeline
    S, N, T, F = 64, 256, 4, 8
    pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=4)
    dev = torch.device("cuda", 0)
    model = MarsCNN.from_keras_weights(random_keras_weights(seed=4, frames=3)).to(dev)
    sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
    pipe = PosturePipeline(sb, model, S * sb.track_cap, overlap=True)
    if pipe.B is pipe.A:
        pipe.close(); sb.close()
        pytest.skip("no second stream on an independent hardware queue: the serial schedule keeps the repair behind Dense-2")
    assert pipe._defer_fixup
    with torch.cuda.stream(pipe.A):
        d_pts = torch.from_numpy(pts).to(dev).double(); d_cnt = torch.from_numpy(cnt).to(dev); d_dt = torch.from_numpy(dts).to(dev)
    pipe.A.synchronize()
    for f in range(F):
        sb.step_dev(d_pts[f].data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr())
        pipe.after_step()
    torch.cuda.synchronize()                      # the last frame's feature tensors are written, its CNN is not queued yet
    d = (F - 1) % pipe.NBUF
    with torch.cuda.stream(pipe.A):
        pipe.feat[d][0, 0, 0, 0, :] = 1.0e5       # sample 0: an input beyond fp16's 65 504
        x_mod = pipe.feat[d].clone()
        owner0 = pipe.owner[d][0].clone()
    torch.cuda.synchronize()
    before = model.range_recomputed
    pipe.drain()
    assert model.range_recomputed == before + 1 and not pipe.range_overflowed
    n = pipe.rows[d]
    assert n > 8
    with torch.no_grad():
        want32 = model(x_mod[:n], arith=model.fp32_arith()).float().cpu().numpy()
    s0, j0 = (int(v) for v in owner0.cpu().numpy())
    own = pipe.owner[d][:n].cpu().numpy()
    ntr = sb.num_tracks()
    trk = sb.tracks(cap=max(int(ntr.max()), 1))
    got0 = trk[s0, j0]["keypoints"]
    assert np.isfinite(got0).all()
    assert np.abs(got0 - want32[0]).max() <= 1e-4 * max(1.0, float(np.abs(want32[0]).max()))
    for i in range(1, min(n, 40)):               # the others: untouched by the repair, the split arithmetic's (within 1e-4 of fp32's)
        s_i, j_i = int(own[i, 0]), int(own[i, 1])
        assert np.abs(trk[s_i, j_i]["keypoints"] - want32[i]).max() <= 1e-4 * max(1.0, float(np.abs(want32[i]).max())), i
    # ... and MORE out-of-range samples than the device-side repair holds (MMW_RANGE_FIXUP_CAP = 64) in one batch: their keypoints
    # would be meaningless -- an error at drain(), not a warning; close() still hands the tracker back
    from mmwave_msc_amd.posture import PostureRangeError
    sb.reset()
    for f in range(F):
        sb.step_dev(d_pts[f].data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr())
        pipe.after_step()
    torch.cuda.synchronize()
    with torch.cuda.stream(pipe.A):
        pipe.feat[d][:80, 0, 0, 0, :] = 1.0e5
    torch.cuda.synchronize()
    with pytest.raises(PostureRangeError):
        pipe.close()
    assert pipe.range_overflowed and pipe._closed
    sb.close()


def test_features_async_tickets_and_uid_scatter():
    import torch
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.synth import make_batch
    S, N, F = 6, 256, 6
    pts, cnt, dts = make_batch(range(50, 50 + S), F, N, 3)
    sb = SceneBatch(_lib.default_config(), S, N, device=0)
    for f in range(F):
        sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
    feat_h, owner_h = sb.features_host()
    cap = S * sb.track_cap
    dev = torch.device("cuda", 0)
    feat = torch.zeros((cap, 3, 8, 8, 5), dtype=torch.float32, device=dev)
    owner = torch.zeros((cap, 2), dtype=torch.int32, device=dev)
    uid = torch.full((cap,), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for ticket in (0, 1, 2):
        sb.features_async(feat.data_ptr(), owner.data_ptr(), uid.data_ptr(), cap, ticket=ticket)
    rows = [sb.features_wait(t) for t in (2, 0, 1)]
    assert rows == [len(owner_h)] * 3
    sb.synchronize()
    n = rows[0]
    assert np.array_equal(feat[:n].cpu().numpy(), feat_h) and np.array_equal(owner[:n].cpu().numpy(), owner_h)
    trk = sb.tracks()
    want_uid = np.array([trk[s, j]["uid"] for s, j in owner_h])
    assert np.array_equal(uid[:n].cpu().numpy(), want_uid)
    # rows in reverse order, one row for a uid that does not exist: matched by (scene, uid), the stranger is dropped
    kp = torch.arange(n * 57, dtype=torch.float32, device=dev).reshape(n, 57)
    perm = torch.arange(n - 1, -1, -1, device=dev)
    kp_p, owner_p, uid_p = kp[perm].contiguous(), owner[:n][perm].contiguous(), uid[:n][perm].contiguous()
    owner_p[:, 1] = 63  # the list position is not what the uid form looks at
    uid_bad = uid_p.clone()
    uid_bad[0] = 999
    torch.cuda.synchronize()
    before = sb.tracks()
    sb.set_keypoints_uid_dev(kp_p.data_ptr(), owner_p.data_ptr(), uid_bad.data_ptr(), n)
    sb.synchronize()
    after = sb.tracks()
    kp_h = kp.cpu().numpy()
    for i, (s, j) in enumerate(owner_h):
        if i == n - 1:   # its row carried the unknown uid
            assert np.array_equal(after[s, j]["keypoints"], before[s, j]["keypoints"])
        else:
            assert np.array_equal(after[s, j]["keypoints"], kp_h[i])
    with pytest.raises(_lib.MmwError):
        sb.features_async(feat.data_ptr(), owner.data_ptr(), uid.data_ptr(), 1, ticket=0)
        sb.features_wait(0)   # cap_rows too small -> MMW_E_CAPACITY, as mmw_features
    sb.close()


# ---- N > 1 on real contexts ------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, total, n_pts, frames, slots, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.dist import all_gather_tables, shard_range, summaries_to_tensor, tensor_to_summaries
    from mmwave_msc_amd.synth import make_batch
    lo, hi = shard_range(total, rank, world)
    pts, cnt, dts = make_batch(range(300 + lo, 300 + hi), frames, n_pts, 3)
    sb = SceneBatch(_lib.default_config(), hi - lo, n_pts, device=0)   # both ranks share the one GPU of the box
    for f in range(frames):
        sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
    local = sb.track_table_host(slots, scene_base=lo)
    glob = tensor_to_summaries(all_gather_tables(summaries_to_tensor(local)), slots)
    sb.close()
    q.put((rank, glob))
    dist.destroy_process_group()


def test_two_ranks_two_contexts_track_table_gather():
    import torch.multiprocessing as mp
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.synth import make_batch
    total, n_pts, frames, slots = 7, 256, 6, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, total, n_pts, frames, slots, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # one context over all scenes = what the gathered table must equal, row for row
    pts, cnt, dts = make_batch(range(300, 300 + total), frames, n_pts, 3)
    sb = SceneBatch(_lib.default_config(), total, n_pts, device=0)
    for f in range(frames):
        sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
    want = sb.track_table_host(slots, scene_base=0)
    sb.close()
    assert want.dtype == _lib.SUMMARY_DTYPE and int(want["alive"].sum()) >= total
    for r in (0, 1):
        got = res[r]
        assert got.shape == (total, slots)
        assert np.array_equal(got["scene"], np.repeat(np.arange(total)[:, None], slots, 1))
        assert np.array_equal(got["slot"], np.repeat(np.arange(slots)[None, :], total, 0))
        for name in want.dtype.names:
            assert np.array_equal(got[name], want[name]), name


def test_track_table_carries_the_fade_square_of_the_output_step():
    """k_table's epilogue = Visualizer.calc_fade_square over Utils.calc_projection_points (Visualizer.py:14-29,
    Utils.py:180-219): equal, after the one rounding to float32, to utils.fade_squares -- which tests/test_host_logic.py
    pins bit-equal to values recorded from the reference's own functions -- on the fp64 state and the float32 keypoints."""
    from mmwave_msc_amd import _lib, utils
    from mmwave_msc_amd import constants as const
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.synth import make_batch
    S, N, F, slots = 5, 256, 6, 4
    pts, cnt, dts = make_batch(range(60, 60 + S), F, N, 3)
    sb = SceneBatch(const.to_config(), S, N, device=0)
    for f in range(F):
        sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
    _, owner = sb.features_host()
    rng = np.random.default_rng(11)
    kp = rng.normal(0, 0.5, size=(len(owner), 57)).astype(np.float32)
    sb.set_keypoints_host(kp, owner)
    ntr = sb.num_tracks()
    trk = sb.tracks(cap=slots)
    tab = sb.track_table_host(slots)
    assert int(ntr.sum()) == len(owner) >= S
    for s in range(S):
        for j in range(slots):
            row = tab[s, j]
            if j >= ntr[s]:
                assert row["alive"] == 0 and row["fade_x"] == 0 and row["fade_z"] == 0 and row["fade_size"] == 0
                continue
            px, pz, size = utils.fade_squares(trk[s, j]["x"], trk[s, j]["keypoints"])
            assert row["fade_x"] == np.float32(px) and row["fade_z"] == np.float32(pz) and row["fade_size"] == np.float32(size)
            assert const.V_SCREEN_FADE_SIZE_MIN - 1e-7 <= row["fade_size"] <= const.V_SCREEN_FADE_SIZE_MAX + 1e-7
    # other monitoring point / fade limits travel through mmw_config
    sb2 = SceneBatch(const.to_config(m_x=-0.5, m_y=-1.25, m_z=0.9, v_screen_fade_size_max=0.5, v_screen_fade_weight=0.02), S, N, device=0)
    for f in range(F):
        sb2.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
    sb2.set_keypoints_host(kp, owner)
    tab2, trk2 = sb2.track_table_host(slots), sb2.tracks(cap=slots)
    old = (const.M_X, const.M_Y, const.M_Z, const.V_SCREEN_FADE_SIZE_MAX, const.V_SCREEN_FADE_WEIGHT)
    try:
        const.M_X, const.M_Y, const.M_Z, const.V_SCREEN_FADE_SIZE_MAX, const.V_SCREEN_FADE_WEIGHT = -0.5, -1.25, 0.9, 0.5, 0.02
        px, pz, size = utils.fade_squares(trk2[0, 0]["x"], trk2[0, 0]["keypoints"])
    finally:
        const.M_X, const.M_Y, const.M_Z, const.V_SCREEN_FADE_SIZE_MAX, const.V_SCREEN_FADE_WEIGHT = old
    assert (tab2[0, 0]["fade_x"], tab2[0, 0]["fade_z"], tab2[0, 0]["fade_size"]) == (np.float32(px), np.float32(pz), np.float32(size))
    sb.close(); sb2.close()


@pytest.mark.parametrize("S,N,T,F", [(256, 256, 4, 30), (4096, 512, 8, 12)])
def test_baseline_configs_at_their_size_vs_oracle(S, N, T, F):
    """BASELINE.json configs[1] (256 scenes x 256 points x 4 tracks) and configs[2] (4096 x 512 x 8, the bench's
    workload with its side-stream DBSCAN workers) AT THEIR SIZE: after F frames the state of EVERY scene -- track
    count / order, x, P, centroid, spread, dispersion, lifetime, point counts, ring lengths -- is bit-equal to the C
    oracle's (run with OpenMP over the scenes), and so are the association vector and the DBSCAN labels of the last frame."""
    import bench
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from oracle import c_oracle as co
    pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=4)
    sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N, device=0)
    ob = co.OracleBatch(co.default_config(tr_max_tracks=T), S, N)
    co.batch_run_f32(ob, pts[:F - 1], cnt[:F - 1], dts[:F - 1], 0)
    for f in range(F):
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
    o_assoc, o_lab, o_dbn = ob.step(pts[F - 1].astype(np.float64), cnt[F - 1], dts[F - 1])
    assert np.array_equal(dbn, o_dbn)
    for s in range(S):
        c = cnt[F - 1, s]
        assert np.array_equal(assoc[s, :c], o_assoc[s, :c]), s
        if dbn[s] >= 0:
            assert np.array_equal(labels[s, : dbn[s]], o_lab[s, : dbn[s]]), s
    ntr = sb.num_tracks()
    trk = sb.tracks(cap=max(int(ntr.max()), 1))
    ln, rn = sb.batch_ring()
    for s in range(S):
        want = ob.scenes[s].tracks()
        assert ntr[s] == len(want), s
        for name in ("x", "P", "centroid", "min_vals", "max_vals", "spread_est", "group_disp_est", "n_est", "lifetime",
                     "point_num", "is_static", "ring_len", "ring_n"):
            assert np.array_equal(trk[s, : ntr[s]][name], want[name]), (s, name)
        assert np.array_equal(rn[s, : ln[s]], ob.scenes[s].batch_ring()), s
    sb.check()
    sb.close()


def test_local_sharded_tracker_two_contexts_on_two_threads_vs_oracle():
    """mmwave_msc_amd.dist.LocalShardedTracker: the job in ONE process, two contexts stepped concurrently from two host threads
    (both on this GPU: the C-ABI's contract is one thread per context, any number of contexts) -- every scene equals its
    oracle run, and the concatenated track table is the global one, ordered by scene id."""
    import torch
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.dist import LocalShardedTracker
    from mmwave_msc_amd.synth import make_batch
    from oracle import c_oracle as co
    S, N, F, T = 37, 192, 9, 4
    pts = np.zeros((F, S, N, 8), np.float32); cnt = np.zeros((F, S), np.int32); dts = np.zeros((F, S))
    for s in range(S):
        p, c, d = make_batch([8800 + s], F, N, 1 + s % T, ragged=(s % 3 == 0))
        pts[:, s], cnt[:, s], dts[:, s] = p[:, 0], c[:, 0], d[:, 0]
    lt = LocalShardedTracker(lambda: _lib.default_config(tr_max_tracks=T), S, N, devices=[0, 0])
    assert [(sh["lo"], sh["hi"]) for sh in lt.shards] == [(0, 19), (19, 37)] and lt.n_total == S

    def run_shard(g, sh):
        sb, lo, hi = sh["sb"], sh["lo"], sh["hi"]
        dev = torch.device("cuda", sh["device"])
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            sb.follow_torch_stream(st)
            d_cnt = torch.from_numpy(np.ascontiguousarray(cnt[:, lo:hi])).to(dev)
            d_dt = torch.from_numpy(np.ascontiguousarray(dts[:, lo:hi])).to(dev)
            for f in range(F):
                p32 = torch.from_numpy(np.ascontiguousarray(pts[f, lo:hi])).to(dev)
                sb.step_dev_f32(p32.data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr())
                st.synchronize()
        sb.check()
        return sb.num_tracks()

    counts = np.concatenate(lt.run(run_shard))
    ob = co.OracleBatch(co.default_config(tr_max_tracks=T), S, N)
    co.batch_run_f32(ob, pts, cnt, dts, 0)
    tab = lt.gather_table(T)
    assert tab.shape == (S, T) and np.array_equal(tab["scene"][:, 0], np.arange(S))
    for sh in lt.shards:
        ntr = sh["sb"].num_tracks()
        trk = sh["sb"].tracks(cap=max(int(ntr.max()), 1))
        for s in range(sh["lo"], sh["hi"]):
            want = ob.scenes[s].tracks()
            k = s - sh["lo"]
            assert counts[s] == ntr[k] == len(want), s
            for name in ("x", "P", "centroid", "spread_est", "group_disp_est", "lifetime", "point_num", "is_static", "ring_n"):
                assert np.array_equal(trk[k, : ntr[k]][name], want[name]), (s, name)
            alive = min(len(want), T)
            assert int(tab["alive"][s].sum()) == alive, s
            assert np.array_equal(tab["x"][s, :alive], want["x"][:alive].astype(np.float32)), s
    lt.close()


def test_local_sharded_tracker_with_posture_model_two_contexts():
    """Two contexts in ONE process, each on its own host thread with its own MarsCNN copy and PosturePipeline (the one-process
    form of SURVEY.md §8e).  The launchers of the CNN kernels keep per-device state (dynamic-LDS attribute, CU count) behind a
    std::once_flag: two first calls from two threads must not race, and every context ends with the oracle's tracker state and
    keypoints within 1e-4 of the fp64 CNN.  (On an 8-GPU node `devices` names eight GPUs; here both contexts share the one.)"""
    import torch
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.dist import LocalShardedTracker
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    from mmwave_msc_amd.posture import PosturePipeline
    from oracle import c_oracle as co
    from oracle.mars_np import mars_forward_np
    import bench
    S, N, F, T = 96, 256, 8, 4
    pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=4)
    w = random_keras_weights(seed=5, frames=3)
    lt = LocalShardedTracker(lambda: _lib.default_config(tr_max_tracks=T), S, N, devices=[0, 0])

    def run_shard(g, sh):
        sb, lo, hi = sh["sb"], sh["lo"], sh["hi"]
        dev = torch.device("cuda", sh["device"])
        torch.cuda.set_device(dev)
        model = MarsCNN.from_keras_weights(w).to(dev)
        pipe = PosturePipeline(sb, model, (hi - lo) * sb.track_cap, overlap=(g == 0))
        with torch.cuda.stream(pipe.A):
            d_pts = torch.from_numpy(np.ascontiguousarray(pts[:, lo:hi])).to(dev).double()
            d_cnt = torch.from_numpy(np.ascontiguousarray(cnt[:, lo:hi])).to(dev)
            d_dt = torch.from_numpy(np.ascontiguousarray(dts[:, lo:hi])).to(dev)
        pipe.A.synchronize()
        for f in range(F):
            sb.step_dev(d_pts[f].data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr())
            pipe.after_step()
        pipe.close()
        sb.check()
        ntr = sb.num_tracks()
        return ntr, sb.tracks(cap=max(int(ntr.max()), 1)), pipe.rows_total

    res = lt.run(run_shard)
    cfg = co.default_config(tr_max_tracks=T)
    worst, n_kp = 0.0, 0
    for sh, (ntr, trk, rows) in zip(lt.shards, res):
        assert rows > 0
        for s in range(sh["lo"], sh["hi"]):
            orc = co.OracleScene(cfg, N)
            for f in range(F):
                c = int(cnt[f, s])
                orc.track(pts[f, s, :c].astype(np.float64), float(dts[f, s]))
                feat, owner = orc.features()
                if len(owner):
                    orc.set_keypoints(mars_forward_np(w, feat.astype(np.float64)).astype(np.float32), owner)
            want = orc.tracks()
            k = s - sh["lo"]
            assert ntr[k] == len(want), s
            for name in ("x", "P", "centroid", "spread_est", "group_disp_est", "lifetime", "point_num", "is_static", "ring_n"):
                assert np.array_equal(trk[k, : ntr[k]][name], want[name]), (s, name)
            if len(want):
                wk = want["keypoints"].astype(np.float64)
                worst = max(worst, float((np.abs(trk[k, : ntr[k]]["keypoints"].astype(np.float64) - wk) / np.maximum(1.0, np.abs(wk))).max()))
                n_kp += len(want)
    assert n_kp > S and worst <= KP_TOL, (n_kp, worst)
    lt.close()


def _rccl_one_rank(port, q):
    """child process: a one-rank process group on RCCL (backend "nccl"), a real mmw_track_table all-gathered through it"""
    import os
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.dist import ShardedTracker, tensor_to_summaries
    from mmwave_msc_amd.synth import make_batch
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    S, N, F, slots = 24, 256, 6, 4
    pts, cnt, dts = make_batch(range(300, 300 + S), F, N, 3)
    st = ShardedTracker(_lib.default_config(), S, N, device=0)     # the context keeps ITS OWN stream: the gather hands over by event
    for f in range(F):
        st.sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
    out = []
    for _ in range(3):                                             # (the second and third call reuse counts and buffers)
        g = st.gather_table(slots, force_collective=True)
        torch.cuda.synchronize()
        out.append(tensor_to_summaries(g, slots).copy())
    want = st.sb.track_table_host(slots, scene_base=0)
    ok = all(all(np.array_equal(o[n], want[n]) for n in want.dtype.names) for o in out)
    q.put((bool(ok), dist.get_backend(), int(want["alive"].sum()), tuple(out[0].shape)))
    st.close()
    dist.destroy_process_group()


def test_rccl_one_rank_all_gather_of_a_real_track_table():
    """SURVEY.md §8(e)'s one collective on the real backend: backend "nccl" (= RCCL) with a world of one -- all the box offers --,
    `ShardedTracker.gather_table(force_collective=True)`: the table k_table writes on the context's own stream, handed to the
    communicator's stream by an event (`mmw_stream_wait`), through `dist.all_gather_into_tensor`, equal to the host read-back."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank, args=(_free_port(), q))
    p.start()
    ok, backend, alive, shape = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert ok and backend == "nccl" and alive >= 24 and shape == (24, 4)
