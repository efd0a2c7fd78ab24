"""Soak of the queue protocol between the association kernel and the DBSCAN chain workers (k_track pushes, k_chain claims on a
side stream, k_post / k_dbscan_big take what is left and wait for the claimed items; csrc/k_dbscan.hip): 300 frames of a
4096-scene context whose targets disappear for four seconds and come back, so that tracks expire (Tracking.py:513-528), scenes
without tracks cluster their whole ring again (apply_DBscan on clouds of up to 1536 points, Utils.py:250-291) and new tracks are
spawned all along -- large and small clouds in every frame, not just in the start-up frames of the short tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

S_DATA, REP, N, T, F = 512, 8, 512, 8, 300


def _scene(args):
    from mmwave_msc_amd.synth import make_scene
    sid, = args
    k = 1 + sid % T
    presence = np.ones((F, k), dtype=bool)
    for j in range(k):
        a = 60 + 10 * j + (sid % 7)
        presence[a: a + 40, j] = False          # 4 s: past TR_LIFETIME_DYNAMIC (3 s) -- the track expires, the target returns
        presence[190 + 5 * j: 230 + 5 * j, j] = False
    return make_scene(sid, F, N, k, presence=presence)


def test_soak_queue_protocol_300_frames_4096_scenes_vs_oracle():
    import multiprocessing as mp
    import torch
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from oracle import c_oracle as co

    with mp.get_context("spawn").Pool(8) as pool:      # (spawn: this process may have initialised the GPU in an earlier test)
        res = pool.map(_scene, [(s,) for s in range(S_DATA)], chunksize=8)
    pts = np.stack([r[0] for r in res], axis=1)       # [F, S_DATA, N, 8] float32
    cnt = np.stack([r[1] for r in res], axis=1)
    dts = np.stack([r[2] for r in res], axis=1)
    n_chk = 96                                        # scenes run through the C oracle
    ob = co.OracleBatch(co.default_config(tr_max_tracks=T), n_chk, N)
    co.batch_run_f32(ob, np.ascontiguousarray(pts[:, :n_chk]), np.ascontiguousarray(cnt[:, :n_chk]), np.ascontiguousarray(dts[:, :n_chk]), 0)

    S = S_DATA * REP
    dev = torch.device("cuda:0")
    sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N, device=0)   # 4096 scenes: the side-stream workers are on by default
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        sb.follow_torch_stream(st)
        d_cnt = torch.from_numpy(cnt).to(dev).repeat(1, REP)
        d_dt = torch.from_numpy(dts).to(dev).repeat(1, REP)
        a = torch.empty((S, N), dtype=torch.int32, device=dev)
        lab = torch.empty((S, sb.UM), dtype=torch.int32, device=dev)
        dbn = torch.empty((S,), dtype=torch.int32, device=dev)
        big = 0
        for f in range(F):
            p = torch.from_numpy(pts[f]).to(dev).double().repeat(REP, 1, 1)
            sb.step_dev(p.data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr(), a.data_ptr(), lab.data_ptr(), dbn.data_ptr())
            if f % 50 == 49:
                big += int((dbn > 256).sum().item())   # (also keeps `p` alive until the step has run)
        st.synchronize()
    assert sb.step_kind() == 4 and sb.side_workers() in (0, 1)
    sb.check()                                         # a worker that gave up a wait it must not give up is a sticky error
    q = sb.diag_queue()
    assert int(q[4]) == 0, q
    assert big > 0                                     # clouds of more than 256 points did recur in the sampled frames
    ntr = sb.num_tracks()
    trk = sb.tracks(cap=max(int(ntr.max()), 1))
    for s in range(n_chk):
        want = ob.scenes[s].tracks()
        assert ntr[s] == len(want), s
        got = trk[s, : ntr[s]]
        for name in ("x", "P", "centroid", "spread_est", "group_disp_est", "lifetime", "point_num", "is_static", "ring_n"):
            assert np.array_equal(got[name], want[name]), (s, name)
    # the eight replicas of every data scene went through different workgroups, queue positions and workers: same state
    for r in range(1, REP):
        assert np.array_equal(ntr[:S_DATA], ntr[r * S_DATA: (r + 1) * S_DATA])
        for s in range(0, S_DATA, 37):
            assert trk[s, : ntr[s]].tobytes() == trk[r * S_DATA + s, : ntr[s]].tobytes(), (r, s)
    sb.close()
