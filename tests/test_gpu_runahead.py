"""Steps queued AHEAD of the GPU (no host wait between them) with the DBSCAN chain workers on the side stream.

The side stream is paced by the queue protocol alone (no event between the two streams: profiles/NOTEBOOK.md round 4), and a
k_chain launch idles out after ~3 ms.  When the context's stream stalls with steps queued behind the stall -- a long upload, a
caller's own kernel -- the k_chain launches of several steps pass before the first k_track runs; a late one must not claim another
step's pushes with ITS arguments (output buffers, big_live).  Every step here writes into its own output buffers; all of them
must equal the oracle's.  (`make DIAG=nogate DIAGFLAGS=-DMMW_MUTANT_CHAIN_NOGATE` builds the library without the epoch test in
k_chain: MMW_LIB_NAME=libmmw_hip_nogate.so fails this test at the 20 ms stall -- a scene's labels land in the buffers of the
step two ahead -- profiles/NOTEBOOK.md round 5.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("stall_ms", [0, 10, 20, 30, 50, 120])
def test_steps_queued_behind_a_stalled_stream_keep_their_own_outputs(stall_ms):
    import torch
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.synth import make_scene
    from oracle import c_oracle as co
    S, N, F, T = 96, 96, 9, 4
    kw = dict(tr_max_tracks=T, db_min_samples=12, chain_side_stream=1, kalman_dense_min_units=1)
    sb = SceneBatch(_lib.default_config(**kw), S, N)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    sb.follow_torch_stream(st)
    pts = np.zeros((F, S, N, 8), np.float32)
    cnt = np.zeros((F, S), np.int32)
    dts = np.zeros((F, S))
    rng = np.random.default_rng(77)
    for s in range(S):
        # three targets that walk in at different frames: every arrival is a dense cloud of unassigned points two or three
        # frames long -- a small-queue push (and a BallTree chain on a worker) in the frames AFTER the first
        presence = np.ones((F, 3), dtype=bool)
        for j in range(3):
            presence[: int(rng.integers(1, F - 1)), j] = False
        pts[:, s], cnt[:, s], dts[:, s] = make_scene(4200 + s, F, N, 3, ragged=(s % 4 == 0), presence=presence)
    P = torch.from_numpy(pts).to(dev).double()
    C = torch.from_numpy(cnt).to(dev)
    D = torch.from_numpy(dts).to(dev)
    UM = sb.UM
    assoc = torch.full((F, S, N), -9, dtype=torch.int32, device=dev)
    labels = torch.full((F, S, UM), -9, dtype=torch.int32, device=dev)
    dbn = torch.full((F, S), -9, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    # one untimed step so that the side-stream probe has run; then the stall, then every remaining step queued at once
    sb.step_dev(P[0].data_ptr(), C[0].data_ptr(), D[0].data_ptr(), assoc[0].data_ptr(), labels[0].data_ptr(), dbn[0].data_ptr())
    torch.cuda.synchronize()
    if sb.side_workers() != 1:
        pytest.skip("the side stream shares a hardware queue with the context's stream on this box: no chain workers")
    with torch.cuda.stream(st):
        if stall_ms:
            torch.cuda._sleep(int(stall_ms * 2.0e6))   # ~2 GHz: the context's stream is busy while the k_chain launches below idle out
        for f in range(1, F):
            sb.step_dev(P[f].data_ptr(), C[f].data_ptr(), D[f].data_ptr(), assoc[f].data_ptr(), labels[f].data_ptr(), dbn[f].data_ptr())
    torch.cuda.synchronize()
    sb.check()
    A, Lb, Dn = assoc.cpu().numpy(), labels.cpu().numpy(), dbn.cpu().numpy()
    cfg = co.default_config(**{k: v for k, v in kw.items() if k in ("tr_max_tracks", "db_min_samples")})
    n_db = n_cl = n_late = 0
    for s in range(S):
        orc = co.OracleScene(cfg, N)
        for f in range(F):
            c = int(cnt[f, s])
            oa, ol = orc.track(pts[f, s, :c].astype(np.float64), float(dts[f, s]))
            assert np.array_equal(A[f, s, :c], oa), (f, s)
            assert (ol is None) == (Dn[f, s] < 0), (f, s, Dn[f, s])
            if ol is not None:
                n_db += 1
                n_cl += int(ol.max() + 1) if len(ol) else 0
                n_late += int(ol.max() + 1) if (len(ol) and f >= 2) else 0
                assert Dn[f, s] == len(ol) and np.array_equal(Lb[f, s, : len(ol)], ol), (f, s)
    # (the clouds that spawn are the ones a chain worker -- or k_post behind it -- clusters: most of them in the queued steps)
    assert n_db > S and n_cl > S and n_late > S, (n_db, n_cl, n_late)
    assert sb.diag_queue()[4] == 0
    sb.close()
