"""BASELINE.json configs[2] at its FULL size on one GPU, on SURVEY.md §8(d)'s literal population: 4096 scenes x 512
points, K = T = 8 targets in EVERY scene (`bench.generate(population="full")`), the context left on its automatic
layout (track-wise Kalman kernels, side-stream DBSCAN workers -- what `bench.py`'s `full_tracks` leg times).

Every frame's association vector and DBSCAN call pattern and the final track state of ALL 4096 scenes are compared
with the C oracle (oracle/c, OpenMP over scenes; Tracking.py:664-703), bit for bit.  `bench.py` makes the same check
inside its run; this is the `-m gpu` test of it.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_full_track_population_4096x512x8_vs_oracle():
    import torch
    import bench
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from oracle import c_oracle as co

    S, N, T, F = 4096, 512, 8, 14
    cores = bench.effective_cores()
    pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=min(cores, 16), population="full")
    # the oracle, frame by frame (its association vectors are compared per frame)
    ob = co.OracleBatch(co.default_config(tr_max_tracks=T), S, N)
    threads = max(1, min(cores, co.max_threads()))
    dev = torch.device("cuda:0")
    sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
    assert sb.step_kind() == 4 and sb.kalman_layout() == 1
    st = torch.cuda.Stream(device=dev)
    most = 0
    with torch.cuda.stream(st):
        sb.follow_torch_stream(st)
        d_cnt = torch.from_numpy(cnt).to(dev)
        d_dt = torch.from_numpy(dts).to(dev)
        a = torch.empty((S, N), dtype=torch.int32, device=dev)
        lab = torch.empty((S, sb.UM), dtype=torch.int32, device=dev)
        dbn = torch.empty((S,), dtype=torch.int32, device=dev)
        for f in range(F):
            p = torch.from_numpy(pts[f]).to(dev).double()
            sb.step_dev(p.data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr(), a.data_ptr(), lab.data_ptr(), dbn.data_ptr())
            oa, ol, od = ob.step(pts[f].astype(np.float64), cnt[f], dts[f], threads=threads)
            st.synchronize()
            assert np.array_equal(dbn.cpu().numpy(), od), f
            assert np.array_equal(a.cpu().numpy(), oa), f
            if f < 4:   # the start-up frames: every scene clusters its ring (labels of all 4096 clouds)
                got = lab.cpu().numpy()
                for s in range(0, S, 5):
                    if od[s] > 0:
                        assert np.array_equal(got[s, : od[s]], ol[s, : od[s]]), (f, s)
    assert sb.side_workers() in (0, 1)
    sb.check()
    ntr = sb.num_tracks()
    trk = sb.tracks(cap=max(int(ntr.max()), 1))
    for s in range(S):
        want = ob.scenes[s].tracks()
        assert ntr[s] == len(want), s
        got = trk[s, : ntr[s]]
        for name in ("x", "P", "centroid", "min_vals", "max_vals", "spread_est", "group_disp_est", "n_est", "lifetime", "point_num",
                     "is_static", "ring_len", "ring_n"):
            assert np.array_equal(got[name], want[name]), (s, name)
        most = max(most, int(ntr[s]))
    assert most >= T and float(ntr.mean()) > 0.9 * T   # K = T: (nearly) every scene tracks all its targets (a frame of several new clusters may exceed TR_MAX_TRACKS, Tracking.py:576-589)
    ln, rn = sb.batch_ring()
    for s in range(0, S, 3):
        assert np.array_equal(rn[s, : ln[s]], ob.scenes[s].batch_ring()), s
    sb.close()
