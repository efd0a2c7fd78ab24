"""Host-side logic that needs no GPU: the CSV frame iterator (incl. the reference's
refill quirk), the config surface, scene sharding, the synthetic generator."""
import os

import numpy as np

from tests._golden import GOLDEN


def test_offline_manager_matches_reference_sequence(tmp_path):
    from mmwave_msc_amd import constants as const
    from mmwave_msc_amd.utils import OfflineManager
    z = np.load(os.path.join(GOLDEN, "offline.npz"))
    (tmp_path / "1.csv").write_text(str(z["csv1"]))
    (tmp_path / "2.csv").write_text(str(z["csv2"]))
    assert const.FB_READ_BUFFER_SIZE == 40
    man = OfflineManager(str(tmp_path))
    seq = []
    while not man.is_finished():
        ok, fc, det = man.get_data()
        if ok:   # the five columns as float64 arrays (the reference: lists; same keys, same values), posix as the reference has it
            assert all(isinstance(det[k], np.ndarray) and det[k].dtype == np.float64 and det[k].shape == det["x"].shape
                       for k in ("x", "y", "z", "doppler", "peakVal")) and len(det["posix"]) == len(det["x"])
        seq.append((1, fc, len(det["x"]), det["posix"][0]) if ok else (0, fc, 0, 0))
    want = z["seq"]
    assert len(seq) == len(want)
    assert np.array_equal(np.array(seq, dtype=np.int64), want)
    # the quirk: frames 40, 79, 118 arrive with exactly one point
    assert [s[1] for s in seq if s[0] and s[2] == 1] == [40, 79, 118]


def test_constants_surface_and_motion_models():
    from mmwave_msc_amd import constants as const
    for name in ("FB_FRAMES_BATCH", "DB_EPS", "DB_Z_WEIGHT", "DB_RANGE_WEIGHT", "DB_MIN_SAMPLES_MIN", "TR_MAX_TRACKS",
                 "TR_LIFETIME_DYNAMIC", "TR_LIFETIME_STATIC", "TR_VEL_THRES", "TR_GATE", "KF_Q_STD", "KF_P_INIT",
                 "KF_GROUP_DISP_EST_INIT", "KF_ENABLE_EST", "KF_A_N", "KF_EST_POINTNUM", "KF_SPREAD_LIM", "KF_A_SPR",
                 "INTENSITY_MU", "INTENSITY_STD", "MODEL_MIN_INPUT", "MODEL_DEFAULT_POSTURE", "MOTION_MODEL",
                 "S_HEIGHT", "S_TILT", "P_MODEL_PATH", "FB_READ_BUFFER_SIZE"):
        assert hasattr(const, name), name
    assert const.MODEL_DEFAULT_POSTURE.shape == (57,)
    f = const.CONST_ACC_MODEL.KF_F(0.1)
    assert f.shape == (9, 9) and f[0, 3] == 0.1 and f[0, 6] == 0.5 * 0.1**2 and f[3, 6] == 0.1 and f[6, 6] == 1
    q = const.CONST_ACC_MODEL.KF_Q_DISCR(0.1)
    assert q.shape == (9, 9) and q[0, 1] == 0.5 * 0.1**3 and q[0, 3] == 0 and q[8, 8] == 1
    assert const.CONST_VEL_MODEL.KF_F(0.2)[2, 5] == 0.2 and const.CONST_VEL_MODEL.KF_Q_DISCR(0.2).shape == (6, 6)
    cfg = const.to_config(tr_max_tracks=8)
    assert cfg.tr_max_tracks == 8 and cfg.dim_x == 9 and cfg.fb_frames_batch == 2


def test_constants_match_reference_values():
    """Container-only cross-check against the real constants.py."""
    import pytest
    from oracle.ref_import import have_reference, load_reference
    if not have_reference():
        pytest.skip("reference not present")
    from mmwave_msc_amd import constants as mine
    ref, _, _ = load_reference()
    for name in dir(ref):
        if name.isupper() and name not in ("MOTION_MODEL", "CONST_ACC_MODEL", "CONST_VEL_MODEL", "SCREEN_CONNECTED"):
            a, b = getattr(ref, name), getattr(mine, name)
            assert np.array_equal(np.asarray(a), np.asarray(b)), name
    for dt in (0.1, 0.37, 1.2):
        for m in ("CONST_ACC_MODEL", "CONST_VEL_MODEL"):
            assert np.array_equal(getattr(ref, m).KF_F(dt), getattr(mine, m).KF_F(dt))
            assert np.array_equal(getattr(ref, m).KF_Q_DISCR(dt), getattr(mine, m).KF_Q_DISCR(dt))
            assert np.array_equal(getattr(ref, m).KF_H, getattr(mine, m).KF_H)


def test_shard_range_partitions_scenes():
    from mmwave_msc_amd.dist import shard_range
    for total, world in [(4096, 8), (10, 3), (7, 8), (256, 1)]:
        blocks = [shard_range(total, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == total
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
        sizes = [b[1] - b[0] for b in blocks]
        assert max(sizes) - min(sizes) <= 1


def test_synth_is_deterministic_and_fp32_exact():
    from mmwave_msc_amd.synth import make_batch, make_scene
    a, ca, da = make_scene(11, 5, 128, 3)
    b, cb, db = make_scene(11, 5, 128, 3)
    assert np.array_equal(a, b) and a.dtype == np.float32 and a.shape == (5, 128, 8)
    assert np.all(a[..., 2] > 0) and np.all(a[..., 2] <= 2.5) and np.all(a[..., 1] > 0)
    p, c, d = make_batch([1, 2, 3], 4, 64, 2, ragged=True)
    assert p.shape == (4, 3, 64, 8) and c.shape == (4, 3) and np.all(c <= 64) and np.all(c >= 32)
    for f in range(4):
        for s in range(3):
            assert np.all(p[f, s, c[f, s]:] == 0)


def test_dataset_formatters_match_reference_goldens():
    """format_batched_frames / format_single_frame_mode (Utils.py:523-574) against outputs recorded from the
    reference (tests/golden/formatters.npz, oracle/gen_golden.py): pure numpy, bit-equal."""
    import os
    import numpy as np
    from mmwave_msc_amd.utils import format_batched_frames, format_single_frame_mode
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "formatters.npz"), allow_pickle=True)
    for case in range(4):
        sizes = g[f"c{case}_n"]
        flat = g[f"c{case}_in"]
        frames, o = [], 0
        for n in sizes:
            frames.append(flat[o:o + n].copy()); o += n
        blk = format_batched_frames(frames)
        assert np.array_equal(blk, g[f"c{case}_block"]), case
        for bs in (1, 2, 3):
            b = blk.copy()
            got = format_single_frame_mode(b, 93.0, 40.0, bs, fuse=False)
            assert np.array_equal(got, g[f"c{case}_mode{bs}"]), (case, bs)
            assert not np.array_equal(b[:, 4], blk[:, 4]) or np.all(blk[:, 4] == (blk[:, 4] - 93.0) / 40.0)  # normalised in place
            assert np.array_equal(format_single_frame_mode(blk.copy(), 93.0, 40.0, bs, fuse=True), g[f"c{case}_fuse{bs}"]), (case, bs)


def test_projection_and_fade_square():
    """calc_projection_points against values recorded from the reference (Utils.py:180-219); calc_fade_square
    (Visualizer.py:14-29) against values recorded from the reference's function, scalar and table forms agreeing."""
    import os
    import types
    import numpy as np
    from mmwave_msc_amd import constants as const
    from mmwave_msc_amd.utils import calc_fade_square, calc_projection_points, fade_squares
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "formatters.npz"), allow_pickle=True)
    for row, want in zip(g["proj_in"], g["proj_out"]):
        assert np.array_equal(np.array(calc_projection_points(*row)), want), row
    xs, zs = calc_projection_points(g["proj_in"][:, 0], g["proj_in"][:, 1], g["proj_in"][:, 2])
    assert np.array_equal(np.stack([xs, zs], axis=1), g["proj_out"])
    # calc_fade_square as the reference computes it (recorded by compiling the function out of Visualizer.py)
    # (fp64 throughout, like the reference under its pinned numpy 1.26: bit-equal to the recording made with widened
    #  keypoints; numpy 2.x keeps one subtraction of the float32 keypoint in float32 -> 1e-6 on that recording)
    fx, fk, want, want32 = g["fade_x"], g["fade_kp"], g["fade_out64"], g["fade_out"]
    px, pz, size = fade_squares(fx, fk)
    assert np.array_equal(np.stack([px, pz, size], axis=1), want)
    assert np.abs(np.stack([px, pz, size], axis=1) - want32).max() <= 1e-6
    for t in range(len(fx)):
        tr = types.SimpleNamespace(state=types.SimpleNamespace(x=fx[t].reshape(9, 1)), keypoints=fk[t])
        (cx, cz), sz = calc_fade_square(tr)
        assert [cx, cz, sz] == want[t].tolist(), t
    assert want[0, 2] == const.V_SCREEN_FADE_SIZE_MIN and want[1, 2] == const.V_SCREEN_FADE_SIZE_MAX
    rng = np.random.default_rng(5)
    sx = rng.uniform(-2, 6, size=(12, 9))
    kp = rng.normal(0, 0.5, size=(12, 57))
    px, pz, size = fade_squares(sx, kp)
    for t in range(12):
        tr = types.SimpleNamespace(state=types.SimpleNamespace(x=sx[t]), keypoints=kp[t])
        (cx, cz), sz = calc_fade_square(tr)
        assert cx == px[t] and cz == pz[t] and sz == size[t]
        assert const.V_SCREEN_FADE_SIZE_MIN <= sz <= const.V_SCREEN_FADE_SIZE_MAX
