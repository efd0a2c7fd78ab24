"""Second witnesses for the oracle legs nothing in this image can pin against the real third-party code (no Keras, no
MARS.h5): the same published layer semantics evaluated by two independent implementations must agree.

* CNN (train.py:33-106): oracle/mars_np.py (numpy slices) against oracle/mars_torch.py (torch's CPU convolutions, unfolded
  BatchNormalization): <= 1e-12 in fp64; the fp32 evaluation -- Keras' own dtype, the CNN of the CPU baselines -- within 1e-5.
* The reference-shaped Python port of normalize_data and of estimate_posture's feature side (oracle/py_tracker.py, what
  bench.py times as cpu_baseline "port") against oracle/c, which the golden fixtures pin on the reference itself."""
import numpy as np
import pytest


@pytest.mark.parametrize("frames", [3, 1])
def test_cnn_oracle_has_a_second_witness(frames):
    import torch
    from mmwave_msc_amd.mars import random_keras_weights
    from oracle.mars_np import mars_forward_np
    from oracle.mars_torch import MarsTorchCPU
    w = random_keras_weights(11, frames)
    rng = np.random.default_rng(5)
    x = rng.normal(0, 1, (9, 3, 8, 8, 5) if frames == 3 else (9, 8, 8, 5))
    x[:, ..., 4] = rng.gamma(1.0, 1.0, x[..., 4].shape)          # intensity column: skewed, as normalised peakVal is
    x[2] = 0.0                                                    # an all-padding sample
    a = mars_forward_np(w, x)
    b = MarsTorchCPU(w, torch.float64).forward(x)
    c = MarsTorchCPU(w, torch.float32).forward(x.astype(np.float32))
    assert a.shape == b.shape == c.shape == (9, 57)
    assert float(np.abs(a - b).max()) <= 1e-12 * max(1.0, float(np.abs(a).max()))
    assert float((np.abs(a - c) / np.maximum(1.0, np.abs(a))).max()) <= 1e-5


def test_python_port_of_normalize_and_features_matches_the_c_oracle():
    import bench_ingest
    from mmwave_msc_amd.synth import make_scene
    from oracle import c_oracle as co
    from oracle.py_tracker import Params, PyScene, py_feature_maps, py_normalize
    p, c, d = make_scene(31, 14, 160, 2, ragged=True)
    cfg = co.default_config()
    raw = bench_ingest.raw_rows_from_normalised(p, float(cfg.tilt_cos), float(cfg.tilt_sin), float(cfg.s_height))
    raw[3, 0, :3] = 0.0                                           # r == 0 (Utils.py:387-390)
    raw[4, :5, 2] = 3.0                                           # rows above the scene filter's ceiling
    P = Params()
    sc, oc = PyScene(P), co.OracleScene(cfg, 160)
    for f in range(14):
        n = int(c[f])
        det = {k: raw[f, :n, i].astype(np.float64) for i, k in enumerate(("x", "y", "z", "doppler", "peakVal"))}
        a, b = py_normalize(P, det), co.normalize(cfg, raw[f, :n].astype(np.float64))
        assert a.shape == b.shape and np.allclose(a, b, rtol=0, atol=1e-12), f
        sc.track(b, float(d[f]))
        oc.track(b, float(d[f]))
        m, o = py_feature_maps(P, sc)
        of, oo = oc.features()
        assert list(o) == [int(v) for v in oo], f
        if len(o):
            assert np.array_equal(m.astype(np.float32), of), f
    assert sc.n_tracks == oc.n_tracks >= 1
