"""The hand-written fused Conv3D x2 (csrc/k_mars.hip, fp32 MFMA) against torch's convolutions and the
fp64 numpy oracle: activations within 1e-4, keypoints within the 1e-4 m tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch", [1, 7, 600])
def test_hip_convs_match_torch_and_oracle(batch):
    import torch
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    from oracle.mars_np import _conv_same, mars_forward_np
    w = random_keras_weights(seed=11, frames=3)
    model = MarsCNN.from_keras_weights(w).to("cuda:0")
    assert model.use_hip_conv
    rng = np.random.default_rng(batch)
    x = rng.normal(0, 0.5, size=(batch, 3, 8, 8, 5)).astype(np.float32)
    x[:, :, 6:, :, :] = 0.0  # zero-padded rows as real feature maps have
    xt = torch.from_numpy(x).to("cuda:0")
    with torch.no_grad():
        got = model._hip_convs(xt).float().cpu().numpy().reshape(batch, 3, 8, 8, 32)
        ref_t = torch.relu(model.conv2(torch.relu(model.conv1(xt.permute(0, 4, 1, 2, 3))))).permute(0, 2, 3, 4, 1).cpu().numpy()
        kp = model(xt).float().cpu().numpy()
    f64 = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
    h = np.maximum(_conv_same(x.astype(np.float64), f64["conv1_w"], f64["conv1_b"]), 0)
    ref = np.maximum(_conv_same(h, f64["conv2_w"], f64["conv2_b"]), 0)
    assert np.abs(got - ref).max() <= 1e-4, np.abs(got - ref).max()
    assert np.abs(got - ref_t).max() <= 1e-4
    assert np.abs(kp - mars_forward_np(w, x)).max() <= 1e-4


@pytest.mark.gpu
def test_trained_weights_drop_into_the_hip_inference_path():
    """mmwave_msc_amd/train.py on the GPU: a few steps of the reference's training loop, then the exported
    Keras-layout tensors through MarsCNN (fused HIP Conv3D kernel + folded BatchNorms) against the trained
    network in eval mode."""
    import numpy as np
    import torch
    from mmwave_msc_amd import train as T
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    x = rng.normal(size=(256, 3, 8, 8, 5)).astype(np.float32)
    y = rng.normal(0.5, 0.3, size=(256, 57)).astype(np.float32)
    net = T.MarsTrainNet(3)
    h = T.fit(net, x, y, x[:32], y[:32], batch_size=64, epochs=3, device="cuda")
    assert h["loss"][-1] < h["loss"][0]
    ref = T.predict(net, x, "cuda")
    inf = T.to_inference(net).to("cuda")
    assert inf.use_hip_conv
    got = inf.predict_numpy(x)
    assert np.max(np.abs(got - ref)) < 1e-3, np.max(np.abs(got - ref))


def test_dense1_split_fp16_holds_the_fp32_tolerance():
    """Dense-1 as three exact fp16 partial products with fp32 accumulation (mars.MarsCNN(arith="f16x3"), the default) against
    the plain fp32 GEMM and the fp64 numpy oracle: both within 1e-4, the split form not worse than fp32 by
    more than a factor two (measured: better), and the split activation of mmw_mars_conv3d_split re-completes to the fp32
    one within 2^-22 relative."""
    import torch
    from mmwave_msc_amd.mars import MarsCNN, SPLIT_SCALE, random_keras_weights
    from oracle.mars_np import mars_forward_np
    w = random_keras_weights(7, 3)
    rng = np.random.default_rng(3)
    feat = rng.normal(0, 0.6, size=(700, 3, 8, 8, 5)).astype(np.float32)
    feat[rng.random(size=(700, 3, 8, 8)) < 0.3] = 0.0
    feat[5] *= 40.0      # large activations: fp16's range is what hi must hold
    feat[6] *= 1e-3      # tiny ones: hi goes subnormal, lo' completes it
    want = mars_forward_np(w, feat.astype(np.float64))
    x = torch.from_numpy(feat).to("cuda:0")
    m16 = MarsCNN.from_keras_weights(w).to("cuda:0")
    m32 = MarsCNN.from_keras_weights(w, arith="f32").to("cuda:0")
    assert m16.arith == "f16x3" and m32.arith == "f32"
    with torch.no_grad():
        k16, k32 = m16(x).double().cpu().numpy(), m32(x).double().cpu().numpy()
        a32 = m32._hip_convs(x)
        a2 = m16._hip_convs_split(x).float()
    scale = np.maximum(1.0, np.abs(want))
    e16, e32 = float((np.abs(k16 - want) / scale).max()), float((np.abs(k32 - want) / scale).max())
    assert e32 <= 1e-4 and e16 <= 1e-4, (e16, e32)
    assert e16 <= 2.0 * e32 + 1e-6, (e16, e32)
    rec = a2[:, :6144] + a2[:, 6144:] / SPLIT_SCALE
    err = (rec - a32).abs()
    assert float((err - a32.abs() * 2.0 ** -21).max()) <= 1e-7, float(err.max())   # 2^-22 relative (+ subnormal floor)
