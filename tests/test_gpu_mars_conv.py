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


def _inputs(n, frames, seed):
    rng = np.random.default_rng(seed)
    shape = (n, 3, 8, 8, 5) if frames == 3 else (n, 8, 8, 5)
    feat = rng.normal(0, 0.6, size=shape).astype(np.float32)
    feat[rng.random(size=shape[:-1]) < 0.3] = 0.0
    feat[5] *= 40.0      # large activations: fp16's range is what hi must hold
    feat[6] *= 1e-3      # tiny ones: hi goes subnormal, lo' completes it
    return feat


@pytest.mark.parametrize("frames,n", [(3, 700), (3, 5), (1, 900), (1, 3)])
def test_split_fp16_cnn_holds_the_fp32_tolerance(frames, n):
    """The conv pair and Dense-1 as three exact fp16 partial products per term with fp32 accumulation
    (mars.MarsCNN(arith="f16x3"), the default; k_mars_conv16 serves define_CNN_3D and define_CNN) against the fp64 numpy
    oracle: keypoints within 1e-4 and not worse than twice the fp32 arithmetic's error (measured: better);
    the split activation re-completes to the fp64 convolution within a few 2^-22."""
    import torch
    from mmwave_msc_amd.mars import MarsCNN, SPLIT_SCALE, deinterleave_split, random_keras_weights
    from oracle.mars_np import _conv_same, mars_forward_np
    w = random_keras_weights(7, frames)
    feat = _inputs(max(n, 8), frames, 3)[:max(n, 8)]
    want = mars_forward_np(w, feat.astype(np.float64))
    x = torch.from_numpy(feat).to("cuda:0")
    m16 = MarsCNN.from_keras_weights(w).to("cuda:0")
    # (the fp32 comparison: the fp32 matrix-core kernel for the 3-frame model; the single-frame model has none, and torch's
    #  convolutions are only ever run on request)
    m32 = MarsCNN.from_keras_weights(w, arith="f32" if frames == 3 else "torch").to("cuda:0")
    assert m16.arith == "f16x3" and m32.arith == m32.fp32_arith() and m16.use_hip_conv
    if frames == 1:
        with pytest.raises(ValueError):
            m16(x[:n], arith="f32")           # no silent hand-over to torch
        with pytest.raises(TypeError):
            m16(x[:n].double())
    with torch.no_grad():
        k16, k32 = m16(x[:n]).double().cpu().numpy(), m32(x[:n]).double().cpu().numpy()
        a_hi, a_lo = (t.double().cpu().numpy() for t in deinterleave_split(m16._hip_convs_split(x)))
    scale = np.maximum(1.0, np.abs(want[:n]))
    e16, e32 = float((np.abs(k16 - want[:n]) / scale).max()), float((np.abs(k32 - want[:n]) / scale).max())
    assert e32 <= 1e-4 and e16 <= 1e-4, (e16, e32)
    assert e16 <= 2.0 * e32 + 2e-6, (e16, e32)
    f64 = {k: np.asarray(v, dtype=np.float64) for k, v in w.items()}
    h = np.maximum(_conv_same(feat.astype(np.float64), f64["conv1_w"], f64["conv1_b"]), 0.0)
    h = np.maximum(_conv_same(h, f64["conv2_w"], f64["conv2_b"]), 0.0).reshape(len(feat), -1)
    rec = a_hi + a_lo / SPLIT_SCALE
    err = np.abs(rec - h) / np.maximum(1.0, np.abs(h))
    assert float(err.max()) <= 2e-5, float(err.max())
    assert float(np.abs(a_lo).max()) <= 1.0 + float(np.abs(h).max())   # lo' stays within hi's magnitude: no overflow of the scaled half
    assert not m16.range_overflow()


@pytest.mark.parametrize("frames,n", [(3, 700), (3, 256), (3, 8500), (1, 1000), (1, 3)])   # 8500 rows x 1536: a whole wave of 256 x 192 tiles + half tiles
def test_fused_dense1_kernel_vs_two_gemms_and_fp64(frames, n):
    """Dense-1 of the split arithmetic as ONE kernel (mmw_mars_dense1_split, csrc/k_dense.hip: train.py:49,87 with BN folded in)
    against the two-GEMM formulation it replaces (hi.W_hi and [hi | lo'].[W_lo' ; W_hi] through torch / hipBLASLt) and against
    fp64 numpy on the SAME split operands: every partial product is exact in all three, so they differ by fp32 summation order
    only; rows past the batch in the last 256-row tile are never read back, and a NaN row stays in its own output row."""
    import torch
    from mmwave_msc_amd.mars import MarsCNN, SPLIT_SCALE, deinterleave_split, random_keras_weights
    w = random_keras_weights(11, frames)
    feat = _inputs(max(n, 8), frames, 5)[:n]
    x = torch.from_numpy(feat).to("cuda:0")
    mk = MarsCNN.from_keras_weights(w).to("cuda:0")
    with torch.no_grad():
        a2 = mk._hip_convs_split(x)
        hk = mk._dense1_split(a2)
        hi, lo = deinterleave_split(a2)
        w_hi, w_lo = (t.t() for t in deinterleave_split(mk.d1_w2_t))          # (K, N)
        bias = mk.dense1_dhwc.bias
        g1 = torch.addmm(bias, hi, w_hi, out_dtype=torch.float32)
        hg = torch.relu(torch.addmm(g1, torch.cat([hi, lo], 1), torch.cat([w_lo, w_hi], 0), out_dtype=torch.float32, alpha=1.0 / SPLIT_SCALE))
        want = torch.relu(bias.double() + hi.double() @ w_hi.double() + (hi.double() @ w_lo.double() + lo.double() @ w_hi.double()) / SPLIT_SCALE)
    assert hk.shape == (n, mk.d1_w2_t.shape[0]) and hk.dtype == torch.float32
    scale = want.abs().clamp(min=1.0)
    ek, eg = float(((hk.double() - want).abs() / scale).max()), float(((hg.double() - want).abs() / scale).max())
    # (fp32 accumulation over K = 2048 / 6144 terms with cancellation: ~1e-5 of the result in either formulation)
    assert ek <= 5e-5 and ek <= 2.0 * eg + 5e-7, (ek, eg)
    # a poisoned row: NaN in, NaN out, in that row only (a compact copy: _dense1_split pads it itself)
    a3 = a2.clone()
    a3[1, 7] = float("nan")
    with torch.no_grad():
        hp = mk._dense1_split(a3)
    assert bool(torch.isnan(hp[1]).all()) and not bool(torch.isnan(hp[0]).any()) and (n < 3 or torch.equal(hp[2:], hk[2:]))


def test_cnn_kernels_repeat_bit_for_bit_under_load():
    """The pipelined Dense-1 tile kernel and the conv kernel hand data between LDS-DMA requests, fragment reads and MFMAs on COUNTED
    waits (csrc/k_dense.hip, k_mars.hip): a count off by one would read a tile that has not landed -- rarely, and only when the memory
    system is slow.  The same batch 60 times on one stream while a second stream keeps the fabric busy with copies, a full chip of tiles
    (31 744 rows) and a ragged one (8500): every repetition equals the first bit for bit (conv output, Dense-1 output, keypoints)."""
    import torch
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    dev = torch.device("cuda:0")
    mk = MarsCNN.from_keras_weights(random_keras_weights(5, 3)).to(dev)
    noise_a = torch.empty(64 << 20, dtype=torch.float32, device=dev).normal_()
    noise_b = torch.empty_like(noise_a)
    side = torch.cuda.Stream(device=dev)
    for n in (31744, 8500):
        x = torch.from_numpy(_inputs(512, 3, 21)).to(dev).repeat((n + 511) // 512, 1, 1, 1, 1)[:n].contiguous()
        with torch.no_grad():
            a0 = mk._hip_convs_split(x)
            h0 = mk._dense1_split(a0)
            k0 = mk(x)
            torch.cuda.synchronize()
            for rep in range(60):
                with torch.cuda.stream(side):
                    for _ in range(4):
                        noise_b.copy_(noise_a)
                a1 = mk._hip_convs_split(x)
                h1 = mk._dense1_split(a1)
                k1 = mk(x)
                assert torch.equal(a1, a0) and torch.equal(h1, h0) and torch.equal(k1, k0), (n, rep)
            torch.cuda.synchronize()


def test_split_arithmetic_knows_fp16s_range():
    """fp16 holds |a| < 65 504: a weight beyond it sends the whole model to the fp32 kernels at load time; an input or an
    activation beyond it is noted PER SAMPLE by the conv kernel, and exactly those samples are computed again in fp32 on the
    device behind Dense-2 (mmw_mars_range_fixup: no host wait, so the pipelined estimate_posture path is covered too) --
    within the usual 1e-4 of the fp64 oracle, the other samples untouched.  More than 64 such samples in one batch, or the
    single-frame model (no fp32 kernel), are reported instead and the synchronous Keras-style entry recomputes the batch.
    train.py:33-106 (Keras fp32) has no such limit."""
    import torch
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    from oracle.mars_np import mars_forward_np
    w = random_keras_weights(3, 3)
    feat = _inputs(64, 3, 9)
    feat[11] *= 1e5                      # inputs of magnitude ~3e5: hi would be inf
    want = mars_forward_np(w, feat.astype(np.float64))
    m = MarsCNN.from_keras_weights(w).to("cuda:0")
    assert m.arith == "f16x3" and m.arith_fallback is None
    x = torch.from_numpy(feat).to("cuda:0")
    scale = np.maximum(1.0, np.abs(want))
    # the 3-frame model repairs itself on the device: the one sample is recomputed in fp32 behind Dense-2 (mmw_mars_range_fixup),
    # without a host wait -- the pipelined estimate_posture path gets finite keypoints where Keras' fp32 does
    with torch.no_grad():
        raw = m(x).float().cpu().numpy()
    assert float((np.abs(raw - want) / scale).max()) <= 1e-4
    assert not m.range_overflow() and m.range_recomputed == 1        # nothing left meaningless; the repair ran (word read and cleared)
    with torch.no_grad():
        again = m(x[:40]).float().cpu().numpy()                       # the flags were consumed: a second batch is repaired on its own
    assert float((np.abs(again - want[:40]) / scale[:40]).max()) <= 1e-4 and not m.range_overflow()
    clean = _inputs(64, 3, 12)
    with torch.no_grad():
        kc = m(torch.from_numpy(clean).to("cuda:0")).float().cpu().numpy()
    assert float((np.abs(kc - mars_forward_np(w, clean.astype(np.float64))) / np.maximum(1.0, np.abs(mars_forward_np(w, clean.astype(np.float64))))).max()) <= 1e-4
    assert not m.range_overflow() and m.range_recomputed == 2         # (the second read above counted the [:40] batch's repair)
    got = m.predict(feat)
    assert m.range_fallbacks == 0 and float((np.abs(got - want) / scale).max()) <= 1e-4
    # more such samples than the fix-up holds in one batch (64): the surplus stays meaningless, the model says so, and the
    # synchronous Keras-style entry -- TrackBuffer.estimate_posture's model.predict, Tracking.py:732 -- computes the batch again in fp32
    many = _inputs(200, 3, 13)
    many[::2] *= 1e5
    want_m = mars_forward_np(w, many.astype(np.float64))
    with torch.no_grad():
        m(torch.from_numpy(many).to("cuda:0"))
    assert m.range_overflow()
    got_m = m.predict(many)
    assert m.range_fallbacks == 1
    # (keypoints of magnitude 1e5 out of cancelling terms: the error is measured against each sample's largest output)
    assert float((np.abs(got_m - want_m) / np.maximum(1.0, np.abs(want_m).max(axis=1, keepdims=True))).max()) <= 1e-4
    # the single-frame model has no fp32 kernel to repair with: it reports, predict() recomputes through torch's kernels
    w1f = random_keras_weights(3, 1)
    f1 = _inputs(16, 1, 9)
    f1[3] *= 1e5
    m1 = MarsCNN.from_keras_weights(w1f).to("cuda:0")
    with torch.no_grad():
        m1(torch.from_numpy(f1).to("cuda:0"))
    assert m1.range_overflow() and not m1.range_overflow()
    want1 = mars_forward_np(w1f, f1.astype(np.float64))
    got1 = m1.predict(f1)
    assert m1.range_fallbacks == 1 and float((np.abs(got1 - want1) / np.maximum(1.0, np.abs(want1))).max()) <= 1e-4
    # a huge BN-folded Dense-1 row (tiny moving variance x large gamma): fp32 from the start
    w2 = dict(w)
    w2["bn1_gamma"] = w["bn1_gamma"].copy(); w2["bn1_var"] = w["bn1_var"].copy()
    w2["bn1_gamma"][4] = 3.0e6; w2["bn1_var"][4] = 1e-9
    m2 = MarsCNN.from_keras_weights(w2).to("cuda:0")
    assert m2.arith == "f32" and "fp16" in m2.arith_fallback
    small = _inputs(32, 3, 10)
    want2 = mars_forward_np(w2, small.astype(np.float64))
    got2 = m2.predict(small)
    # (keypoints of magnitude 1e8 out of cancelling 1e9 terms: the error is measured against each sample's largest output)
    e2 = float((np.abs(got2 - want2) / np.maximum(1.0, np.abs(want2).max(axis=1, keepdims=True))).max())
    assert e2 <= 1e-4 and m2.range_fallbacks == 0, (e2, float(np.abs(want2).max()))


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 13, 64])
def test_small_batch_head_matches_the_fp64_oracle(n):
    """MarsCNN.forward_small (mmw_mars_conv3d + mmw_mars_head_small: Dense-1 + ReLU + Dense-2 as thin fp32 kernels over the
    whole chip, the drop-in's estimate_posture path; train.py:71-106) against the fp64 numpy oracle within 1e-4, and against
    the tile kernels' result for the same batch."""
    import torch
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    from oracle.mars_np import mars_forward_np
    w = random_keras_weights(21, 3)
    feat = _inputs(max(n, 8), 3, 9)[:n]
    want = mars_forward_np(w, feat.astype(np.float64))
    m = MarsCNN.from_keras_weights(w).to("cuda:0")
    assert m.has_small_path()
    x = torch.from_numpy(feat).to("cuda:0")
    with torch.no_grad():
        got = m.forward_small(x).double().cpu().numpy()
        big = m(x).double().cpu().numpy()
    scale = np.maximum(1.0, np.abs(want))
    assert got.shape == (n, 57)
    assert float((np.abs(got - want) / scale).max()) <= 1e-4
    assert float((np.abs(got - big) / scale).max()) <= 1e-4      # (two arithmetics: Keras' fp32 here, the split-fp16 tiles there)
    with pytest.raises(ValueError):
        m.forward_small(torch.zeros((65, 3, 8, 8, 5), device="cuda:0"))
