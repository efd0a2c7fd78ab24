"""Training loop of the MARS regressor (mmwave_msc_amd/train.py; reference src/train.py:28-254) on CPU:
the model in training form, the Keras-layout export into the inference path, the metrics and the loop itself."""
import numpy as np
import pytest
import torch

from mmwave_msc_amd import train as T
from mmwave_msc_amd.mars import MarsCNN
from oracle.mars_np import mars_forward_np


def _toy(n, frames, seed):
    """Feature maps whose keypoints are a smooth function of the input: learnable in a few epochs."""
    rng = np.random.default_rng(seed)
    shape = (n, frames, 8, 8, 5) if frames > 1 else (n, 8, 8, 5)
    x = rng.normal(0.0, 1.0, size=shape).astype(np.float32)
    w = rng.normal(0.0, 0.05, size=(int(np.prod(shape[1:])), 57))
    y = (np.tanh(x.reshape(n, -1) @ w) + 0.5).astype(np.float32)
    return x, y


@pytest.mark.parametrize("frames", [3, 1])
def test_export_matches_inference_path_and_numpy_oracle(frames):
    torch.manual_seed(0)
    net = T.MarsTrainNet(frames)
    x, y = _toy(96, frames, 1)
    T.fit(net, x, y, epochs=2, batch_size=32, seed=3)       # moves the BatchNorm statistics off their initial values
    net.eval()
    with torch.no_grad():
        ref = net(torch.from_numpy(x)).numpy()
    w = T.export_keras_weights(net)
    assert w["conv1_w"].shape == ((3, 3, 3, 5, 16) if frames > 1 else (3, 3, 5, 16))
    assert w["dense1_w"].shape == (frames * 64 * 32, 512 * (3 if frames > 1 else 1))
    inf = T.to_inference(net)                                  # what estimate_posture runs (BatchNorms folded)
    got = inf.predict_numpy(x)
    assert np.max(np.abs(got - ref)) < 2e-4, np.max(np.abs(got - ref))
    orc = mars_forward_np(w, x.astype(np.float64))             # fp64 restatement of the Keras model
    assert np.max(np.abs(orc - ref)) < 2e-4, np.max(np.abs(orc - ref))


def test_fit_reduces_loss_and_reports_history():
    torch.manual_seed(1)
    xtr, ytr = _toy(512, 3, 5)
    xva, yva = xtr[:64], ytr[:64]
    net = T.MarsTrainNet(3)
    before = T.evaluate(net, xtr, ytr)
    h = T.fit(net, xtr, ytr, xva, yva, batch_size=128, epochs=6, seed=0)
    after = T.evaluate(net, xtr, ytr)
    assert len(h["loss"]) == 6 and len(h["val_loss"]) == 6
    assert after["loss"] < 0.8 * before["loss"] and h["loss"][-1] < h["loss"][0], (before, after, h["loss"])
    assert abs(after["rmse"] ** 2 - after["mse"]) < 1e-12 and after["mape"] > 0


def test_keras_semantics_of_the_model():
    net = T.MarsTrainNet(3)
    assert net.bn1.eps == 1e-3 and net.bn2.eps == 1e-3
    assert net.bn1.mom == 0.95 and net.bn2.mom == 0.95   # Keras momentum (train.py:83,88)
    assert float(net.conv1.bias.detach().abs().sum()) == 0.0  # zeros initialiser
    lim = np.sqrt(6.0 / (27 * 5 + 27 * 16))          # glorot_uniform bound of conv1
    assert float(net.conv1.weight.detach().abs().max()) <= lim + 1e-6
    net.eval()                                       # dropout is the identity at inference
    x = torch.randn(4, 3, 8, 8, 5)
    with torch.no_grad():
        assert torch.equal(net(x), net(x))


def test_paper_table_layout():
    rng = np.random.default_rng(0)
    y = rng.normal(size=(200, 57))
    p = y.copy()
    p[:, 0:19] += 0.01     # x error 1 cm on every joint
    p[:, 19:38] -= 0.02    # y error 2 cm
    t = T.paper_table(y, p)
    assert t.shape == (20, 6)
    assert np.allclose(t[:19, 0], 1.0) and np.allclose(t[:19, 1], 1.0)   # x MAE, x RMSE
    assert np.allclose(t[:19, 2], 2.0) and np.allclose(t[:19, 3], 2.0)   # y MAE, y RMSE
    assert np.allclose(t[:, 4:6], 0.0)
    assert np.allclose(t[19], [1.0, 1.0, 2.0, 2.0, 0.0, 0.0])            # average row


def test_cli_round_trip(tmp_path):
    f, l = tmp_path / "feat", tmp_path / "lab"
    f.mkdir(); l.mkdir()
    for name, n, seed in (("training", 160, 1), ("validate", 32, 2), ("testing", 32, 3)):
        x, y = _toy(n, 3, seed)
        np.save(f / f"{name}_mmWave.npy", x)
        np.save(l / f"{name}_labels.npy", y)
    out = tmp_path / "model" / "MARS.npz"
    assert T.main(["--features", str(f), "--labels", str(l), "--out", str(out), "--epochs", "1", "--batch-size", "32", "--device", "cpu"]) == 0
    m = MarsCNN.from_npz(str(out))
    assert m.predict(np.zeros((2, 3, 8, 8, 5), np.float32)).shape == (2, 57)


def test_batchnorm_moving_statistics_follow_keras():
    """Keras: moving = moving * momentum + batch * (1 - momentum), with the BIASED batch variance (torch's own BatchNorm
    accumulates the unbiased one); inference uses gamma * (x - mean) / sqrt(var + 1e-3) + beta."""
    import torch
    from mmwave_msc_amd import train as T
    torch.manual_seed(0)
    bn = T.KerasBatchNorm(4, 1e-3, 0.95)
    x1, x2 = torch.randn(6, 4, 3, 2, 2) * 2 + 1, torch.randn(5, 4, 3, 2, 2) - 3
    bn.train()
    y1 = bn(x1)
    m1, v1 = x1.transpose(0, 1).reshape(4, -1).mean(1), x1.transpose(0, 1).reshape(4, -1).var(1, unbiased=False)
    assert torch.allclose(y1, (x1 - m1.view(1, 4, 1, 1, 1)) / torch.sqrt(v1.view(1, 4, 1, 1, 1) + 1e-3), atol=1e-5)
    bn(x2)
    m2, v2 = x2.transpose(0, 1).reshape(4, -1).mean(1), x2.transpose(0, 1).reshape(4, -1).var(1, unbiased=False)
    assert torch.allclose(bn.running_mean, (0.0 * 0.95 + 0.05 * m1) * 0.95 + 0.05 * m2, atol=1e-6)
    assert torch.allclose(bn.running_var, (1.0 * 0.95 + 0.05 * v1) * 0.95 + 0.05 * v2, atol=1e-6)
    bn.eval()
    z = bn(x1)
    assert torch.allclose(z, (x1 - bn.running_mean.view(1, 4, 1, 1, 1)) / torch.sqrt(bn.running_var.view(1, 4, 1, 1, 1) + 1e-3), atol=1e-5)
