"""Training of the MARS keypoint regressor on PyTorch-ROCm (first "next" row).

What the reference does (src/train.py): `define_CNN_3D` / `define_CNN` (33-106) -- Conv(16, 3, same, relu) ->
Dropout(0.3) -> Conv(32, 3, same, relu) -> Dropout(0.3) -> BatchNormalization(momentum 0.95) -> Flatten ->
Dense(512 [x3 for the 3-frame model], relu) -> BatchNormalization(momentum 0.95) -> Dropout(0.4) -> Dense(57) --
compiled with MSE loss and Adam(learning_rate 1e-3, beta_1 0.5) (60-66, 98-104), fitted for 150 epochs at batch
size 128 with a validation set (28-30, 130-138), evaluated on a test split with MAE / MSE / MAPE / RMSE and a
per-joint MAE/RMSE table in centimetres (140-232), and saved when the test MAE improves (244-254).

This module is that loop on torch: `MarsTrainNet` is the model in training form (Keras semantics: glorot-uniform
kernels, zero biases, BatchNorm epsilon 1e-3, momentum 0.95 and biased moving variance, channels-last Flatten
order), `fit` / `evaluate` / `paper_table` are the loop and its metrics, and `export_keras_weights` writes the
tensors in Keras layouts -- exactly the `.npz` that `mmwave_msc_amd.mars.MarsCNN.from_npz` (the inference path
`estimate_posture` uses) loads, so trained weights drop into the tracker without conversion.

Runs on any torch device; on an MI355X pass `device="cuda"`.
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .mars import BN_EPS, N_KEYPOINTS, MarsCNN

BATCH_SIZE = 128   # train.py:29
EPOCHS = 150       # train.py:30
KERAS_ADAM_EPS = 1e-7


class KerasBatchNorm(nn.Module):
    """Keras `BatchNormalization(momentum)` over channel axis 1: training normalises with the batch's mean and BIASED
    variance and moves `moving_mean` / `moving_variance` towards exactly those (torch's BatchNorm accumulates the
    unbiased variance instead, so its exported `running_var` would differ from a Keras-trained model's by n/(n-1))."""

    def __init__(self, channels: int, eps: float = BN_EPS, keras_momentum: float = 0.95):
        super().__init__()
        self.eps, self.mom = float(eps), float(keras_momentum)
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shape = [1, -1] + [1] * (x.dim() - 2)
        if self.training:
            dims = [d for d in range(x.dim()) if d != 1]
            mean = x.mean(dim=dims)
            var = x.var(dim=dims, unbiased=False)
            with torch.no_grad():
                self.running_mean.mul_(self.mom).add_(mean.detach(), alpha=1.0 - self.mom)
                self.running_var.mul_(self.mom).add_(var.detach(), alpha=1.0 - self.mom)
        else:
            mean, var = self.running_mean, self.running_var
        return (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + self.eps) * self.weight.view(shape) + self.bias.view(shape)


class MarsTrainNet(nn.Module):
    """define_CNN_3D (frames = 3, input (B,3,8,8,5)) or define_CNN (frames = 1, input (B,8,8,5)), trainable."""

    def __init__(self, frames: int = 3, n_keypoints: int = N_KEYPOINTS):
        super().__init__()
        self.frames = int(frames)
        self.three_d = self.frames > 1
        conv = nn.Conv3d if self.three_d else nn.Conv2d
        self.conv1 = conv(5, 16, 3, padding=1)
        self.conv2 = conv(16, 32, 3, padding=1)
        self.bn1 = KerasBatchNorm(32, BN_EPS, 0.95)          # train.py:45,83
        flat = (self.frames if self.three_d else 1) * 64 * 32
        hidden = 512 * (3 if self.three_d else 1)
        self.dense1 = nn.Linear(flat, hidden)
        self.bn2 = KerasBatchNorm(hidden, BN_EPS, 0.95)
        self.dense2 = nn.Linear(hidden, n_keypoints)
        for m in (self.conv1, self.conv2, self.dense1, self.dense2):  # Keras defaults: glorot_uniform, zeros
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x channels-last like the Keras model's input: (B,3,8,8,5) or (B,8,8,5)."""
        if self.three_d:
            h = x.permute(0, 4, 1, 2, 3)
        else:
            h = x.permute(0, 3, 1, 2)
        h = F.dropout(F.relu(self.conv1(h)), 0.3, self.training)
        h = F.dropout(F.relu(self.conv2(h)), 0.3, self.training)
        h = self.bn1(h)
        # Keras Flatten on a channels-last tensor: (d,h,w,c) order
        h = h.permute(0, 2, 3, 4, 1) if self.three_d else h.permute(0, 2, 3, 1)
        h = F.relu(self.dense1(h.flatten(1)))
        h = F.dropout(self.bn2(h), 0.4, self.training)
        return self.dense2(h)


def export_keras_weights(net: MarsTrainNet) -> dict:
    """The model's tensors in Keras layouts, keyed as `mars.MarsCNN.from_keras_weights` expects."""
    perm = (2, 3, 4, 1, 0) if net.three_d else (2, 3, 1, 0)   # (out,in,k...) -> (k...,in,out)
    g = lambda t: t.detach().cpu().numpy().astype(np.float32)
    return {
        "conv1_w": g(net.conv1.weight.permute(perm)), "conv1_b": g(net.conv1.bias),
        "conv2_w": g(net.conv2.weight.permute(perm)), "conv2_b": g(net.conv2.bias),
        "bn1_gamma": g(net.bn1.weight), "bn1_beta": g(net.bn1.bias), "bn1_mean": g(net.bn1.running_mean), "bn1_var": g(net.bn1.running_var),
        "dense1_w": g(net.dense1.weight.t()), "dense1_b": g(net.dense1.bias),
        "bn2_gamma": g(net.bn2.weight), "bn2_beta": g(net.bn2.bias), "bn2_mean": g(net.bn2.running_mean), "bn2_var": g(net.bn2.running_var),
        "dense2_w": g(net.dense2.weight.t()), "dense2_b": g(net.dense2.bias),
    }


def save_npz(net: MarsTrainNet, path: str) -> None:
    """`keypoint_model.save(...)` (train.py:252) in the format the inference path loads."""
    np.savez(path, **export_keras_weights(net))


def to_inference(net: MarsTrainNet) -> MarsCNN:
    return MarsCNN.from_keras_weights(export_keras_weights(net))


@torch.no_grad()
def predict(net: nn.Module, x: np.ndarray, device=None, batch_size: int = 1024) -> np.ndarray:
    dev = torch.device(device) if device is not None else next(net.parameters()).device
    net.eval()
    out = []
    for i in range(0, len(x), batch_size):
        out.append(net(torch.from_numpy(np.ascontiguousarray(x[i:i + batch_size], dtype=np.float32)).to(dev)).float().cpu().numpy())
    return np.concatenate(out) if out else np.zeros((0, N_KEYPOINTS), np.float32)


def evaluate(net: nn.Module, x: np.ndarray, y: np.ndarray, device=None) -> dict:
    """`model.evaluate`: the compiled loss and metrics (train.py:101-104) over the whole set."""
    p = predict(net, x, device).astype(np.float64)
    y = np.asarray(y, dtype=np.float64)
    err = p - y
    mse = float(np.mean(err * err))
    return {"loss": mse, "mae": float(np.mean(np.abs(err))), "mse": mse,
            "mape": float(100.0 * np.mean(np.abs(err) / np.maximum(np.abs(y), 1e-7))),   # Keras clips |y| at epsilon
            "rmse": float(np.sqrt(mse))}


def fit(net: MarsTrainNet, x_train, y_train, x_val=None, y_val=None, batch_size: int = BATCH_SIZE, epochs: int = EPOCHS,
        device="cpu", seed: int = 0, verbose: bool = False) -> dict:
    """`keypoint_model.fit(...)` (train.py:130-138): Adam(1e-3, beta_1 0.5), MSE, reshuffled every epoch."""
    dev = torch.device(device)
    net.to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.5, 0.999), eps=KERAS_ADAM_EPS)
    xt = torch.from_numpy(np.ascontiguousarray(x_train, dtype=np.float32)).to(dev)
    yt = torch.from_numpy(np.ascontiguousarray(y_train, dtype=np.float32)).to(dev)
    gen = torch.Generator(device="cpu").manual_seed(int(seed))
    hist = {"loss": [], "mae": [], "val_loss": [], "val_mae": []}
    n = xt.shape[0]
    for ep in range(int(epochs)):
        net.train()
        order = torch.randperm(n, generator=gen).to(dev)
        tot = tot_abs = 0.0
        for i in range(0, n, batch_size):
            idx = order[i:i + batch_size]
            if idx.numel() < 2:   # BatchNorm needs more than one sample per batch
                continue
            out = net(xt[idx])
            loss = F.mse_loss(out, yt[idx])
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            tot += float(loss.detach()) * idx.numel()
            tot_abs += float((out.detach() - yt[idx]).abs().mean()) * idx.numel()
        hist["loss"].append(tot / n)
        hist["mae"].append(tot_abs / n)
        if x_val is not None:
            ev = evaluate(net, x_val, y_val, dev)
            hist["val_loss"].append(ev["loss"])
            hist["val_mae"].append(ev["mae"])
        if verbose:
            print(f"epoch {ep + 1}/{epochs}  loss {hist['loss'][-1]:.5f}  mae {hist['mae'][-1]:.5f}"
                  + (f"  val_loss {hist['val_loss'][-1]:.5f}  val_mae {hist['val_mae'][-1]:.5f}" if x_val is not None else ""))
    return hist


def paper_table(labels: np.ndarray, preds: np.ndarray) -> np.ndarray:
    """Per-joint errors in centimetres, the table train.py:160-232 builds: 19 joint rows + the average row,
    columns x-MAE, x-RMSE, y-MAE, y-RMSE, z-MAE, z-RMSE (labels are 19 x, 19 y, 19 z)."""
    labels = np.asarray(labels, dtype=np.float64)
    preds = np.asarray(preds, dtype=np.float64)
    err = preds - labels
    mae = np.mean(np.abs(err), axis=0).reshape(3, 19)            # rows x, y, z
    rmse = np.sqrt(np.mean(err * err, axis=0)).reshape(3, 19)
    per_joint = np.concatenate((mae.T, rmse.T), axis=1) * 100.0   # [19][xmae ymae zmae xrmse yrmse zrmse]
    avg = np.concatenate((mae.mean(axis=1)[None, :], rmse.mean(axis=1)[None, :]), axis=1) * 100.0
    table = np.around(np.concatenate((per_joint, avg), axis=0), 2)
    return table[:, [0, 3, 1, 4, 2, 5]]


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="Train the MARS keypoint CNN (reference src/train.py) on PyTorch-ROCm")
    ap.add_argument("--features", required=True, help="directory with training_mmWave.npy, validate_mmWave.npy, testing_mmWave.npy")
    ap.add_argument("--labels", required=True, help="directory with training_labels.npy, validate_labels.npy, testing_labels.npy")
    ap.add_argument("--out", default="model/MARS.npz")
    ap.add_argument("--epochs", type=int, default=EPOCHS)
    ap.add_argument("--batch-size", type=int, default=BATCH_SIZE)
    ap.add_argument("--folds", type=int, default=0,
                    help="the reference trains 10 dataset folds, formatted/mmWave/{i} and formatted/kinect/{i} (train.py:110-124): with "
                         "--folds K, --features and --labels name the parents of the sub-directories 0..K-1; 0 = the two directories themselves")
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    a = ap.parse_args(argv)
    ld = lambda d, f: np.load(os.path.join(d, f))
    tables = []
    for fold in range(max(a.folds, 1)):
        fdir = os.path.join(a.features, str(fold)) if a.folds > 0 else a.features
        ldir = os.path.join(a.labels, str(fold)) if a.folds > 0 else a.labels
        xtr, xva, xte = ld(fdir, "training_mmWave.npy"), ld(fdir, "validate_mmWave.npy"), ld(fdir, "testing_mmWave.npy")
        ytr, yva, yte = ld(ldir, "training_labels.npy"), ld(ldir, "validate_labels.npy"), ld(ldir, "testing_labels.npy")
        frames = xtr.shape[1] if xtr.ndim == 5 else 1
        net = MarsTrainNet(frames, ytr.shape[1])
        score_min = 10.0   # train.py:128: re-set inside the loop, so EVERY fold whose test MAE is below 10 overwrites the file
        fit(net, xtr, ytr, xva, yva, a.batch_size, a.epochs, a.device, seed=fold, verbose=True)
        tr, te = evaluate(net, xtr, ytr, a.device), evaluate(net, xte, yte, a.device)
        print("train MAPE = ", tr["mape"])
        print("test MAPE = ", te["mape"])
        tables.append(paper_table(yte, predict(net, xte, a.device)))
        print(tables[-1])
        if te["mae"] < score_min:   # train.py:251-254
            os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
            save_npz(net, a.out)
            score_min = te["mae"]
    if len(tables) > 1:   # train.py:257-258: the mean table over the folds
        print(np.mean(tables, axis=0))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
