#!/usr/bin/env python3
"""Writes the HDF5 fixtures of tests/test_h5weights.py with h5py -- TEST INFRASTRUCTURE, run by hand with an
interpreter that has h5py (this image: /opt/conda/bin/python3.9 oracle/gen_h5_fixture.py); the product never imports
h5py.  The files mimic what Keras 2.x writes for the reference's model (train.py:33-106, saved at train.py:252):

  tests/golden/keras_like_save.h5      `model.save` layout: /model_weights/<layer>/<layer>/<var>:0, layer names with
                                       the running per-class counters a 10-run training script produces, weight-less
                                       layers (dropout, flatten) as empty groups, /optimizer_weights with Adam state
                                       (incl. an int64 scalar), string attributes (model_config JSON, layer_names, ...)
  tests/golden/keras_like_weights.h5   `model.save_weights` layout: layers at the root, 2-D convolutions
  tests/golden/keras_like_latest.h5    the same tensors written with libver='latest' (superblock v3, OHDR object
                                       headers): the reader must refuse it with a message, not mis-read it
  tests/golden/keras_like.npz          the tensors of the first two files under the keys load_keras_h5 returns

Tensors are tiny (the reader does not care about shapes); values are seeded.  A full-size file is produced by
`full_size(path)` for the test that feeds MarsCNN.from_h5 (not committed: 38 MB).
"""
import json
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")


def tensors(rng, three_d=True, small=True, frames=3):
    k = (3, 3, 3) if three_d else (3, 3)
    flat = (frames if three_d else 1) * 64 * 32
    hidden = 512 * (3 if three_d else 1)
    if small:
        flat, hidden = 24, 10
    f = lambda *sh: rng.standard_normal(sh).astype(np.float32)
    return {
        "conv1_w": f(*k, 5, 16), "conv1_b": f(16), "conv2_w": f(*k, 16, 32), "conv2_b": f(32),
        "bn1_gamma": f(32), "bn1_beta": f(32), "bn1_mean": f(32), "bn1_var": np.abs(f(32)) + 0.5,
        "dense1_w": f(flat, hidden), "dense1_b": f(hidden),
        "bn2_gamma": f(hidden), "bn2_beta": f(hidden), "bn2_mean": f(hidden), "bn2_var": np.abs(f(hidden)) + 0.5,
        "dense2_w": f(hidden, 57), "dense2_b": f(57),
    }


def write_layers(g, w, names):
    conv1, drop1, conv2, drop2, bn1, flat, dense1, bn2, drop3, dense2 = names
    order = [conv1, drop1, conv2, drop2, bn1, flat, dense1, bn2, drop3, dense2]
    g.attrs["layer_names"] = np.array([n.encode() for n in order])
    g.attrs["backend"] = b"tensorflow"
    g.attrs["keras_version"] = b"2.15.0"

    def layer(name, items):
        lg = g.create_group(name)
        lg.attrs["weight_names"] = np.array([f"{name}/{v}:0".encode() for v, _ in items])
        for v, arr in items:
            lg.create_dataset(f"{name}/{v}:0", data=arr)

    layer(conv1, [("kernel", w["conv1_w"]), ("bias", w["conv1_b"])])
    layer(drop1, [])
    layer(conv2, [("kernel", w["conv2_w"]), ("bias", w["conv2_b"])])
    layer(drop2, [])
    layer(bn1, [("gamma", w["bn1_gamma"]), ("beta", w["bn1_beta"]), ("moving_mean", w["bn1_mean"]), ("moving_variance", w["bn1_var"])])
    layer(flat, [])
    layer(dense1, [("kernel", w["dense1_w"]), ("bias", w["dense1_b"])])
    layer(bn2, [("gamma", w["bn2_gamma"]), ("beta", w["bn2_beta"]), ("moving_mean", w["bn2_mean"]), ("moving_variance", w["bn2_var"])])
    layer(drop3, [])
    layer(dense2, [("kernel", w["dense2_w"]), ("bias", w["dense2_b"])])


NAMES_3D = ["conv3d_18", "dropout_27", "conv3d_19", "dropout_28", "batch_normalization_18", "flatten_9", "dense_18",
            "batch_normalization_19", "dropout_29", "dense_19"]
NAMES_2D = ["conv2d", "dropout", "conv2d_1", "dropout_1", "batch_normalization", "flatten", "dense", "batch_normalization_1",
            "dropout_2", "dense_1"]


def save_layout(path, w, names, rng, **kw):
    with h5py.File(path, "w", **kw) as f:
        f.attrs["keras_version"] = "2.15.0"
        f.attrs["backend"] = "tensorflow"
        f.attrs["model_config"] = json.dumps({"class_name": "Sequential", "config": {"name": "sequential_9", "layers": [
            {"class_name": n.rstrip("_0123456789"), "config": {"name": n, "trainable": True, "dtype": "float32"}} for n in names]}})
        f.attrs["training_config"] = json.dumps({"loss": "mse", "metrics": ["mae", "mse", "mape", "RootMeanSquaredError"],
                                                 "optimizer_config": {"class_name": "Adam", "config": {"learning_rate": 0.001, "beta_1": 0.5}}})
        write_layers(f.create_group("model_weights"), w, names)
        og = f.create_group("optimizer_weights")
        og.attrs["weight_names"] = np.array([b"Adam/iter:0", b"Adam/m/kernel:0", b"Adam/v/kernel:0"])
        ag = og.create_group("Adam")
        ag.create_dataset("iter:0", data=np.int64(1234))
        ag.create_dataset("m/kernel:0", data=rng.standard_normal(w["conv1_w"].shape).astype(np.float32))
        ag.create_dataset("v/kernel:0", data=rng.standard_normal(w["conv1_w"].shape).astype(np.float32))


def full_size(path, seed=7):
    rng = np.random.default_rng(seed)
    w = tensors(rng, True, small=False)
    save_layout(path, w, NAMES_3D, rng)
    return w


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--full":
        w = full_size(sys.argv[2])
        np.savez(sys.argv[2] + ".npz", **w)
        return
    rng = np.random.default_rng(20240914)
    w3 = tensors(rng, True)
    w2 = tensors(rng, False)
    save_layout(os.path.join(GOLD, "keras_like_save.h5"), w3, NAMES_3D, rng)
    with h5py.File(os.path.join(GOLD, "keras_like_weights.h5"), "w") as f:
        write_layers(f, w2, NAMES_2D)
    save_layout(os.path.join(GOLD, "keras_like_latest.h5"), w3, NAMES_3D, rng, libver="latest")
    np.savez(os.path.join(GOLD, "keras_like.npz"), **{"save_" + k: v for k, v in w3.items()}, **{"weights_" + k: v for k, v in w2.items()})
    for n in ("keras_like_save.h5", "keras_like_weights.h5", "keras_like_latest.h5", "keras_like.npz"):
        print(n, os.path.getsize(os.path.join(GOLD, n)), "bytes")


if __name__ == "__main__":
    main()
