"""The h5py-free Keras `.h5` weight reader against files written by h5py
(oracle/gen_h5_fixture.py, committed under tests/golden/)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from mmwave_msc_amd import h5weights

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KEYS = ["conv1_w", "conv1_b", "conv2_w", "conv2_b", "bn1_gamma", "bn1_beta", "bn1_mean", "bn1_var", "dense1_w", "dense1_b",
        "bn2_gamma", "bn2_beta", "bn2_mean", "bn2_var", "dense2_w", "dense2_b"]


def test_model_save_layout_with_running_layer_counters():
    """/model_weights/<layer>/<layer>/<var>:0, layers named conv3d_18/conv3d_19/..., ten members per group (two
    symbol-table leaves), optimizer state with an int64 scalar, JSON attributes in continuation blocks."""
    z = np.load(os.path.join(GOLD, "keras_like.npz"))
    w = h5weights.load_keras_h5(os.path.join(GOLD, "keras_like_save.h5"))
    assert sorted(w) == sorted(KEYS)
    for k in KEYS:
        assert w[k].dtype == np.float32 and w[k].shape == z["save_" + k].shape
        assert np.array_equal(w[k], z["save_" + k]), k
    assert w["conv1_w"].shape == (3, 3, 3, 5, 16)


def test_save_weights_layout_2d_model():
    z = np.load(os.path.join(GOLD, "keras_like.npz"))
    w = h5weights.load_keras_h5(os.path.join(GOLD, "keras_like_weights.h5"))
    for k in KEYS:
        assert np.array_equal(w[k], z["weights_" + k]), k
    assert w["conv1_w"].shape == (3, 3, 5, 16)


def test_every_dataset_is_listed_by_path():
    ds = h5weights.read_h5_datasets(os.path.join(GOLD, "keras_like_save.h5"))
    assert "/model_weights/conv3d_18/conv3d_18/kernel:0" in ds
    assert "/model_weights/batch_normalization_19/batch_normalization_19/moving_variance:0" in ds
    assert "/optimizer_weights/Adam/m/kernel:0" in ds
    assert "/optimizer_weights/Adam/iter:0" not in ds  # int64: not a weight tensor
    assert not any("dropout" in k or "flatten" in k for k in ds)  # weight-less layers are empty groups
    assert len(ds) == 16 + 2


def test_unsupported_container_is_refused_not_misread(tmp_path):
    with pytest.raises(h5weights.H5FormatError, match="superblock version"):
        h5weights.load_keras_h5(os.path.join(GOLD, "keras_like_latest.h5"))
    p = tmp_path / "not.h5"
    p.write_bytes(b"PK\x03\x04" + bytes(2000))
    with pytest.raises(h5weights.H5FormatError, match="not an HDF5 file"):
        h5weights.read_h5_datasets(str(p))


def test_wrong_model_is_named(tmp_path):
    # a file with only one dense layer: the error says what was found
    src = os.path.join(GOLD, "keras_like_weights.h5")
    ds = h5weights.read_h5_datasets(src)
    assert len(ds) == 16
    import mmwave_msc_amd.h5weights as H
    orig = H.read_h5_datasets
    try:
        H.read_h5_datasets = lambda path: {k: v for k, v in orig(path).items() if "/dense_1/" not in k}
        with pytest.raises(H.H5FormatError, match="2 dense"):
            H.load_keras_h5(src)
    finally:
        H.read_h5_datasets = orig


def test_full_size_file_feeds_the_inference_model(tmp_path):
    """A MARS-sized `.h5` (38 MB) written by h5py when an interpreter with h5py is around (this image: conda's
    python3.9); `MarsCNN.from_h5` must give the model `from_keras_weights` gives for the same tensors."""
    py = next((p for p in ("/opt/conda/bin/python3.9", shutil.which("python3.9") or "") if p and os.path.exists(p)), None)
    if py is None or subprocess.run([py, "-c", "import h5py"], capture_output=True).returncode != 0:
        pytest.skip("no interpreter with h5py to write the full-size file")
    gen = os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle", "gen_h5_fixture.py")
    path = str(tmp_path / "MARS.h5")
    subprocess.run([py, gen, "--full", path], check=True)
    import torch
    from mmwave_msc_amd.mars import MarsCNN
    z = np.load(path + ".npz")
    a = MarsCNN.load(path)
    b = MarsCNN.from_keras_weights({k: z[k] for k in z.files})
    assert a.frames == 3
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert na == nb and torch.equal(pa, pb), na
    for (na, pa), (nb, pb) in zip(a.named_buffers(), b.named_buffers()):
        assert na == nb and torch.equal(pa, pb), na
