"""Dataset side (mmwave_msc_amd/dataset.py) against outputs recorded from the reference's
own preprocessing functions on a synthetic experiment (oracle/gen_golden.py: gen_preprocess)."""
import csv
import os

import numpy as np
import pytest

from mmwave_msc_amd import dataset

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "preprocess.npz"))


@pytest.fixture()
def logs(tmp_path, gold):
    off = np.load(os.path.join(GOLD, "offline.npz"))
    mm = tmp_path / "log" / "mmWave" / "A1"
    mm.mkdir(parents=True)
    for k in (1, 2):
        (mm / f"{k}.csv").write_text(str(off[f"csv{k}"]))
    kin = tmp_path / "log" / "kinect"
    kin.mkdir(parents=True)
    (kin / "A1.csv").write_text(str(gold["kinect_in"]))
    return tmp_path, str(mm), str(kin / "A1.csv")


def test_kinect_row_transforms_give_the_reference_text(gold):
    probe = [str(v) for v in gold["probe"]]
    tr = dataset.translate_kinect(list(probe))
    assert tr == [str(v) for v in gold["probe_translated"]]
    assert dataset.static_kinect(list(tr)) == [str(v) for v in gold["probe_static"]]
    assert dataset.relative_kinect(list(tr), [0.25, 2.0]) == [str(v) for v in gold["probe_relative"]]
    assert probe == [str(v) for v in gold["probe"]]  # inputs are not modified


def test_pairing_by_time_stamp(logs, gold):
    _, mm, kin = logs
    pairs = dataset.pair(kin, mm)
    assert np.array_equal(np.array(pairs, dtype=np.int64), gold["pairs"])
    assert len(pairs) < 130  # every seventh frame has no Kinect row within 20 ms


def test_filter_kinect_frames_matches_reference_file(logs, gold, tmp_path):
    _, mm, kin = logs
    pairs = [tuple(p) for p in gold["pairs"]]
    # the reference run found exactly these frames valid: the ones whose rows were written
    kept = {int(r[1]) for r in csv.reader(str(gold["kinect_out"]).splitlines())}
    # reconstruct an invalid list that yields the same selection: every paired frame whose Kinect frame was not kept
    invalid = [p[0] for p in pairs if p[1] not in kept]
    out = tmp_path / "k.csv"
    dataset.filter_kinect_frames(pairs, invalid, kin, str(out))
    assert out.read_bytes().decode() == str(gold["kinect_out"])


def test_npy_formatters_match_reference(gold, tmp_path):
    pre = tmp_path / "pre" / "mmWave" / "training" / "A1"
    pre.mkdir(parents=True)
    for name, txt in zip(gold["pre_files"], gold["pre_txt"]):
        (pre / str(name)).write_text(str(txt))
    kdir = tmp_path / "pre" / "kinect" / "training"
    kdir.mkdir(parents=True)
    (kdir / "A1.csv").write_bytes(str(gold["kinect_out"]).encode())
    fm = dataset.format_mmwave_to_npy(str(pre.parent), str(tmp_path / "training_mmWave.npy"))
    fk = dataset.format_kinect_to_npy(str(kdir), str(tmp_path / "training_labels.npy"))
    assert fm.shape == gold["fmt_mmwave"].shape and fm.dtype == gold["fmt_mmwave"].dtype
    assert np.array_equal(fm, gold["fmt_mmwave"])
    assert np.array_equal(fk, gold["fmt_labels"])
    assert np.array_equal(np.load(tmp_path / "training_mmWave.npy"), gold["fmt_mmwave"])
    # and they are what train.py consumes: (B, 8, 8, 5) features, (B, 57) labels
    assert fm.shape[1:] == (8, 8, 5) and fk.shape[1] == 57 and fm.shape[0] == fk.shape[0]


def test_extract_parts_orders_experiments():
    names = ["B10.csv", "A2.csv", "B2.csv", "A10.csv"]
    assert sorted(names, key=dataset.extract_parts) == ["A2.csv", "B2.csv", "A10.csv", "B10.csv"]


@pytest.mark.gpu
def test_preprocess_experiment_writes_the_reference_files(logs, gold):
    """The whole loop -- OfflineManager, normalize_data, the GPU TrackBuffer (incl. BatchedData.pop_frame between
    shards), relative_coordinates, format_batched_frames, pandas CSV output -- byte for byte."""
    tmp, mm, kin = logs
    out_dir = tmp / "pre" / "mmWave" / "A1"
    out_kin = tmp / "pre" / "kinect"
    out_kin.mkdir(parents=True)
    pairs, invalid, cen = dataset.preprocess_experiment(mm, kin, str(out_dir), str(out_kin / "A1.csv"),
                                                        centroid_npy=str(tmp / "A1_centroid.npy"), max_pts=64)
    assert np.array_equal(np.array(pairs, dtype=np.int64), gold["pairs"])
    files = sorted(os.listdir(out_dir), key=lambda x: int(os.path.splitext(x)[0]))
    assert files == [str(f) for f in gold["pre_files"]]
    for f, txt in zip(files, gold["pre_txt"]):
        assert (out_dir / f).read_text() == str(txt), f
    assert (out_kin / "A1.csv").read_bytes().decode() == str(gold["kinect_out"])
    assert np.array_equal(cen, gold["centroids"])
    assert np.array_equal(np.load(tmp / "A1_centroid.npy"), gold["centroids"])


def test_split_sets_and_add_noise_give_the_reference_tree(tmp_path):
    """split_sets / add_noise (preprocessing.py:406-509) on the synthetic pre-processed tree of tests/golden/splitsets.npz:
    the tree the reference's own functions left (recorded by oracle/gen_golden.py with numpy's global generator seeded),
    file by file and byte for byte."""
    import json
    from mmwave_msc_amd import dataset
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "splitsets.npz"))
    after_split, after_noise = json.loads(str(z["after_split"])), json.loads(str(z["after_noise"]))
    modes = ("training", "validate", "testing")
    for rel, txt in after_split.items():          # the input tree = what is not under a mode directory
        if rel.split(os.sep)[1] in modes:
            continue
        p = tmp_path / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        with open(p, "w", newline="") as fh:
            fh.write(txt)

    def dump():
        out = {}
        for base, _, files in os.walk(tmp_path):
            for f in files:
                q = os.path.join(base, f)
                out[os.path.relpath(q, tmp_path)] = open(q, newline="").read()
        return out

    dataset.split_sets([str(tmp_path / "kinect"), str(tmp_path / "mmWave")], json.loads(str(z["prefixes"])))
    assert dump() == after_split
    np.random.seed(int(z["seed"]))
    dataset.add_noise(str(tmp_path / "mmWave" / "training"), str(tmp_path / "kinect" / "training"))
    got = dump()
    assert sorted(got) == sorted(after_noise)
    assert got == after_noise
