"""Dataset side of the keypoint model: from logged experiments to the `.npy` files
`train.py` fits on.  Counterpart of the reference's src/preprocessing.py, which is a script (it runs on import, reads
its paths from constants and needs the recorded logs); here the same steps are functions over explicit paths, and the
tracker inside `preprocess_experiment` is the GPU `TrackBuffer`.

  pair                     mmWave frame <-> Kinect frame by time stamp, 20 ms window      preprocessing.py:27-49
  translate_kinect,
  relative_kinect,
  static_kinect            skeleton rows into the radar's frame / relative to the track    preprocessing.py:90-145
  filter_kinect_frames     the Kinect rows that have a valid mmWave frame                  preprocessing.py:52-87
  preprocess_experiment    one pass of preprocess_dataset()'s loop body                    preprocessing.py:148-275
  format_mmwave_to_npy,
  format_kinect_to_npy     pre-processed CSVs -> `<mode>_mmWave.npy` / `<mode>_labels.npy`  preprocessing.py:298-384
  split_sets               experiments into training / validate / testing by name prefix    preprocessing.py:406-468
  add_noise                jittered copies `N_<experiment>` of the training experiments     preprocessing.py:471-509

Text output is written the way the reference writes it (pandas `to_csv`, `csv.writer`, `str(float)`), so that files
compare byte for byte (tests/test_dataset.py: outputs recorded from the reference's own functions on a synthetic
experiment).
"""
from __future__ import annotations

import csv
import os
import shutil

import numpy as np
import pandas as pd

from . import constants as const
from .utils import (OfflineManager, format_batched_frames, format_single_frame_mode, normalize_data,
                    relative_coordinates)

KINECT_Z = 0.8     # preprocessing.py:22-24
KINECT_X = 0.22
RELATIVE_ENABLED = True
KINECT_TILT_DEG = 6.5
PAIR_WINDOW_MS = 20


def _shards(mmwave_dir):
    return sorted(os.listdir(mmwave_dir), key=lambda x: int(os.path.splitext(x)[0]))


def pair(kinect_csv: str, mmwave_dir: str):
    """[(mmWave frame number, Kinect frame number)] for every mmWave frame that has a Kinect row within 20 ms
    (column 6 of the mmWave log and column 0 of the Kinect log are posix milliseconds)."""
    kin = pd.read_csv(kinect_csv, header=None)
    kt = kin[0].to_numpy().astype(np.float64)
    out = []
    for name in _shards(mmwave_dir):
        df = pd.read_csv(os.path.join(mmwave_dir, name), header=None)
        first_rows = df.drop_duplicates(subset=0)
        for frame, stamp in zip(first_rows[0].to_numpy(), first_rows[6].to_numpy()):
            k = int(np.abs(kt - float(stamp)).argsort()[:1][0])
            if abs(stamp - kt[k]) < PAIR_WINDOW_MS:
                out.append((int(frame), kin.iloc[k, 1]))
    return out


def _joint_view(row):
    """The 19 x (x, z, y) block of a Kinect row (fields 2 .. len-2; the reference walks them by index mod 3)."""
    n = (len(row) - 3) // 3
    vals = np.array([float(v) for v in row[2:2 + 3 * n]], dtype=np.float64).reshape(n, 3)
    return n, vals


def _put_back(row, n, vals):
    out = list(row)
    flat = vals.reshape(-1)
    for k in range(3 * n):
        out[2 + k] = str(flat[k])
    return out


def translate_kinect(row):
    """Kinect camera frame -> radar frame: shift x, rotate (z, y) by the camera tilt, lift by the camera height."""
    a = np.radians(KINECT_TILT_DEG)
    n, v = _joint_view(row)
    x, z, y = v[:, 0], v[:, 1], v[:, 2]
    out = np.empty_like(v)
    out[:, 0] = x + KINECT_X
    out[:, 1] = y * np.sin(a) + z * np.cos(a) + KINECT_Z
    out[:, 2] = y * np.cos(a) - z * np.sin(a)
    res = _put_back(row, n, out)
    # the reference's index walk stops one field early when the row length makes the last triple incomplete:
    # fields it never touches keep their text
    for i in range(2 + 3 * n, len(row)):
        res[i] = row[i]
    return res


def relative_kinect(row, centroid):
    n, v = _joint_view(row)
    out = v.copy()
    out[:, 0] = v[:, 0] - centroid[0]
    out[:, 2] = v[:, 2] - centroid[1]
    res = _put_back(row, n, out)
    for k in range(n):  # z fields are not rewritten: they keep their text
        res[2 + 3 * k + 1] = row[2 + 3 * k + 1]
    return res


def static_kinect(row):
    """Lower back on x = 0 / y = 0 (fields 2 and 13), the lower foot on z = 0 (fields 39, 51)."""
    x_abs, y_abs = float(row[2]), float(row[13])
    z_abs = min(float(row[39]), float(row[51]))
    n, v = _joint_view(row)
    out = v - np.array([x_abs, z_abs, y_abs])
    return _put_back(row, n, out)


def filter_kinect_frames(pairs, invalid_frames, kinect_csv: str, out_csv: str):
    """Keeps the Kinect rows paired with a valid mmWave frame, translated (and, with RELATIVE_ENABLED, made static)."""
    invalid = set(invalid_frames)
    paired = {p[1] for p in pairs}
    dropped = {p[1] for p in pairs if p[0] in invalid}
    with open(kinect_csv, "r", newline="") as fin, open(out_csv, "w", newline="") as fout:
        wr = csv.writer(fout)
        for row in csv.reader(fin):
            k = int(row[1])
            if k in paired and k not in dropped:
                t = translate_kinect(row)
                if RELATIVE_ENABLED:
                    t = static_kinect(t)
                wr.writerow(t)


def preprocess_experiment(mmwave_dir: str, kinect_csv: str, out_dir: str, out_kinect_csv: str, centroid_npy=None,
                          device: int = 0, max_pts: int = 512):
    """The body of preprocess_dataset() for one experiment: replay the log through the tracker, and for every frame
    that is paired with a Kinect frame and whose first track was just updated (lifetime 0) save that track's
    three-frame cloud -- relative to its centroid -- as 192 rows (frame, x, y, z, doppler, intensity).  Returns
    (pairs, invalid frame numbers, centroids)."""
    from .tracking import BatchedData, TrackBuffer
    pairs = pair(kinect_csv, mmwave_dir)
    paired = {p[0] for p in pairs}
    if os.path.exists(out_dir):
        shutil.rmtree(out_dir)
    os.makedirs(out_dir)
    pending = pd.DataFrame()
    in_file, file_no = 0, 1
    cur = os.path.join(out_dir, f"{file_no}.csv")
    centroids, invalid = [], []
    tb, batch = TrackBuffer(max_pts=max_pts, device=device), BatchedData()
    man = OfflineManager(mmwave_dir)
    first = True
    while not man.is_finished():
        ok, frame, det = man.get_data()
        valid = False
        if frame in paired:
            if ok:
                tb.dt = 0.1 if first else det["posix"][0] / 1000 - tb.t
                first = False
                tb.t = det["posix"][0] / 1000
                eff = normalize_data(det)
                if eff.shape[0] != 0:
                    tb.track(eff, batch)
                    tracks = tb.effective_tracks
                    if len(tracks) > 0 and tracks[0].lifetime == 0 and len(tracks[0].batch.effective_data) > 0:
                        valid = True
                        frames = list(tracks[0].batch.buffer)
                        if RELATIVE_ENABLED:
                            frames = relative_coordinates(frames, tracks[0].cluster.centroid)
                            centroids.append(tracks[0].cluster.centroid[:2])
                        block = format_batched_frames(frames)
                        pending = pd.concat([pending, pd.DataFrame({
                            "Frame": frame, "X": block[:, 0], "Y": block[:, 1], "Z": block[:, 2], "Doppler": block[:, 3],
                            "Intensity": block[:, 4]})], ignore_index=True)
                        in_file += 1
                        if len(pending) >= const.FB_WRITE_BUFFER_SIZE or in_file >= const.FB_EXPERIMENT_FILE_SIZE:
                            pending.to_csv(cur, mode="a", index=False, header=False)
                            pending = pending.iloc[0:0]
                            if in_file >= const.FB_EXPERIMENT_FILE_SIZE:
                                in_file = 0
                                file_no += 1
                                cur = os.path.join(out_dir, f"{file_no}.csv")
            else:
                batch.pop_frame()
        if not valid:
            invalid.append(frame)
    pd.DataFrame(pending).to_csv(cur, mode="a", index=False, header=False)
    tb.close()
    cen = np.array(centroids)
    if centroid_npy is not None:
        np.save(centroid_npy, cen)
    filter_kinect_frames(pairs, invalid, kinect_csv, out_kinect_csv)
    return pairs, invalid, cen


def extract_parts(filename):
    """('B12.csv' -> (12, 'B', '.csv')): the sort key of experiment names."""
    base, ext = os.path.splitext(filename)
    return int("".join(c for c in base if c.isdigit())), "".join(c for c in base if c.isalpha()), ext


def format_mmwave_to_npy(experiments_directory: str, out_file: str, mean=const.INTENSITY_MU, std_dev=const.INTENSITY_STD,
                         batch_size: int = 1, fuse: bool = True) -> np.ndarray:
    """Every saved 192-row block of every experiment (name order of `extract_parts`, shards by number) as one CNN
    input (`format_single_frame_mode`), stacked and saved."""
    blocks = []
    for exp in sorted(os.listdir(experiments_directory), key=extract_parts):
        path = os.path.join(experiments_directory, exp)
        for name in _shards(path):
            with open(os.path.join(path, name), "r") as fh:
                rows = list(csv.reader(fh))
            start = 0
            for i in range(1, len(rows) + 1):
                if i == len(rows) or int(rows[i][0]) != int(rows[start][0]):
                    if i > start:
                        arr = np.array([[float(v) for v in r[1:6]] for r in rows[start:i]], dtype=np.float32)
                        blocks.append(format_single_frame_mode(arr, mean, std_dev, batch_size, fuse))
                    start = i
    out = np.array(blocks)
    np.save(out_file, out)
    return out


def format_kinect_to_npy(kinect_directory: str, out_file: str) -> np.ndarray:
    """Labels: per Kinect row the 19 joints as 19 x, 19 z-field, 19 y-field values (fields 2..58, transposed)."""
    rows = []
    for exp in sorted(os.listdir(kinect_directory), key=extract_parts):
        frames = pd.read_csv(os.path.join(kinect_directory, exp), header=None)
        for _, fr in frames.iterrows():
            rows.append(np.array(fr[2:59]).reshape(-1, 3).T.flatten())
    out = np.array(rows)
    np.save(out_file, out)
    return out


def split_sets(directories, prefixes) -> None:
    """preprocessing.py:406-468: under each of `directories` (the pre-processed Kinect and mmWave trees) copy every
    experiment -- a directory of CSV shards or a single CSV file -- into `validate/`, `testing/` or `training/`: the first
    list of `prefixes` names the validation experiments, the second the test experiments (substring match on the name, as the
    reference's `str.find`), everything else trains.  Existing mode directories are replaced (the reference removes them
    first and fails when one is missing; here a missing one is simply created)."""
    validate_prefix, testing_prefix = prefixes[0], prefixes[1]
    modes = ("training", "validate", "testing")
    for directory in directories:
        for mode in modes:
            shutil.rmtree(os.path.join(directory, mode), ignore_errors=True)
    for directory in directories:
        experiments = os.listdir(directory)
        for mode in modes:
            os.makedirs(os.path.join(directory, mode))
        for experiment in experiments:
            if any(experiment.find(mode) != -1 for mode in modes):
                continue
            source = os.path.join(directory, experiment)
            if any(experiment.find(prefix) != -1 for prefix in validate_prefix):
                mode = "validate"
            elif any(experiment.find(prefix) != -1 for prefix in testing_prefix):
                mode = "testing"
            else:
                mode = "training"
            target = os.path.join(directory, mode, experiment)
            if os.path.isdir(source):
                shutil.copytree(source, target)
            else:
                shutil.copy(source, target)


def add_noise(mmwave_training_dir: str, kinect_training_dir: str, mean: float = 0.0, std: float = 0.022, rng=None) -> None:
    """preprocessing.py:471-509: for every training experiment a copy `N_<experiment>` whose mmWave rows have Gaussian noise
    (sigma 2.2 cm) added to the non-zero x, y, z fields -- columns 1..3 of the pre-processed CSVs --, one draw per value in file
    and row order; the Kinect labels of the copy are the original's.  `rng` = anything with `normal(loc=, scale=)`; the default
    is numpy's global generator, which the reference uses (so `np.random.seed(s)` reproduces its files byte for byte)."""
    rng = np.random if rng is None else rng
    for experiment in os.listdir(mmwave_training_dir):
        input_path = os.path.join(mmwave_training_dir, experiment)
        distorted_path = os.path.join(mmwave_training_dir, f"N_{experiment}")
        if os.path.exists(distorted_path):
            shutil.rmtree(distorted_path)
        os.mkdir(distorted_path)
        for filename in os.listdir(input_path):
            with open(os.path.join(input_path, filename), "r") as fh:
                rows = list(csv.reader(fh))
            for row in rows:
                for i in range(1, 4):
                    if float(row[i]) != 0:
                        row[i] = str(float(row[i]) + rng.normal(loc=mean, scale=std))
            with open(os.path.join(distorted_path, filename), "w", newline="") as fh:
                csv.writer(fh).writerows(rows)
    for experiment in os.listdir(kinect_training_dir):
        shutil.copyfile(os.path.join(kinect_training_dir, experiment), os.path.join(kinect_training_dir, f"N_{experiment}"))
