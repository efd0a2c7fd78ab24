"""Reference-shaped point-cloud utilities (`src/Utils.py` hot subset) on the GPU:

    normalize_data(detObj)            Utils.py:342-434   -> mmw_normalize
    apply_DBscan(pointcloud, ...)     Utils.py:250-291   -> mmw_dbscan (BallTree-faithful)
    format_single_frame(track_cloud)  Utils.py:468-520   -> mmw_format_frames
    relative_coordinates(...)         Utils.py:437-465   (host glue: a subtraction per frame)
    OfflineManager                    Utils.py:53-177    (CSV frame iterator incl. its 40-frame quirk)
    RingBuffer                        Utils.py:10-50

The numeric functions need libmmw_hip and a gfx950 device; there is no CPU path.
"""
from __future__ import annotations

import csv
import os
from collections import deque

import numpy as np

from . import constants as const
from .batch import SceneBatch

_UTIL_MAX_PTS = 1024  # MMW_MAX_PTS_LIMIT; with ring 2: DBSCAN clouds up to 2048 points, ring 4 (dbscan_labels on larger ones): 4096
_ctx_cache = {}


def _util_ctx(**cfg_over) -> SceneBatch:
    """A 1-scene context used by the stateless helpers (cached per config)."""
    key = tuple(sorted(cfg_over.items()))
    sb = _ctx_cache.get(key)
    if sb is None:
        kw = dict(fb_frames_batch=1)
        kw.update(cfg_over)
        ring = kw["fb_frames_batch"] + 1
        sb = SceneBatch(const.to_config(**kw), 1, min(_UTIL_MAX_PTS, 4096 // ring))
        _ctx_cache[key] = sb
    return sb


class RingBuffer:
    """Fixed-size FIFO (reference Utils.py:10-50)."""

    def __init__(self, size, init_val=None):
        self.size = size
        self.buffer = deque(maxlen=size)
        self.append(0 if init_val is None else init_val)

    def append(self, item):
        self.buffer.append(item)

    def get_max(self):
        return np.max(self.buffer)

    def get_mean(self):
        return np.mean(self.buffer)


class OfflineManager:
    """Frame iterator over logged CSV shards `<path>/<k>.csv` (k = 1, 2, ...), one row
    per detected point: frame, x, y, z, doppler, peakVal, posix_ms (DataLogging.py:60-82).

    Behaviour kept from the reference (Utils.py:86-166), including its refill quirk: a
    refill stops right after the FIRST row of the FB_READ_BUFFER_SIZE-th new frame, so that
    frame (40, 79, 118, ...) is delivered with a single point and the rest of its rows are
    parsed into a dict entry nobody asks for.
    """

    _KEYS = ("x", "y", "z", "doppler", "peakVal", "posix")

    def __init__(self, experiment_path):
        self.experiment_path = experiment_path
        self.frame_count = 0
        self.pointer = [0, 1]   # [rows already consumed in the current shard, shard number]
        self.read_next_frames()

    def read_next_frames(self):
        self.pointclouds = {}
        self.last_frame = None
        want = const.FB_READ_BUFFER_SIZE
        while len(self.pointclouds) < want:
            path = os.path.join(self.experiment_path, f"{self.pointer[1]}.csv")
            if not os.path.isfile(path):
                break
            stopped_inside = False
            with open(path, "r") as fh:
                for row_no, row in enumerate(csv.reader(fh)):
                    if row_no < self.pointer[0]:
                        continue
                    frame = int(row[0])
                    vals = [float(row[1]), float(row[2]), float(row[3]), float(row[4]), float(row[5]), int(row[6])]
                    slot = self.pointclouds.get(frame)
                    if slot is None:
                        self.pointclouds[frame] = {k: [v] for k, v in zip(self._KEYS, vals)}
                    else:
                        for k, v in zip(self._KEYS, vals):
                            slot[k].append(v)
                    self.last_frame = frame
                    if len(self.pointclouds) >= want:
                        self.pointer[0] = row_no + 1
                        stopped_inside = True
                        break
            if not stopped_inside:
                self.pointer = [0, self.pointer[1] + 1]
        self._finish_frames()

    def _finish_frames(self):
        """The columns of a parsed frame as float64 arrays (the reference keeps Python lists; its live source, ReadDataIWR1443,
        hands numpy arrays under the same keys): the consumers index and stack them either way, and the per-frame calls do not
        pay a list -> array conversion of five columns each time (tens of microseconds of a 150 us frame)."""
        for slot in self.pointclouds.values():
            for k in self._KEYS[:5]:
                if isinstance(slot[k], list):
                    slot[k] = np.asarray(slot[k], dtype=np.float64)

    def get_data(self):
        self.frame_count += 1
        if self.frame_count > self.last_frame:
            self.read_next_frames()
        data = self.pointclouds.get(self.frame_count)
        return (data is not None), self.frame_count, data

    def is_finished(self):
        return self.last_frame is None


def altered_EuclideanDist(p1, p2):
    """The clustering metric (Utils.py:222-247); the GPU kernels evaluate the same
    expression in the same order (csrc/mmw_math.hpp: alt_dist)."""
    w = 1 - ((p1[1] + p2[1]) / 2) * const.DB_RANGE_WEIGHT
    return w * ((p1[0] - p2[0]) ** 2 + (p1[1] - p2[1]) ** 2 + const.DB_Z_WEIGHT * ((p1[2] - p2[2]) ** 2))


def dbscan_labels(pointcloud, eps=None, min_samples=None) -> np.ndarray:
    """sklearn-compatible labels of the reference's DBSCAN call (Utils.py:272-278)."""
    pc = np.asarray(pointcloud, dtype=np.float64)
    eps = const.DB_EPS if eps is None else eps
    min_samples = const.DB_MIN_SAMPLES_MIN if min_samples is None else min_samples
    # sklearn's own refusals, as ValueError (InvalidParameterError is one): parameters first (_validate_params), then the array
    if not (float(eps) > 0.0):
        raise ValueError(f"The 'eps' parameter of DBSCAN must be a float in the range (0.0, inf). Got {eps!r} instead.")
    if int(min_samples) != min_samples or min_samples < 1:
        raise ValueError(f"The 'min_samples' parameter of DBSCAN must be an int in the range [1, inf). Got {min_samples!r} instead.")
    if pc.ndim != 2:
        raise ValueError(f"Expected 2D array, got {pc.ndim}D array instead")
    if pc.shape[0] == 0:
        raise ValueError(f"Found array with 0 sample(s) (shape={pc.shape}) while a minimum of 1 is required by DBSCAN.")
    pc = pc.reshape(-1, 8)
    n = pc.shape[0]
    sb = _util_ctx()
    if n > sb.UM:
        sb = _util_ctx(fb_frames_batch=3)   # capacity only: the largest cloud a context can hold (4 frames of 1024 points)
    if n > sb.UM:
        raise ValueError(f"apply_DBscan on {n} points: the GPU BallTree emulation holds at most {sb.UM}")
    pts = np.zeros((1, n, 8))
    pts[0] = pc
    labels, _ = sb.dbscan_host(pts, np.array([n], np.int32), eps=float(eps), min_samples=int(min_samples))
    return labels[0, :n].copy()


def apply_DBscan(pointcloud, eps=None, min_samples=None):
    """List of clusters (ascending label, rows in input order), noise dropped (Utils.py:281-291)."""
    pc = np.asarray(pointcloud)
    labels = dbscan_labels(pc, eps, min_samples)
    return [[pc[i] for i in np.nonzero(labels == k)[0]] for k in range(int(labels.max()) + 1 if len(labels) else 0)]


def normalize_data(detObj) -> np.ndarray:
    """Sensor frame -> room frame + scene filter (Utils.py:342-434): dict of equal-length
    lists x,y,z,doppler,peakVal -> (N', 8) float64 [x,y,z,vx,vy,vz,doppler,peakVal]."""
    raw = np.vstack((detObj["x"], detObj["y"], detObj["z"], detObj["doppler"], detObj["peakVal"])).T.astype(np.float64)
    n = raw.shape[0]
    if n == 0:
        return np.empty((0, 8), dtype="float")
    sb = _util_ctx()
    if n > sb.max_pts:
        raise ValueError(f"normalize_data on {n} points (limit {sb.max_pts})")
    buf = np.zeros((1, sb.max_pts, 5))
    buf[0, :n] = raw
    pts, n_out = sb.normalize_host(buf, np.array([n], np.int32))
    return pts[0, : n_out[0]].copy()


def relative_coordinates(absolute_coords, reference):
    """x,y relative to `reference` for every frame (Utils.py:437-465).  Host glue."""
    shift = np.zeros(8)
    shift[0], shift[1] = reference[0], reference[1]
    return [np.asarray(frame, dtype=np.float64) - shift for frame in absolute_coords]


def format_single_frame(track_cloud, mean=None, std_dev=None) -> np.ndarray:
    """MARS feature map of one track (Utils.py:468-520): per frame take columns
    (x,y,z,doppler,intensity), normalise intensity, pad/cut to 64 rows, sort by x,
    reshape to (FB_FRAMES_BATCH+1, 8, 8, 5) -- (8, 8, 5) when FB_FRAMES_BATCH == 0."""
    ring = const.FB_FRAMES_BATCH + 1
    frames = list(track_cloud)
    if len(frames) > ring:
        raise IndexError(f"{len(frames)} frames for a feature map of {ring}")
    over = dict(fb_frames_batch=const.FB_FRAMES_BATCH)
    if mean is not None:
        over["intensity_mu"] = float(mean)
    if std_dev is not None:
        over["intensity_std"] = float(std_dev)
    sb = _util_ctx(**over)
    rows = np.zeros((1, ring, 64, 8))
    counts = np.zeros((1, ring), dtype=np.int32)
    for k, fr in enumerate(frames):
        fr = np.asarray(fr, dtype=np.float64).reshape(-1, 8)
        m = min(64, fr.shape[0])
        rows[0, k, :m] = fr[:m]
        counts[0, k] = m
    b_rows = sb.buf("ff_rows", rows.nbytes).upload(rows)
    b_cnt = sb.buf("ff_cnt", counts.nbytes).upload(counts)
    b_ref = sb.buf("ff_ref", 16).upload(np.zeros(2))
    b_out = sb.buf("ff_out", ring * 64 * 5 * 4)
    sb._chk(sb.L.mmw_format_frames(sb.h, b_rows.ptr, b_cnt.ptr, b_ref.ptr, b_out.ptr, 1))
    out = b_out.download((ring, 8, 8, 5), np.float32)
    return out[0] if ring == 1 else out


# ---- dataset-side formatters (training-data preparation; host numpy, not on the per-frame path) ------------
def format_batched_frames(frame_clouds) -> np.ndarray:
    """The 3 x 64 row block preprocessing saves per track (Utils.py:523-549): frames newest first, per frame the
    columns (x, y, z, doppler, intensity), cut or zero-padded to 64 rows, NOT sorted; always (192, 5)."""
    full = np.zeros((3 * 64, 5))
    for k, cloud in enumerate(reversed(list(frame_clouds))):
        rows = np.asarray(cloud, dtype=np.float64)[:64][:, [0, 1, 2, -2, -1]]
        full[k * 64: k * 64 + rows.shape[0]] = rows
    return full


def format_single_frame_mode(track_cloud: np.ndarray, mean, std_dev, batch_size, fuse=False) -> np.ndarray:
    """A saved (192, 5) block into CNN input (Utils.py:552-574).  Like the reference it normalises the intensity
    column of `track_cloud` IN PLACE.  fuse=False: the first `batch_size` frames as (batch_size, 8, 8, 5);
    fuse=True: the non-zero rows of those frames merged into ONE frame -- cut / padded to 64, sorted by x --
    as (8, 8, 5)."""
    track_cloud[:, 4] = (track_cloud[:, 4] - mean) / std_dev
    limited = track_cloud[: 64 * batch_size]
    if not fuse:
        return limited.reshape((batch_size, 8, 8, 5))
    rows = limited[np.any(limited != 0, axis=1)][:64]
    # dtype as the reference's concatenate gives it: a full frame keeps the input's, a padded one is joined with
    # float64 zeros (the dataset formatter feeds float32 blocks)
    frame = np.zeros((64, 5), dtype=rows.dtype if rows.shape[0] == 64 else np.result_type(rows.dtype, np.float64))
    frame[: rows.shape[0]] = rows
    return frame[np.argsort(frame[:, 0])].reshape((8, 8, 5))


# ---- output step after the path (what the reference's visualiser computes per track) ------------------------
def calc_projection_points(x_origin, y_origin, z_origin):
    """Where the line from the monitoring point (const.M_X, M_Y, M_Z) to a point cuts the screen plane y = 0
    (Utils.py:180-219): (x, z) on the screen.  Scalars or arrays (one value per track)."""
    x_o, y_o, z_o = (np.asarray(v, dtype=np.float64) for v in (x_origin, y_origin, z_origin))
    x_d, y_d, z_d = x_o - const.M_X, y_o - const.M_Y, z_o - const.M_Z
    with np.errstate(divide="ignore", invalid="ignore"):
        x_p = np.where(x_d == 0, x_o, -const.M_Y / (y_d / np.where(x_d == 0, 1.0, x_d)) + const.M_X)
        z_p = np.where(z_d == 0, z_o, -const.M_Y / (y_d / np.where(z_d == 0, 1.0, z_d)) + const.M_Z)
    if x_p.ndim == 0:
        return float(x_p), float(z_p)
    return x_p, z_p


def calc_fade_square(track):
    """Centre and size of the square the smart window fades behind a person (Visualizer.py:14-29): the head
    keypoint (x index 3, y index 41, z index 22 of the 57-vector) relative to the track position, projected onto
    the screen; the size shrinks with range.  `track` is anything with `.state.x` and `.keypoints`."""
    x = np.asarray(track.state.x, dtype=np.float64).reshape(-1)      # (dim_x, 1) column in the reference
    kp = np.asarray(track.keypoints, dtype=np.float64).reshape(-1)
    x0, x1 = float(x[0]), float(x[1])
    centre = calc_projection_points(x0 + float(kp[3]), x1 + float(kp[41]), float(kp[22]))
    size = max(const.V_SCREEN_FADE_SIZE_MIN,
               min(const.V_SCREEN_FADE_SIZE_MAX, const.V_SCREEN_FADE_SIZE_MAX - (x1 + float(kp[12])) * const.V_SCREEN_FADE_WEIGHT))
    return centre, size


def fade_squares(state_x: np.ndarray, keypoints: np.ndarray):
    """calc_fade_square for a whole track table at once: state_x[..., >=2], keypoints[..., 57] ->
    (x_proj[...], z_proj[...], size[...]); what a consumer of the all-gathered table (dist.py) calls."""
    sx, kp = np.asarray(state_x, dtype=np.float64), np.asarray(keypoints, dtype=np.float64)
    x_p, z_p = calc_projection_points(sx[..., 0] + kp[..., 3], sx[..., 1] + kp[..., 41], kp[..., 22])
    size = np.clip(const.V_SCREEN_FADE_SIZE_MAX - (sx[..., 1] + kp[..., 12]) * const.V_SCREEN_FADE_WEIGHT,
                   const.V_SCREEN_FADE_SIZE_MIN, const.V_SCREEN_FADE_SIZE_MAX)
    return x_p, z_p, size

