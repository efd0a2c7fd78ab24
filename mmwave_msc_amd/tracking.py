"""Single-scene, reference-shaped face of the GPU tracker.

Mirrors the public surface of the reference's `src/Tracking.py` that its callers
use (offline_main.py:32-60, Visualizer.py:16-27, 232-283):

    trackbuffer = TrackBuffer(); batch = BatchedData()
    trackbuffer.dt = ...; trackbuffer.t = ...
    trackbuffer.track(effective_data, batch)
    trackbuffer.estimate_posture(model)
    for track in trackbuffer.effective_tracks:
        track.state.x, track.cluster.centroid, track.keypoints, track.batch.effective_data,
        track.lifetime, track.color ...

All state lives on the GPU inside a 1-scene `SceneBatch`; the objects below are
views that read it back on demand.  The batched API (`batch.SceneBatch`) is the
throughput path; this one exists so the offline loop drops in unchanged.
"""
from __future__ import annotations

import time
from typing import List

import numpy as np

from . import _lib
from . import constants as const
from .batch import SceneBatch

ACTIVE, INACTIVE = 1, 0
STATIC, DYNAMIC = True, False


class BatchedData:
    """Handle on the scene's global frame ring (reference Tracking.py:21-71).  The
    frames themselves live on the device; `buffer` / `effective_data` fetch them."""

    def __init__(self, init_data=None):
        self._owner = None   # bound TrackBuffer
        # BatchedData(init_data) (Tracking.py:38-41): the ring starts with this frame instead of an empty one; it is
        # uploaded when a TrackBuffer binds the object (its first track() call)
        self._init = None if init_data is None else np.asarray(init_data, dtype=np.float64).reshape(-1, 8)
        self.size = const.FB_FRAMES_BATCH + 1
        self._size_changed = False

    def _bind(self, owner):
        self._owner = owner
        if self._init is not None and len(self._init):
            owner._sb.set_batch_frame(0, self._init)
        if self._size_changed:
            owner._sb.set_batch_size(self.size, [0])

    def change_buffer_size(self, new_size):
        """Tracking.py:60-64: from the next add_frame on, frames are popped while len(buffer) >= new_size."""
        self.size = new_size
        self._size_changed = True
        if self._owner is not None and self._owner._sb is not None:
            self._owner._sb.set_batch_size(int(new_size), [0])

    @property
    def buffer(self):
        if self._owner is None or self._owner._sb is None:
            return [np.empty((0, 8)) if self._init is None else self._init.copy()]
        sb = self._owner._sb
        ln, _ = sb.batch_ring()
        return [sb.batch_ring_frame(0, k) for k in range(int(ln[0]))]

    @property
    def effective_data(self):
        fr = self.buffer
        if not fr:
            return np.array([])  # state after clear() (Tracking.py:57-58)
        return np.concatenate(fr, axis=0)

    def pop_frame(self):
        """Remove the oldest frame (Tracking.py:66-71; called by the dataset pre-processing between shards)."""
        if self._owner is not None and self._owner._sb is not None:
            self._owner._sb.pop_frame([0])
            self._owner._tracks_cache = None


class _TrackRing:
    def __init__(self, tb, index, rec):
        self._tb, self._index, self._rec = tb, index, rec

    @property
    def buffer(self):
        return [self._tb._sb.track_ring_frame(0, self._index, k) for k in range(int(self._rec["ring_len"]))]

    @property
    def effective_data(self):
        return np.concatenate(self.buffer, axis=0)


class KalmanState:
    def __init__(self, rec, dx):
        self.x = np.array(rec["x"][:dx], dtype=np.float64).reshape(dx, 1)
        self.P = np.array(rec["P"][:dx, :dx], dtype=np.float64)
        self.H = const.MOTION_MODEL.KF_H
        self.dim_x, self.dim_z = dx, 6


class PointCluster:
    def __init__(self, rec):
        self.point_num = int(rec["point_num"])
        self.centroid = np.array(rec["centroid"])
        self.min_vals = np.array(rec["min_vals"])
        self.max_vals = np.array(rec["max_vals"])
        self.status = bool(rec["is_static"])


class ClusterTrack:
    """Read-only view of one entry of `effective_tracks` (reference Tracking.py:139-230)."""

    def __init__(self, tb, index, rec, color):
        dx = tb._dx
        self.N_est = float(rec["n_est"])
        self.spread_est = np.array(rec["spread_est"])
        self.group_disp_est = np.array(rec["group_disp_est"])
        self.cluster = PointCluster(rec)
        self.batch = _TrackRing(tb, index, rec)
        self.state = KalmanState(rec, dx)
        self.status = ACTIVE
        self.lifetime = float(rec["lifetime"])
        self.keypoints = np.array(rec["keypoints"])
        self.predict_x = self.state.x
        self.color = color
        self.uid = int(rec["uid"])

    def get_Rm(self):
        return np.diag((self.spread_est / 2) ** 2)


class TrackBuffer:
    """GPU-backed counterpart of the reference TrackBuffer (Tracking.py:451-734)."""

    def __init__(self, max_pts: int = 512, device: int = 0):
        self.next_track_id = 0
        self.dt = 0
        self.t = time.time()
        self._max_pts, self._device = int(max_pts), int(device)
        self._sb = None
        self._batch = None
        self._tracks_cache = None
        self._colors = {}
        self._dx = const.MOTION_MODEL.KF_DIM[0]
        self.last_assoc = None       # _calc_dist_fun output of the last frame (-1 = None)
        self.last_db_labels = None   # apply_DBscan labels of the last frame, or None

    def _ensure(self):
        if self._sb is None:
            ring = const.FB_FRAMES_BATCH + 1
            cfg = const.to_config(ring_rows=max(64, ring * self._max_pts))
            self._dx = const.MOTION_MODEL.KF_DIM[0]
            self._sb = SceneBatch(cfg, 1, self._max_pts, self._device)
        return self._sb

    # -- reference API -----------------------------------------------------------
    def track(self, pointcloud, batch: BatchedData):
        sb = self._ensure()
        if self._batch is None:
            self._batch = batch
            batch._bind(self)
        elif batch is not self._batch:
            raise ValueError("this TrackBuffer is bound to another BatchedData (one global ring per scene)")
        pc = np.asarray(pointcloud, dtype=np.float64).reshape(-1, 8)
        n = pc.shape[0]
        if n > self._max_pts:
            raise ValueError(f"frame has {n} points; TrackBuffer(max_pts={self._max_pts})")
        pts = np.zeros((1, self._max_pts, 8))
        pts[0, :n] = pc
        # an empty cloud still IS a track() call (predict, ageing / expiry, _update_all, an empty ring frame): the C-ABI's
        # count 0 means "frame skipped" (offline_main.py:56), MMW_EMPTY_FRAME means this
        assoc, labels, dbn = sb.step_host(pts, np.array([n if n > 0 else _lib.EMPTY_FRAME], np.int32), np.array([float(self.dt)]))
        self.last_assoc = assoc[0, :n].copy()
        self.last_db_labels = labels[0, : dbn[0]].copy() if dbn[0] >= 0 else None
        self._tracks_cache = None

    def estimate_posture(self, model):
        """Tracking.py:705-734.  `model` is either a `mars.MarsCNN` (runs on the GPU) or any
        object with a Keras-style `.predict(ndarray[B,3,8,8,5]) -> ndarray[B,57]`."""
        sb = self._ensure()
        feat, owner = sb.features_host()
        if len(owner) == 0:
            return
        if hasattr(model, "predict_numpy"):
            kp = model.predict_numpy(feat)
        else:
            kp = np.asarray(model.predict(feat), dtype=np.float32)
        sb.set_keypoints_host(kp, owner)
        self._tracks_cache = None

    @property
    def effective_tracks(self) -> List[ClusterTrack]:
        if self._sb is None:
            return []
        if self._tracks_cache is None:
            nt = int(self._sb.num_tracks()[0])
            recs = self._sb.tracks(cap=max(nt, 1))[0, :nt]
            out = []
            for j in range(nt):
                uid = int(recs[j]["uid"])
                if uid not in self._colors:
                    self._colors[uid] = np.random.rand(3)  # cosmetic, as Tracking.py:228
                out.append(ClusterTrack(self, j, recs[j], self._colors[uid]))
            self.next_track_id = max([self.next_track_id] + [t.uid + 1 for t in out])
            self._tracks_cache = out
        return self._tracks_cache

    def has_active_tracks(self) -> bool:
        return len(self.effective_tracks) > 0

    def close(self):
        if self._sb is not None:
            self._sb.close()
            self._sb = None
