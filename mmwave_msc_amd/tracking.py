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
        self._n_tracks = 0           # len(effective_tracks) after the last frame (came back with the frame's results)
        self._posture = None         # device buffers of estimate_posture's on-device path (a mars.MarsCNN on this GPU)
        self._raw_buf = None         # track_raw's upload block [1, max_pts, 5]
        self._fused_model = None     # attach_posture_model: estimate_posture runs inside track() / track_raw()'s round trip
        self._posture_done = False   # ... and has run for the last frame
        self._pending = None         # the reference's ValueError of the frame in flight (raised once its results are taken)

    def _ensure(self):
        if self._sb is None:
            ring = const.FB_FRAMES_BATCH + 1
            cfg = const.to_config(ring_rows=max(64, ring * self._max_pts))
            self._dx = const.MOTION_MODEL.KF_DIM[0]
            self._sb = SceneBatch(cfg, 1, self._max_pts, self._device)
        return self._sb

    def attach_posture_model(self, model):
        """The loop body of offline_main.py:53-60 as ONE round trip: with a `mars.MarsCNN` (3-frame model, on this GPU) attached,
        `track()` / `track_raw()` queue `estimate_posture(model)` behind the step -- feature tensors, CNN (Keras' fp32 arithmetic)
        and the keypoint assignment on the device -- before they wait for the frame's results, and the loop's own
        `estimate_posture(model)` call finds the work done.  A caller that does NOT estimate the posture after every tracked
        frame should not attach.  None detaches.  Returns True when the model was attached."""
        sb = self._ensure()
        if model is None:
            sb.attach_posture(None)
            self._fused_model = None
            return False
        ok = getattr(model, "has_small_path", lambda: False)() and sb.ring == 3 and sb.track_cap <= 64
        if ok:
            p0 = next(model.parameters(), None)
            ok = p0 is not None and p0.is_cuda and p0.device.index == self._device
        if not ok:
            return False
        import torch
        torch.cuda.synchronize(self._device)   # (the weights may have been uploaded on another stream a moment ago)
        sb.attach_posture(model)
        self._fused_model = model
        return True

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
        # count 0 means "frame skipped" (offline_main.py:56), MMW_EMPTY_FRAME means this.  ONE round trip (mmw_frame_host).
        r = self._frame(sb, np.array([n if n > 0 else _lib.EMPTY_FRAME], np.int32), pts=pts)
        self._take(r, n)
        self._reraise()

    def _frame(self, sb, n, **kw):
        """One round trip (mmw_frame_host).  The reference's track() raises ValueError when apply_DBscan is reached with a NaN / an
        infinite value in the ring (sklearn's input validation, Utils.py:272-278) -- AFTER everything else the frame does: the
        results are taken as usual, the scene's sticky bit is cleared (the state is exactly what the reference's is when the
        exception leaves track(); the error comes back with the next frame while the row is in the ring) and the exception --
        a ValueError -- is raised once they are in place (`_reraise`)."""
        self._pending = None
        try:
            return sb.frame_host(n, np.array([float(self.dt)]), posture=self._fused_model is not None, reuse_out=True, **kw)
        except _lib.MmwNonFinite as e:
            sb.clear_errors(_lib.ERRBIT_NONFINITE_NAN | _lib.ERRBIT_NONFINITE_INF)
            self._pending = e
            return sb._frame_out

    def _reraise(self):
        e, self._pending = self._pending, None
        if e is not None:
            raise e

    def _take(self, r, n):
        self._posture_done = "posture_rows" in r
        dbn = int(r["db_n"][0])
        self.last_assoc = r["assoc"][0, :n].copy()
        self.last_db_labels = r["labels"][0, :dbn].copy() if dbn >= 0 else None
        self._n_tracks = int(r["n_tracks"][0])
        self._tracks_cache = None

    def track_raw(self, detObj, batch: BatchedData, want_rows: bool = False):
        """`effective_data = normalize_data(detObj)` + `track(effective_data, batch)` (offline_main.py:53-57) as ONE round trip to
        the GPU: the raw rows go up, Utils.normalize_data and TrackBuffer.track run back to back on the device, the results come
        down.  A frame none of whose rows pass the scene filter is skipped, as the reference's loop does.  Returns the number
        of rows that reached track() -- or (rows, effective_data) with want_rows."""
        sb = self._ensure()
        if self._batch is None:
            self._batch = batch
            batch._bind(self)
        elif batch is not self._batch:
            raise ValueError("this TrackBuffer is bound to another BatchedData (one global ring per scene)")
        m = len(detObj["x"])
        if m > self._max_pts:
            raise ValueError(f"frame has {m} points; TrackBuffer(max_pts={self._max_pts})")
        buf = self._raw_buf   # (only the first m rows travel)
        if buf is None:
            buf = self._raw_buf = np.zeros((1, self._max_pts, 5))
        for i, k in enumerate(("x", "y", "z", "doppler", "peakVal")):
            buf[0, :m, i] = detObj[k]
        r = self._frame(sb, np.array([m], np.int32), raw=buf, want_rows=want_rows)
        n = int(r["n_out"][0])
        if n > 0:
            self._take(r, n)
        self._reraise()
        return (n, r["rows"][0, :n].copy()) if want_rows else n

    def estimate_posture(self, model):
        """Tracking.py:705-734.  `model` is either a `mars.MarsCNN` (runs on the GPU) or any
        object with a Keras-style `.predict(ndarray[B,3,8,8,5]) -> ndarray[B,57]`."""
        sb = self._ensure()
        if self._posture_done and model is self._fused_model:
            self._posture_done = False
            return   # attach_posture_model: done behind the step, in the frame's own round trip
        if self._n_tracks == 0 and self.last_assoc is not None:
            return   # no track after the last frame: nothing to estimate (known from the frame's own results, no round trip)
        if getattr(model, "use_hip_conv", False) and hasattr(model, "range_overflow"):
            import torch
            p0 = next(model.parameters(), None)
            if p0 is not None and p0.is_cuda and p0.device.index == self._device:
                return self._estimate_posture_on_device(model, torch)
        feat, owner = sb.features_host()
        if len(owner) == 0:
            return
        if hasattr(model, "predict_numpy"):
            kp = model.predict_numpy(feat)
        else:
            kp = np.asarray(model.predict(feat), dtype=np.float32)
        sb.set_keypoints_host(kp, owner)
        self._tracks_cache = None

    def _estimate_posture_on_device(self, model, torch):
        """A mars.MarsCNN on this GPU: the feature tensors, the CNN and the keypoint scatter stay on the device (the tracker
        runs on the model's torch stream); the host waits once for the row count -- the exact batch size of the matrix
        kernels.  A handful of tracks (always, for one scene) run MarsCNN.forward_small -- Keras' fp32 arithmetic on thin kernels;
        a larger batch the split-fp16 kernels, with one more wait for their range word (Keras' fp32 has no such limit: a batch
        that left fp16's range is computed again in fp32, as predict() does)."""
        sb = self._sb
        dev = torch.device("cuda", self._device)
        if self._posture is None:
            cap = sb.track_cap
            shape = (cap, sb.ring, 8, 8, 5) if sb.ring > 1 else (cap, 8, 8, 5)
            # The context moves to a stream of its own that torch knows (so that the feature tensors, the CNN and the scatter are
            # ordered without host waits) and STAYS there: every later track() / frame_host call runs on it.  Whatever was queued
            # before -- the model's weights uploaded on another stream a moment ago, work on a stream the caller had bound the
            # context to -- is waited for once, here.
            torch.cuda.synchronize(dev)
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                self._posture = dict(stream=st, feat=torch.zeros(shape, dtype=torch.float32, device=dev),
                                     owner=torch.zeros((cap, 2), dtype=torch.int32, device=dev))
            st.synchronize()
            sb.follow_torch_stream(st)
        P = self._posture
        with torch.cuda.stream(P["stream"]), torch.no_grad():
            n = sb.features_dev(P["feat"].data_ptr(), P["owner"].data_ptr(), P["feat"].shape[0])
            if n == 0:
                return
            if n <= model.SMALL_BATCH and model.has_small_path():
                kp = model.forward_small(P["feat"][:n])          # Keras' fp32 arithmetic on thin kernels: no range word, no wait
            else:
                kp = model(P["feat"][:n])
                if model.arith == "f16x3" and model.range_overflow():
                    model.range_fallbacks += 1
                    kp = model(P["feat"][:n], arith=model.fp32_arith())
                kp = kp.float().contiguous()
            sb.set_keypoints_dev(kp.data_ptr(), P["owner"].data_ptr(), n)
            P["kp"] = kp   # (alive until the scatter has run: the next call on this stream replaces it)
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
