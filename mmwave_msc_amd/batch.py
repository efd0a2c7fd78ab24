"""SceneBatch: S independent scenes (S reference `TrackBuffer`s + their global
`BatchedData`) resident on one MI355X, stepped together through the C-ABI.

This is the batched face of the hot path; `tracking.TrackBuffer` is the
single-scene, reference-shaped face built on top of it.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import MmwError, RING_MAX, NKP, SUMMARY_DTYPE, TRACK_DTYPE, error_for


class DevBuf:
    """A device allocation owned by a context (hipMalloc through the C-ABI)."""

    def __init__(self, batch: "SceneBatch", nbytes: int):
        self.batch, self.nbytes = batch, int(nbytes)
        p = C.c_void_p()
        batch._chk(batch.L.mmw_dev_alloc(batch.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes, (a.nbytes, self.nbytes)
        self.batch._chk(self.batch.L.mmw_memcpy_h2d(self.batch.h, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes, (out.nbytes, self.nbytes)
        self.batch._chk(self.batch.L.mmw_memcpy_d2h(self.batch.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr and self.batch.h:
            self.batch.L.mmw_dev_free(self.batch.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class SceneBatch:
    def __init__(self, cfg: "_lib.MmwConfig | None" = None, n_scenes: int = 1, max_pts: int = 512, device: int = 0):
        self.L = _lib.load()
        self.cfg = cfg if cfg is not None else _lib.default_config()
        self.h = None
        h = C.c_void_p()
        rc = self.L.mmw_create(C.byref(self.cfg), int(n_scenes), int(max_pts), int(device), C.byref(h))
        if rc != 0:
            raise MmwError(rc, (self.L.mmw_last_error(None) or b"").decode())
        self.h = h
        self.S, self.max_pts, self.device = int(n_scenes), int(max_pts), int(device)
        dims = [C.c_int32() for _ in range(5)]
        self._chk(self.L.mmw_get_dims(self.h, *[C.byref(d) for d in dims]))
        self.track_cap, self.ring, self.ring_rows = dims[2].value, dims[3].value, dims[4].value
        self.UM = self.ring * self.max_pts
        self._bufs = {}
        self._frame_out = None      # frame_host(reuse_out=True): the result arrays of the last call
        self._posture_model = None

    # -- plumbing -------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise error_for(rc, (self.L.mmw_last_error(self.h) or b"").decode())

    def close(self):
        if self.h:
            for b in list(self._bufs.values()):
                b.free()
            self._bufs.clear()
            self.L.mmw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def buf(self, name: str, nbytes: int) -> DevBuf:
        b = self._bufs.get(name)
        if b is None or b.nbytes < nbytes:
            if b is not None:
                b.free()
            b = DevBuf(self, nbytes)
            self._bufs[name] = b
        return b

    def alloc(self, nbytes: int) -> DevBuf:
        return DevBuf(self, nbytes)

    def set_stream(self, stream_ptr):
        """Raw hipStream_t (0 / None = the context's own non-blocking stream; 1 = HIP's legacy default stream)."""
        self._chk(self.L.mmw_set_stream(self.h, stream_ptr))

    def follow_torch_stream(self, stream=None):
        """Run this context's kernels on a torch stream (default: torch's current one), so that torch ops and the
        `*_dev` calls that share device tensors with them are ordered without host synchronisation.  torch reports
        the legacy default stream as pointer 0, which `mmw_set_stream` reads as "the context's own stream" -- the
        legacy handle (include/mmw.h: MMW_STREAM_LEGACY) is passed for it."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        ptr = int(s.cuda_stream)
        self.set_stream(ptr if ptr else 1)

    def stream_wait(self, stream=None):
        """mmw_stream_wait: work queued on `stream` (a torch stream; default torch's current one on this device) after this call
        starts only when everything queued on the context's stream so far has finished -- no host wait.  For consumers of device
        buffers the context wrote (the all-gather of the track table, a torch op on feature rows) that run on another stream."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._chk(self.L.mmw_stream_wait(self.h, int(s.cuda_stream)))

    def wait_stream(self, stream=None):
        """mmw_wait_stream, the other direction: what this context queues from now on starts only when everything queued on
        `stream` (default torch's current one) so far has finished -- before the context REWRITES a device buffer a consumer on
        that stream may still be reading (the track table of the previous all-gather)."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._chk(self.L.mmw_wait_stream(self.h, int(s.cuda_stream)))

    def set_chain_side_stream(self, on: bool):
        """Small-cloud DBSCAN workers beside the association kernel (second stream) on / off, from the next step on."""
        self._chk(self.L.mmw_set_chain_side_stream(self.h, 1 if on else 0))

    def side_workers(self) -> int:
        """mmw_side_workers: 0 = the DBSCAN chain workers are not in use (not configured, or their streams share a hardware
        queue with the context's stream), 1 = in use, 2 = configured but no step has checked the streams yet."""
        return int(self.L.mmw_side_workers(self.h))

    def diag_queue(self) -> np.ndarray:
        """mmw_diag_queue: the 32 words of the DBSCAN work queues ([4] = bounded waits given up, also reported by check())."""
        out = np.zeros(32, dtype=np.int32)
        self._chk(self.L.mmw_diag_queue(self.h, out.ctypes.data))
        return out

    def streams_concurrent(self, stream_a: int, stream_b: int) -> bool:
        """mmw_streams_concurrent: do kernels on HIP stream b (raw handles) run beside a running kernel of stream a?"""
        rc = int(self.L.mmw_streams_concurrent(self.h, stream_a, stream_b))
        if rc < 0:
            self._chk(rc)
        return rc == 1

    def step_kind(self) -> int:
        """mmw_step_kind: 1 = the one-workgroup step (k_scene), 2 = two launches, 4 = the bulk kernels."""
        return int(self.L.mmw_step_kind(self.h))

    def kalman_layout(self) -> int:
        """mmw_kalman_layout: 1 = the batched Kalman kernels are laid out over the tracks of the context, 0 = per scene."""
        return int(self.L.mmw_kalman_layout(self.h))

    def synchronize(self):
        self._chk(self.L.mmw_synchronize(self.h))

    def pop_frame(self, scenes=None):
        """BatchedData.pop_frame() on the global ring of the given scenes (default: all)."""
        if scenes is None:
            self._chk(self.L.mmw_pop_frame(self.h, None))
            return
        flags = np.zeros(self.S, dtype=np.int32)
        flags[np.asarray(scenes, dtype=np.int64)] = 1
        self._chk(self.L.mmw_pop_frame(self.h, flags.ctypes.data))

    def set_batch_size(self, new_size: int, scenes=None):
        """BatchedData.change_buffer_size(new_size) on the global ring of the given scenes (default: all)."""
        if scenes is None:
            self._chk(self.L.mmw_set_batch_size(self.h, None, int(new_size)))
            return
        flags = np.zeros(self.S, dtype=np.int32)
        flags[np.asarray(scenes, dtype=np.int64)] = 1
        self._chk(self.L.mmw_set_batch_size(self.h, flags.ctypes.data, int(new_size)))

    def set_batch_frame(self, scene: int, rows: np.ndarray):
        """BatchedData(init_data): the scene's global ring becomes one frame holding `rows` (n, 8)."""
        rows = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 8)
        self._chk(self.L.mmw_set_batch_frame(self.h, int(scene), rows.ctypes.data, len(rows)))

    def reset(self):
        self._chk(self.L.mmw_reset(self.h))

    def reset_scenes(self, mask):
        """Fresh TrackBuffer / BatchedData for the scenes where `mask` is true (mmw_reset_scenes); the others keep their state."""
        m = np.ascontiguousarray(np.asarray(mask).astype(np.int32).reshape(self.S))
        self._chk(self.L.mmw_reset_scenes(self.h, m.ctypes.data))

    def errors(self) -> np.ndarray:
        """Sticky error bits per scene (mmw_get_errors): 1 singular, 2 division by zero, 4 capacity, 8 bad point count,
        16 / 32 apply_DBscan reached with a NaN / an infinite value in the ring (sklearn's ValueError, Utils.py:272-278)."""
        out = np.zeros(self.S, dtype=np.int32)
        self._chk(self.L.mmw_get_errors(self.h, out.ctypes.data))
        return out

    def clear_errors(self, bits: int, scenes=None):
        """mmw_clear_errors: clears the given sticky bits (of the given scenes; default all) and nothing else -- for a caller that
        catches the reference's ValueError (non-finite rows: the scene's state is what the reference's is) and carries on."""
        if scenes is None:
            self._chk(self.L.mmw_clear_errors(self.h, None, int(bits)))
            return
        flags = np.zeros(self.S, dtype=np.int32)
        flags[np.asarray(scenes, dtype=np.int64)] = 1
        self._chk(self.L.mmw_clear_errors(self.h, flags.ctypes.data, int(bits)))

    def check(self):
        self._chk(self.L.mmw_check(self.h))

    # -- hot path -------------------------------------------------------------
    def step_dev(self, pts_ptr, n_ptr, dt_ptr, assoc_ptr=None, labels_ptr=None, dbn_ptr=None):
        """TrackBuffer.track for all scenes; every argument is a device pointer (int)."""
        self._chk(self.L.mmw_step(self.h, pts_ptr, n_ptr, dt_ptr, assoc_ptr, labels_ptr, dbn_ptr))

    def step_dev_f32(self, pts_ptr, n_ptr, dt_ptr, assoc_ptr=None, labels_ptr=None, dbn_ptr=None):
        """mmw_step_f32: the same with the frame's rows as fp32 ([S][max_pts][8] float, device pointer), promoted exactly."""
        self._chk(self.L.mmw_step_f32(self.h, pts_ptr, n_ptr, dt_ptr, assoc_ptr, labels_ptr, dbn_ptr))

    def normalize_dev(self, raw_ptr, n_raw_ptr, pts_ptr, n_out_ptr, f32: bool = False):
        """Utils.normalize_data on device buffers: raw[S][max_pts][5] (fp64, or fp32 with f32=True) -> pts[S][max_pts][8] fp64."""
        fn = self.L.mmw_normalize_f32 if f32 else self.L.mmw_normalize
        self._chk(fn(self.h, raw_ptr, n_raw_ptr, pts_ptr, n_out_ptr))

    def normalize_tlv_dev(self, packets_ptr, packets_bytes, tlv_offset_ptr, uart_cfg, pts_ptr, n_out_ptr):
        """mmw_normalize_tlv: the radar's own wire format (detected-points TLV bodies, 12 B per object) decoded as ReadIWR14xx.read
        does (ReadDataIWR1443.py:153-171) and normalised (Utils.normalize_data) in one kernel: packets (device bytes),
        tlv_offset[S] (device int64: byte offset of each scene's TLV body, < 0 = none), uart_cfg (`_lib.MmwUartCfg`, host)
        -> pts[S][max_pts][8] fp64, n_out[S] (device).  Nothing outside packets[0 .. packets_bytes) is read; a body that does not
        fit, or announces more than max_pts objects, gives n_out = _lib.BAD_FRAME (the next step raises the scene's bad count)."""
        self._chk(self.L.mmw_normalize_tlv(self.h, packets_ptr, int(packets_bytes), tlv_offset_ptr, C.byref(uart_cfg), pts_ptr, n_out_ptr))

    def step_host(self, pts: np.ndarray, n: np.ndarray, dt: np.ndarray, raise_nonfinite: bool = True, check: bool = True):
        """Host convenience (H2D + step + D2H).  Returns (assoc[S,NP], labels[S,UM], db_n[S]).  raise_nonfinite=False: a scene
        whose apply_DBscan call raised sklearn's ValueError (a NaN / infinite row in its ring) does not raise here -- its db_n is
        DB_RAISED (-2), its sticky bit stays set (`errors()`, `clear_errors()`), every other scene's results are as always.
        check=False: no scene error raises (the caller reads `errors()`); failures of the call itself still do."""
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        n = np.ascontiguousarray(n, dtype=np.int32)
        dt = np.ascontiguousarray(dt, dtype=np.float64)
        assert pts.shape == (self.S, self.max_pts, 8) and n.shape == (self.S,) and dt.shape == (self.S,)
        assoc = np.full((self.S, self.max_pts), -1, dtype=np.int32)
        labels = np.full((self.S, self.UM), -1, dtype=np.int32)
        dbn = np.full(self.S, -1, dtype=np.int32)
        rc = self.L.mmw_step_host(self.h, pts.ctypes.data, n.ctypes.data, dt.ctypes.data,
                                  assoc.ctypes.data, labels.ctypes.data, dbn.ctypes.data)
        scene_error = rc in (_lib.E_NONFINITE, _lib.E_SINGULAR, _lib.E_DIVZERO, _lib.E_CAPACITY)   # (the results are complete: the check is the last thing)
        if not ((rc == _lib.E_NONFINITE and not raise_nonfinite) or (scene_error and not check)):
            self._chk(rc)
        return assoc, labels, dbn

    def attach_posture(self, model=None):
        """mmw_attach_posture: hands the context a `mars.MarsCNN` (3-frame model, on this GPU) so that `frame_host(posture=True)`
        runs TrackBuffer.estimate_posture behind the step in the same round trip; None detaches.  The context keeps the
        weights' device pointers: the model must stay where it is (in-place updates are fine) until detached."""
        if model is None:
            self._chk(self.L.mmw_attach_posture(self.h, None))
            self._posture_model = None
            return
        w1 = model.dense1_dhwc.weight
        m = _lib.MmwPostureModel(model.k_w1.data_ptr(), model.k_b1.data_ptr(), model.k_w2.data_ptr(), model.k_b2.data_ptr(),
                                 w1.data_ptr(), w1.stride(0), model.dense1_dhwc.bias.data_ptr(), model.dense2.weight.data_ptr(),
                                 model.dense2.bias.data_ptr())
        self._chk(self.L.mmw_attach_posture(self.h, C.byref(m)))
        self._posture_model = model   # (keeps the tensors alive)

    def frame_host(self, n: np.ndarray, dt: np.ndarray, raw: np.ndarray = None, pts: np.ndarray = None, want_rows: bool = False,
                   want_labels: bool = True, posture: bool = False, reuse_out: bool = False):
        """mmw_frame_host: one frame of every scene from host memory in ONE round trip.  `raw`[S,NP,5] radar rows (normalised
        on the device, Utils.normalize_data) or `pts`[S,NP,8] normalised rows; n[S], dt[S].  Returns a dict: assoc[S,NP],
        db_n[S], n_out[S] (rows that reached track()), n_tracks[S], labels[S,UM] (want_labels), rows[S,NP,8] (want_rows, raw form)."""
        assert (raw is None) != (pts is None)
        src = np.ascontiguousarray(raw if raw is not None else pts, dtype=np.float64)
        assert src.shape == (self.S, self.max_pts, 5 if raw is not None else 8), src.shape
        n = np.ascontiguousarray(n, dtype=np.int32)
        dt = np.ascontiguousarray(dt, dtype=np.float64)
        if reuse_out and self._frame_out is not None:
            out = self._frame_out   # (the caller copies what it keeps before the next call: TrackBuffer)
            out.pop("posture_rows", None)
        else:
            out = {"assoc": np.full((self.S, self.max_pts), -1, dtype=np.int32), "db_n": np.full(self.S, -1, dtype=np.int32),
                   "n_out": np.zeros(self.S, dtype=np.int32), "n_tracks": np.zeros(self.S, dtype=np.int32),
                   "labels": np.full((self.S, self.UM), -1, dtype=np.int32)}
            if reuse_out:
                self._frame_out = out
        if not want_labels and not reuse_out:
            out.pop("labels")
        if want_rows and raw is not None:
            out["rows"] = np.zeros((self.S, self.max_pts, 8))
        elif "rows" in out:
            out.pop("rows")
        args = (self.h, src.ctypes.data if raw is not None else None, src.ctypes.data if raw is None else None,
                n.ctypes.data, dt.ctypes.data, out["rows"].ctypes.data if "rows" in out else None,
                out["n_out"].ctypes.data, out["assoc"].ctypes.data,
                out["labels"].ctypes.data if want_labels else None, out["db_n"].ctypes.data, out["n_tracks"].ctypes.data)
        if posture:   # ... and estimate_posture with the attached model behind the step (mmw_frame_posture_host)
            rows = C.c_int32(0)
            rc = self.L.mmw_frame_posture_host(*args, C.byref(rows))
            out["posture_rows"] = int(rows.value)   # (written before the first scene error is reported: a caller that catches the
            self._chk(rc)                           #  reference's ValueError must still see that estimate_posture ran on the device)
        else:
            self._chk(self.L.mmw_frame_host(*args))
        return out

    def normalize_host(self, raw: np.ndarray, n_raw: np.ndarray):
        """Utils.normalize_data for all scenes: raw[S,NP,5] -> (pts[S,NP,8], n_out[S])."""
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        n_raw = np.ascontiguousarray(n_raw, dtype=np.int32)
        assert raw.shape == (self.S, self.max_pts, 5)
        b_raw = self.buf("norm_raw", raw.nbytes).upload(raw)
        b_n = self.buf("norm_n", n_raw.nbytes).upload(n_raw)
        b_out = self.buf("norm_out", self.S * self.max_pts * 64)
        b_no = self.buf("norm_no", self.S * 4)
        self._chk(self.L.mmw_normalize(self.h, b_raw.ptr, b_n.ptr, b_out.ptr, b_no.ptr))
        n_out = b_no.download((self.S,), np.int32)
        pts = b_out.download((self.S, self.max_pts, 8), np.float64)
        for s in range(self.S):
            pts[s, n_out[s]:] = 0.0
        return pts, n_out

    def dbscan_host(self, pts: np.ndarray, n: np.ndarray, eps=None, min_samples=None, raise_nonfinite: bool = True):
        """Utils.apply_DBscan labels for S clouds: pts[S,max_n,8] -> (labels[S,max_n], n_clusters[S]).  A cloud with a NaN / an
        infinite value raises ValueError as sklearn's input validation does (Utils.py:272-278); with raise_nonfinite=False such
        clouds come back with n_clusters = -16 (NaN) / -32 (infinity) and labels -1."""
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        n = np.ascontiguousarray(n, dtype=np.int32)
        max_n = pts.shape[1]
        assert pts.shape == (self.S, max_n, 8)
        b_p = self.buf("db_pts", pts.nbytes).upload(pts)
        b_n = self.buf("db_n", n.nbytes).upload(n)
        b_l = self.buf("db_lab", self.S * max_n * 4)
        b_c = self.buf("db_ncl", self.S * 4)
        self._chk(self.L.mmw_dbscan(self.h, b_p.ptr, b_n.ptr, max_n,
                                    self.cfg.db_eps if eps is None else float(eps),
                                    self.cfg.db_min_samples if min_samples is None else int(min_samples),
                                    b_l.ptr, b_c.ptr))
        labels = b_l.download((self.S, max_n), np.int32)
        ncl = b_c.download((self.S,), np.int32)
        for s in range(self.S):
            labels[s, n[s]:] = -1
            if ncl[s] < 0:   # sklearn's input validation refused the cloud (a NaN / an infinite value): no labels
                labels[s] = -1
        if raise_nonfinite and (ncl < 0).any():
            s = int(np.flatnonzero(ncl < 0)[0])
            raise _lib.MmwNonFinite(_lib.E_NONFINITE, f"cloud {s}: " + ("Input X contains NaN." if ncl[s] == -_lib.ERRBIT_NONFINITE_NAN
                                                                       else "Input X contains infinity or a value too large for dtype('float64')."))
        return labels, ncl

    def features_dev(self, feat_ptr, owner_ptr, cap_rows: int) -> int:
        nrows = C.c_int32(0)
        self._chk(self.L.mmw_features(self.h, feat_ptr, owner_ptr, int(cap_rows), C.byref(nrows)))
        return nrows.value

    def features_async(self, feat_ptr, owner_ptr, uid_ptr, cap_rows: int, ticket: int = 0):
        """mmw_features without the host wait (pipelined callers): rows behind the last step on the context's stream."""
        self._chk(self.L.mmw_features_async(self.h, feat_ptr, owner_ptr, uid_ptr, int(cap_rows), int(ticket)))

    def features_wait(self, ticket: int = 0) -> int:
        """Rows of the `features_async` call with this ticket (waits for its total only, not for the stream)."""
        nrows = C.c_int32(0)
        self._chk(self.L.mmw_features_wait(self.h, int(ticket), C.byref(nrows)))
        return nrows.value

    def set_keypoints_uid_dev(self, kp_ptr, owner_ptr, uid_ptr, n_rows: int):
        """Keypoint scatter matched by track creation ordinal (rows taken one or more frames ago)."""
        self._chk(self.L.mmw_set_keypoints_uid(self.h, kp_ptr, owner_ptr, uid_ptr, int(n_rows)))

    def features_host(self, cap_rows=None):
        """Returns (feat[B,ring,8,8,5] float32 (or [B,8,8,5] when ring == 1), owner[B,2])."""
        cap = int(cap_rows if cap_rows is not None else self.S * self.track_cap)
        per = self.ring * 64 * 5
        b_f = self.buf("feat", max(cap, 1) * per * 4)
        b_o = self.buf("owner", max(cap, 1) * 8)
        nrows = self.features_dev(b_f.ptr, b_o.ptr, cap)
        shape = (nrows, self.ring, 8, 8, 5) if self.ring > 1 else (nrows, 8, 8, 5)
        if nrows == 0:
            return np.zeros(shape, np.float32), np.zeros((0, 2), np.int32)
        feat = b_f.download((nrows, per), np.float32).reshape(shape)
        owner = b_o.download((nrows, 2), np.int32)
        return feat, owner

    def set_keypoints_dev(self, kp_ptr, owner_ptr, n_rows: int):
        self._chk(self.L.mmw_set_keypoints(self.h, kp_ptr, owner_ptr, int(n_rows)))

    def set_keypoints_host(self, kp: np.ndarray, owner: np.ndarray):
        kp = np.ascontiguousarray(kp, dtype=np.float32).reshape(-1, NKP)
        owner = np.ascontiguousarray(owner, dtype=np.int32).reshape(-1, 2)
        if len(owner) == 0:
            return
        b_k = self.buf("kp", kp.nbytes).upload(kp)
        b_o = self.buf("kp_owner", owner.nbytes).upload(owner)
        self.set_keypoints_dev(b_k.ptr, b_o.ptr, len(owner))

    # -- read-back ------------------------------------------------------------
    def num_tracks(self) -> np.ndarray:
        out = np.zeros(self.S, dtype=np.int32)
        self._chk(self.L.mmw_get_num_tracks(self.h, out.ctypes.data))
        return out

    def tracks(self, cap=None) -> np.ndarray:
        """effective_tracks of every scene as a structured array [S, cap] (zero past n_tracks)."""
        cap = int(cap if cap is not None else self.track_cap)
        out = np.zeros((self.S, cap), dtype=TRACK_DTYPE)
        self._chk(self.L.mmw_get_tracks(self.h, out.ctypes.data, cap))
        return out

    def batch_ring(self):
        ln = np.zeros(self.S, dtype=np.int32)
        rn = np.zeros((self.S, RING_MAX), dtype=np.int32)
        self._chk(self.L.mmw_get_batch_ring(self.h, ln.ctypes.data, rn.ctypes.data))
        return ln, rn

    def track_ring_frame(self, scene: int, track: int, k: int) -> np.ndarray:
        out = np.zeros((self.ring_rows, 8))
        n = C.c_int32(0)
        self._chk(self.L.mmw_get_track_ring_frame(self.h, scene, track, k, out.ctypes.data, C.byref(n)))
        return out[: n.value].copy()

    def batch_ring_frame(self, scene: int, k: int) -> np.ndarray:
        out = np.zeros((self.max_pts, 8))
        n = C.c_int32(0)
        self._chk(self.L.mmw_get_batch_ring_frame(self.h, scene, k, out.ctypes.data, C.byref(n)))
        return out[: n.value].copy()

    def inner_calls(self, cap_labels=None):
        """seek_inner_clusters calls of the last step (contexts with seek_inner = 1): per scene a list of label arrays,
        one per call, in track-list order (mmw_get_inner)."""
        cap = int(cap_labels if cap_labels is not None else 2 * self.ring * self.ring_rows)
        n = np.zeros(self.S, dtype=np.int32)
        rows = np.zeros((self.S, 16), dtype=np.int32)
        lab = np.full((self.S, cap), -2, dtype=np.int32)
        self._chk(self.L.mmw_get_inner(self.h, n.ctypes.data, rows.ctypes.data, lab.ctypes.data, cap))
        out = []
        for s in range(self.S):
            calls, off = [], 0
            for k in range(min(int(n[s]), 16)):
                calls.append(lab[s, off: off + rows[s, k]].copy())
                off += int(rows[s, k])
            out.append(calls)
        return out

    def track_table_dev(self, table_ptr, slots: int, scene_base: int = 0):
        self._chk(self.L.mmw_track_table(self.h, table_ptr, int(slots), int(scene_base)))

    def track_table_host(self, slots: int, scene_base: int = 0) -> np.ndarray:
        b = self.buf("table", self.S * slots * SUMMARY_DTYPE.itemsize)
        self.track_table_dev(b.ptr, slots, scene_base)
        return b.download((self.S, slots), SUMMARY_DTYPE)

    # -- profiling ------------------------------------------------------------
    def stats(self) -> np.ndarray:
        out = np.zeros(8, dtype=np.uint64)
        self._chk(self.L.mmw_stats_get(self.h, out.ctypes.data))
        return out

    def stats_ext(self) -> np.ndarray:
        """[0..7] as `stats`; [30] k_features algorithmic bytes, [31] feature tensors written (include/mmw.h)."""
        out = np.zeros(32, dtype=np.uint64)
        self._chk(self.L.mmw_stats_get_ext(self.h, out.ctypes.data))
        return out

    def stats_reset(self):
        self._chk(self.L.mmw_stats_reset(self.h))

    def profile(self, on, kernels=None):
        """HIP-event timing of the kernels: all of them (on=True), none (on=False), or the ids in `kernels`."""
        mask = 0
        if kernels is not None:
            for k in kernels:
                mask |= 2 << int(k)
        elif on:
            mask = 1
        self._chk(self.L.mmw_profile_enable(self.h, mask))

    def profile_reset(self):
        self._chk(self.L.mmw_profile_reset(self.h))

    def profile_get(self, kid: int):
        ms, cnt = C.c_double(0), C.c_int64(0)
        self._chk(self.L.mmw_profile_get(self.h, kid, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value
