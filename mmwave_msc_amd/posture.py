"""`TrackBuffer.estimate_posture` (reference Tracking.py:705-734) for every scene of a
`SceneBatch`, every frame, pipelined one frame behind the tracker.

The reference calls `estimate_posture(model)` right after `track()` (offline_main.py:57-60):
features of the eligible tracks -> `model.predict` -> `track.keypoints`.  Batched over
thousands of scenes the CNN (matrix cores) takes ~15x the tracker (vector ALUs / LDS), and
the two do not compete for the same units, so frame f's CNN runs on a second stream while
frame f+1 is tracked:

    tracker stream A:  step(f) feat(f) | wait cnn(f-2) scatter(f-2) | step(f+1) feat(f+1) | wait cnn(f-1) scatter(f-1) | ...
    CNN stream     B:  ... cnn(f-2) | wait feat(f-1) cnn(f-1) | wait feat(f) cnn(f) | ...

* `mmw_features_async` leaves the row count in pinned host memory; the host waits for THAT copy only, so the matrix
  kernels get their exact batch size.  The CNN of frame f-1 is queued as soon as the count of frame f-1 is there -- i.e.
  behind the CNN of frame f-2, which is still running: stream B never runs dry.  For that the tracker stream must not sit
  behind the CNN that is running NOW, so the scatter lags TWO frames (three buffers): with a lag of one, `feat(f-1)` was
  queued behind `wait cnn(f-2)`, the host got its count only after that CNN had ended, and every step paid tracker +
  features + CNN in sequence (1.72 ms at 4096 scenes where the CNN alone is 1.34).
* The scatter of frame f-2 runs after frame f was tracked, when list positions may have moved
  (`_maintain_tracks`), so it matches tracks by creation ordinal (`mmw_set_keypoints_uid`);
  it runs on the TRACKER stream, so it cannot race a spawn re-using a record.
* The tracker never reads keypoints, so after `drain()` every live track holds exactly what the
  frame-by-frame reference loop would have left in it.
"""
from __future__ import annotations

import torch

from ._lib import NKP


class PostureRangeError(RuntimeError):
    """drain() / close(): the split-fp16 CNN left keypoints meaningless (samples beyond fp16's range that the device-side fp32 repair
    could not take); every other track's keypoints are valid, `range_overflowed` stays set on the pipeline."""


class PosturePipeline:
    def __init__(self, sb, model, cap_rows: int, tracker_stream=None, cnn_stream=None, overlap: bool = True, time_cnn: bool = False):
        self.sb, self.model, self.cap = sb, model, int(cap_rows)
        self.dev = torch.device("cuda", sb.device)
        self.A = tracker_stream if tracker_stream is not None else torch.cuda.Stream(device=self.dev)
        sb.follow_torch_stream(self.A)
        self.B = self.A
        self.overlap_note = "one stream"
        if overlap:
            # A second stream only overlaps if it sits on another hardware queue (the runtime deals streams onto ~4 of them):
            # a few candidates are probed (mmw_streams_concurrent), and without an independent one the schedule is serial
            cands = [cnn_stream] if cnn_stream is not None else []
            self._spare = []
            for _ in range(8):
                st = cands.pop(0) if cands else torch.cuda.Stream(device=self.dev)
                if sb.streams_concurrent(self.A.cuda_stream, st.cuda_stream):
                    self.B = st
                    self.overlap_note = "two streams on independent hardware queues"
                    break
                self._spare.append(st)   # (kept alive: a freed stream would hand its queue slot to the next candidate)
            else:
                self.overlap_note = "no stream on an independent hardware queue found: serial"
            sb.follow_torch_stream(self.A)
        # with the CNN on its own stream beside the tracker, the tracker's own side-stream workers (k_chain) only take
        # compute units from the statically tiled matrix kernels: off in that schedule -- if the library had them on
        # (mmw_side_workers; its choice, not re-derived here) -- for the LIFE of the pipeline (drain() is also called in
        # the middle of a run), and back on in close()
        self._side_was_on = sb.side_workers() != 0
        self._closed = False
        if self._side_was_on and self.B is not self.A:
            sb.set_chain_side_stream(False)
        self.range_overflowed = False   # the split-fp16 CNN met an input / activation outside fp16's range (see drain())
        shape = (self.cap, sb.ring, 8, 8, 5) if sb.ring > 1 else (self.cap, 8, 8, 5)
        # frame f uses buffer f % NBUF; the scatter of a frame follows its CNN at once on one stream (lag 1), two frames
        # later on two (see the module docstring)
        self.NBUF = 3
        self.lag = 2 if self.B is not self.A else 1
        with torch.cuda.stream(self.A):
            self.feat = [torch.zeros(shape, dtype=torch.float32, device=self.dev) for _ in range(self.NBUF)]
            self.owner = [torch.zeros((self.cap, 2), dtype=torch.int32, device=self.dev) for _ in range(self.NBUF)]
            self.uid = [torch.zeros((self.cap,), dtype=torch.int32, device=self.dev) for _ in range(self.NBUF)]
            self.kp = [torch.zeros((self.cap, NKP), dtype=torch.float32, device=self.dev) for _ in range(self.NBUF)]
            # the split-fp16 CNN's fp32 repair of out-of-range samples (mars.MarsCNN.range_fixup: five launches, empty on almost
            # every frame) leaves the CNN stream, which is the critical path of this schedule: it runs on the tracker stream in
            # front of the frame's scatter, off a fix-up list of the frame's own
            self._defer_fixup = (self.B is not self.A and getattr(model, "arith", None) == "f16x3" and hasattr(model, "range_fixup")
                                 and getattr(model, "has_range_fixup", lambda: False)())
            self.sflags = [model.new_fixup_list(self.dev) for _ in range(self.NBUF)] if self._defer_fixup else None
        self.A.synchronize()
        self.ev_feat = [torch.cuda.Event() for _ in range(self.NBUF)]
        self.ev_cnn = [torch.cuda.Event() for _ in range(self.NBUF)]
        self.time_cnn = bool(time_cnn)
        self._cnn_pairs = []       # (start, stop) timing events around model() on the CNN stream
        self.f = 0                 # frames submitted
        self.cnn_done = 0          # frames whose CNN has been queued
        self.scattered = 0         # frames whose keypoints have been scattered
        self.rows = [0] * self.NBUF
        self.rows_total = 0        # feature tensors pushed through the CNN since construction / reset_counters()

    def reset_counters(self):
        self.rows_total = 0
        self._cnn_pairs = []

    def cnn_ms(self):
        """Mean device time of model() per frame since reset_counters() (None unless time_cnn); synchronises."""
        if not self._cnn_pairs:
            return None
        self.B.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._cnn_pairs) / len(self._cnn_pairs)

    def _cnn(self, g: int):
        """queue the CNN of frame g (its features sit in buffer g % NBUF) on the CNN stream"""
        d = g % self.NBUF
        n = self.sb.features_wait(ticket=d)
        self.rows[d] = n
        self.rows_total += n
        if n == 0:
            return
        with torch.cuda.stream(self.B), torch.no_grad():
            self.B.wait_event(self.ev_feat[d])
            timed = self.time_cnn and g % 4 == 0   # (every fourth frame: the two marker packets sit on the schedule's critical path)
            if timed:
                pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                pair[0].record(self.B)
            if self._defer_fixup:
                self.model(self.feat[d][:n], fixup=False, sflags=self.sflags[d], out=self.kp[d][:n])   # (Dense-2 writes the frame's keypoint buffer)
            else:
                self.kp[d][:n].copy_(self.model(self.feat[d][:n]))
            if timed:
                pair[1].record(self.B)
                self._cnn_pairs.append(pair)
            self.ev_cnn[d].record(self.B)

    def _scatter(self, g: int):
        """keypoints of frame g into the tracks that still exist, on the tracker stream behind that frame's CNN"""
        d = g % self.NBUF
        n = self.rows[d]
        if n == 0:
            return
        self.A.wait_event(self.ev_cnn[d])
        if self._defer_fixup:
            with torch.cuda.stream(self.A), torch.no_grad():
                self.model.range_fixup(self.feat[d][:n], self.kp[d], self.sflags[d])
        self.sb.set_keypoints_uid_dev(self.kp[d].data_ptr(), self.owner[d].data_ptr(), self.uid[d].data_ptr(), n)

    def after_step(self):
        """Call once after every `sb.step_dev(...)` (issued on the tracker stream)."""
        d = self.f % self.NBUF
        self.sb.features_async(self.feat[d].data_ptr(), self.owner[d].data_ptr(), self.uid[d].data_ptr(), self.cap, ticket=d)
        self.ev_feat[d].record(self.A)
        self.f += 1
        while self.cnn_done < self.f - 1:           # the CNN of the frame before this one
            self._cnn(self.cnn_done)
            self.cnn_done += 1
        while self.scattered < self.f - self.lag:   # ... and the scatter of the frame `lag` back
            self._scatter(self.scattered)
            self.scattered += 1

    def drain(self):
        """CNN + scatter of every submitted frame that has not had them; afterwards the keypoints are those of the reference loop."""
        while self.cnn_done < self.f:
            self._cnn(self.cnn_done)
            self.cnn_done += 1
        while self.scattered < self.f:
            self._scatter(self.scattered)
            self.scattered += 1
        self.f = self.cnn_done = self.scattered = 0
        self.A.synchronize()
        self.B.synchronize()
        # The asynchronous path cannot recompute a frame that is long scattered.  Keypoints that are MEANINGLESS -- samples outside
        # fp16's range that were not repaired: more than MMW_RANGE_FIXUP_CAP (64) of them in one batch of the 3-frame model, any
        # at all in the single-frame model (no fp32 kernel) -- are an error, not a warning: Keras' fp32 would have been finite.
        if getattr(self.model, "arith", None) == "f16x3" and hasattr(self.model, "range_overflow") and self.model.range_overflow():
            self.range_overflowed = True
            raise PostureRangeError("MarsCNN (split-fp16 arithmetic): inputs or activations left fp16's range during this run and were NOT repaired "
                                    "(the 3-frame model repairs up to 64 samples per batch on the device, the single-frame model none): the keypoints "
                                    "of the samples concerned are meaningless -- use MarsCNN(arith='f32' / 'torch') for such data")

    def close(self):
        """drain(), then hand the tracker back as it was found: its side-stream workers on again if this pipeline turned
        them off.  The pipeline must not be used afterwards."""
        if self._closed:
            return
        try:
            self.drain()
        finally:   # (also when drain() raises PostureRangeError)
            self._closed = True
            if self._side_was_on and self.B is not self.A and self.sb.h:
                self.sb.set_chain_side_stream(True)

    def __del__(self):
        try:
            if not self._closed and self.sb.h and self._side_was_on and self.B is not self.A:
                self.sb.set_chain_side_stream(True)
        except Exception:
            pass
