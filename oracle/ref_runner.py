"""Drive the REAL reference (imported from /root/reference/src, container-only)
behind the same small interface as `c_oracle.OracleScene`, with hooks that
record the integer decisions (association vector, DBSCAN labels).

TEST INFRASTRUCTURE ONLY.  Used by oracle/gen_golden.py and by the
container-only pinning tests (skipped when /root/reference is absent).
"""
from __future__ import annotations

import numpy as np

from .c_oracle import TRACK_DTYPE
from .ref_import import load_reference


class _Recorder:
    assoc = None
    labels = None
    inner = []      # seek_inner_clusters calls of the frame: (pre-maintenance track position, labels)


def _enable_seek_inner(tracking):
    """Tracking.py:656 (`# new_inner_clusters.append(track.seek_inner_clusters())`) is commented out in the reference.
    For the seek_inner goldens the method is re-compiled from the reference's OWN source text with exactly that line
    un-commented (at generation time, in this container; nothing of it is stored), and `seek_inner_clusters` is wrapped
    so that the labels of its apply_DBscan call are recorded per call instead of being mistaken for the frame's."""
    import inspect
    import textwrap
    if getattr(tracking, "_mmw_inner", False):
        return
    src = textwrap.dedent(inspect.getsource(tracking.TrackBuffer._associate_points_to_tracks))
    marker = "# new_inner_clusters.append(track.seek_inner_clusters())"
    assert src.count(marker) == 1, "the reference's call site moved"
    src = src.replace(marker, "new_inner_clusters.append(track.seek_inner_clusters())")
    ns = {}
    exec(compile(src, "<Tracking.py:_associate_points_to_tracks with line 656 active>", "exec"), tracking.__dict__, ns)
    tracking.TrackBuffer._associate_points_to_tracks_shipped = tracking.TrackBuffer._associate_points_to_tracks
    tracking.TrackBuffer._associate_points_to_tracks_inner = ns["_associate_points_to_tracks"]
    orig = tracking.ClusterTrack.seek_inner_clusters

    def seek(self):
        saved, _Recorder.labels = _Recorder.labels, None
        out = orig(self)
        if _Recorder.labels is not None:
            _Recorder.inner.append((_Recorder.current_tracks.index(self), _Recorder.labels))
        _Recorder.labels = saved
        return out

    tracking.ClusterTrack.seek_inner_clusters = seek
    tracking._mmw_inner = True


def _install_hooks(utils, tracking):
    if getattr(tracking, "_mmw_hooked", False):
        return
    base_dbscan = utils.DBSCAN

    class RecordingDBSCAN(base_dbscan):
        def fit_predict(self, X, y=None, sample_weight=None):
            lab = super().fit_predict(X, y=y, sample_weight=sample_weight)
            _Recorder.labels = np.asarray(lab, dtype=np.int32).copy()
            return lab

    utils.DBSCAN = RecordingDBSCAN
    orig = tracking.TrackBuffer._calc_dist_fun

    def calc(self, full_set):
        a = orig(self, full_set)
        _Recorder.assoc = np.array([-1 if v is None else int(v) for v in a], dtype=np.int32)
        return a

    tracking.TrackBuffer._calc_dist_fun = calc
    tracking._mmw_hooked = True


class RefScene:
    def __init__(self, overrides=None, init_data=None):
        self.const, self.utils, self.tracking = load_reference()
        _install_hooks(self.utils, self.tracking)
        self._saved = {}
        overrides = dict(overrides or {})
        self.seek_inner = bool(overrides.pop("SEEK_INNER", False))
        if self.seek_inner:
            _enable_seek_inner(self.tracking)
        for k, v in overrides.items():
            self._saved[k] = getattr(self.const, k)
            setattr(self.const, k, v)
        self.tb = self.tracking.TrackBuffer()
        self.batch = self.tracking.BatchedData() if init_data is None else self.tracking.BatchedData(np.asarray(init_data, dtype=np.float64))

    def close(self):
        for k, v in self._saved.items():
            setattr(self.const, k, v)
        self._saved = {}

    def track(self, pts, dt):
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 8)
        _Recorder.assoc = np.full(pts.shape[0], -1, dtype=np.int32)
        _Recorder.labels = None
        _Recorder.inner = []
        _Recorder.current_tracks = list(self.tb.effective_tracks)
        tb_cls = self.tracking.TrackBuffer
        if hasattr(tb_cls, "_associate_points_to_tracks_inner"):
            tb_cls._associate_points_to_tracks = (tb_cls._associate_points_to_tracks_inner if self.seek_inner
                                                  else tb_cls._associate_points_to_tracks_shipped)
        self.tb.dt = dt
        self.tb.track(pts, self.batch)
        self.inner = list(_Recorder.inner)
        return _Recorder.assoc.copy(), (None if _Recorder.labels is None else _Recorder.labels.copy())

    @property
    def n_tracks(self):
        return len(self.tb.effective_tracks)

    def tracks(self):
        out = np.zeros(self.n_tracks, dtype=TRACK_DTYPE)
        for j, t in enumerate(self.tb.effective_tracks):
            dx = t.state.x.shape[0]
            out[j]["x"][:dx] = t.state.x[:, 0]
            out[j]["P"][:dx, :dx] = t.state.P
            out[j]["centroid"] = t.cluster.centroid
            out[j]["min_vals"] = t.cluster.min_vals
            out[j]["max_vals"] = t.cluster.max_vals
            out[j]["spread_est"] = t.spread_est
            out[j]["group_disp_est"] = t.group_disp_est
            out[j]["n_est"] = t.N_est
            out[j]["lifetime"] = t.lifetime
            out[j]["point_num"] = t.cluster.point_num
            out[j]["is_static"] = int(bool(t.cluster.status))
            out[j]["ring_len"] = len(t.batch.buffer)
            for k, fr in enumerate(t.batch.buffer):
                out[j]["ring_n"][k] = len(fr)
            out[j]["keypoints"] = np.asarray(t.keypoints, dtype=np.float32)
        return out

    def track_ring_sizes(self):
        return np.array([t.batch.size for t in self.tb.effective_tracks], dtype=np.int32)

    def batch_ring(self):
        return np.array([len(f) for f in self.batch.buffer], dtype=np.int32)

    def features(self):
        """estimate_posture's feature side (Tracking.py:718-730) without the model."""
        feats, owner = [], []
        for idx, t in enumerate(self.tb.effective_tracks):
            if len(t.batch.effective_data) > self.const.MODEL_MIN_INPUT:
                rel = self.utils.relative_coordinates(list(t.batch.buffer), t.cluster.centroid[:2])
                feats.append(self.utils.format_single_frame(rel))
                owner.append(idx)
        if not feats:
            return np.zeros((0,), dtype=np.float32), np.zeros((0,), dtype=np.int32)
        return np.array(feats).astype(np.float32), np.array(owner, dtype=np.int32)

    def estimate_posture(self, model):
        self.tb.estimate_posture(model)
