"""Reference-faithful Python restatement of `TrackBuffer.track` -- the stand-in for
"the reference CPU path" on machines where /root/reference does not exist.

TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg, tests).  It keeps the
reference's *computational shape* on purpose -- numpy arrays, a per-point
`np.linalg.inv`/`np.linalg.det` in the gate (Tracking.py:553-560), and
`sklearn.cluster.DBSCAN` driven by a Python callable metric (Utils.py:272-278) --
so that its frames/s is a fair proxy for the reference's.  Own code, own
structure (plain functions over a small record type; the Kalman algebra written
out instead of filterpy), same arithmetic; pinned against the golden vectors
(tests/test_py_oracle.py).

Restates: Tracking.py:21-71 (BatchedData), 87-97, 120-136, 210-341, 372-398,
513-703; Utils.py:222-291; constants.py:176-246; filterpy 1.4.5 predict/update.
"""
from __future__ import annotations

import math
from collections import deque

import numpy as np
from scipy.linalg import block_diag
from sklearn.cluster import DBSCAN


class Params:
    """constants.py values used by the tracker (defaults = constants.py)."""

    S_HEIGHT, S_TILT = 1.8, -5                      # constants.py:41-42
    INTENSITY_MU, INTENSITY_STD = 27.0187, 70.351   # constants.py:108-109
    MODEL_MIN_INPUT = 0                             # constants.py:111

    def __init__(self, **kw):
        self.FB_FRAMES_BATCH = 2
        self.DB_Z_WEIGHT, self.DB_RANGE_WEIGHT, self.DB_EPS, self.DB_MIN_SAMPLES_MIN = 0.4, 0.03, 0.3, 35
        self.TR_MAX_TRACKS, self.TR_LIFETIME_DYNAMIC, self.TR_LIFETIME_STATIC = 4, 3, 7
        self.TR_VEL_THRES, self.TR_GATE = 0.12, 4.5
        self.KF_Q_STD, self.KF_P_INIT, self.KF_GROUP_DISP_EST_INIT = 1, 0.1, 0.1
        self.KF_ENABLE_EST, self.KF_A_N, self.KF_EST_POINTNUM = False, 0.9, 10
        self.KF_SPREAD_LIM, self.KF_A_SPR = [0.2, 0.2, 2, 1.2, 1.2, 0.2], 0.9
        self.DIM_X = 9
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


def _q3(dt, var):
    return np.array([[0.25 * dt**4, 0.5 * dt**3, 0.5 * dt**2],
                     [0.5 * dt**3, dt**2, dt],
                     [0.5 * dt**2, dt, 1]]) * var


def _F(p, dt):
    f = np.eye(p.DIM_X)
    for i in range(3):
        f[i, i + 3] = dt
    if p.DIM_X == 9:
        for i in range(3):
            f[i, i + 6] = 0.5 * dt**2
            f[i + 3, i + 6] = dt
    return f


def _Q(p, dt):
    return block_diag(*[_q3(dt, p.KF_Q_STD)] * (p.DIM_X // 3))


def _H(p):
    return np.eye(6, p.DIM_X)


class Ring:
    def __init__(self, size, first):
        self.size = size
        self.buffer = deque([first], maxlen=size)
        self.effective_data = np.concatenate(list(self.buffer), axis=0)

    def add_frame(self, frame):
        while len(self.buffer) >= self.size:
            self.buffer.popleft()
        self.buffer.append(frame)
        self.effective_data = np.concatenate(list(self.buffer), axis=0)

    def clear(self):
        self.buffer.clear()
        self.effective_data = np.array([])


class Track:
    pass


def _cluster_stats(p, t, cloud):
    t.cloud = cloud
    t.point_num = cloud.shape[0]
    t.centroid = np.mean(cloud[:, :6], axis=0)
    t.min_vals = np.min(cloud[:, :6], axis=0)
    t.max_vals = np.max(cloud[:, :6], axis=0)
    t.is_static = math.sqrt(np.sum(t.centroid[3:6] ** 2)) < p.TR_VEL_THRES


def _new_track(p, cloud):
    t = Track()
    _cluster_stats(p, t, cloud)
    t.N_est = 0
    t.spread_est = np.zeros(6)
    t.group_disp_est = np.eye(6) * p.KF_GROUP_DISP_EST_INIT
    t.batch = Ring(p.FB_FRAMES_BATCH + 1, cloud)
    t.x = np.array([list(t.centroid) + [0] * (p.DIM_X - 6)], dtype=float).T
    t.P = np.eye(p.DIM_X) * p.KF_P_INIT
    t.lifetime = 0
    return t


def _rm(t):
    return np.diag((t.spread_est / 2) ** 2)


def _associate(p, t, cloud):
    _cluster_stats(p, t, cloud)
    t.batch.add_frame(cloud)
    n = t.point_num
    if p.KF_ENABLE_EST:
        t.N_est = n if n > t.N_est else (1 - p.KF_A_N) * t.N_est + p.KF_A_N * n
    else:
        t.N_est = max(p.KF_EST_POINTNUM, n)
    for m in range(6):
        spread = t.max_vals[m] - t.min_vals[m]
        if n != 1:
            spread = spread * (n + 1) / (n - 1)
        spread = min(2 * p.KF_SPREAD_LIM[m], spread)
        spread = max(p.KF_SPREAD_LIM[m], spread)
        if spread > t.spread_est[m]:
            t.spread_est[m] = spread
        else:
            t.spread_est[m] = (1.0 - p.KF_A_SPR) * t.spread_est[m] + p.KF_A_SPR * spread
    disp = np.zeros((6, 6))
    for i in range(6):
        for j in range(6):
            disp[i, j] = np.mean((cloud[:, i] - t.centroid[i]) * (cloud[:, j] - t.centroid[j]))
    a = n / t.N_est
    t.group_disp_est = (1 - a) * t.group_disp_est + a * disp


def _predict(p, t, dt):
    f = _F(p, dt)
    t.x = np.dot(f, t.x)
    t.P = 1.0 * np.dot(np.dot(f, t.P), f.T) + _Q(p, dt)


def _update(p, t):
    h = _H(p)
    z = np.array(t.centroid).reshape(6, 1)
    n, n_est = t.point_num, t.N_est
    r = (_rm(t) / n) + ((n_est - n) / ((n_est - 1) * n)) * t.group_disp_est
    y = z - np.dot(h, t.x)
    pht = np.dot(t.P, h.T)
    s = np.dot(h, pht) + r
    k = np.dot(pht, np.linalg.inv(s))
    t.x = t.x + np.dot(k, y)
    ikh = np.eye(p.DIM_X) - np.dot(k, h)
    t.P = np.dot(np.dot(ikh, t.P), ikh.T) + np.dot(np.dot(k, r), k.T)
    var = z[:1, 0] - t.x[:1, 0]
    if abs(var.any()) > 0.6 and t.lifetime == 0:
        t.x[:1, 0] += var * 0.4


class PyScene:
    """One TrackBuffer + its global BatchedData."""

    def __init__(self, params: Params | None = None):
        self.p = params or Params()
        self.tracks = []
        self.batch = Ring(self.p.FB_FRAMES_BATCH + 1, np.empty((0, 8)))
        p = self.p

        def metric(p1, p2):
            w = 1 - ((p1[1] + p2[1]) / 2) * p.DB_RANGE_WEIGHT
            return w * ((p1[0] - p2[0]) ** 2 + (p1[1] - p2[1]) ** 2 + p.DB_Z_WEIGHT * ((p1[2] - p2[2]) ** 2))

        self._metric = metric

    def _gate(self, cloud):
        p = self.p
        assoc = np.full(cloud.shape[0], -1, dtype=np.int32)
        dist = np.empty((cloud.shape[0], len(self.tracks)))
        h = _H(p)
        for j, t in enumerate(self.tracks):
            hx = np.dot(h, t.x).flatten()
            c = t.P[:6, :6] + _rm(t) + t.group_disp_est
            for i, pt in enumerate(cloud):
                y = np.array(pt[:6]) - hx
                dist[i][j] = np.log(np.abs(np.linalg.det(c))) + np.dot(np.dot(y.T, np.linalg.inv(c)), y)
                if dist[i][j] < p.TR_GATE:
                    if assoc[i] < 0 or dist[i][j] < dist[i][assoc[i]]:
                        assoc[i] = j
        return assoc

    def track(self, cloud, dt):
        """Returns (assoc, dbscan labels or None)."""
        p = self.p
        cloud = np.asarray(cloud, dtype=np.float64).reshape(-1, 8)
        for t in self.tracks:
            _predict(p, t, t.lifetime + dt)
        assoc = self._gate(cloud)
        unassigned = np.empty((0, 8))
        groups = [[] for _ in self.tracks]
        for i, pt in enumerate(cloud):
            if assoc[i] < 0:
                unassigned = np.append(unassigned, [pt], axis=0)
            else:
                groups[assoc[i]].append(pt)
        for j, t in enumerate(self.tracks):
            if len(groups[j]) == 0:
                t.lifetime += dt
            else:
                t.lifetime = 0
                _associate(p, t, np.array(groups[j]))
        self.tracks[:] = [t for t in self.tracks
                          if not (t.lifetime > (p.TR_LIFETIME_STATIC if t.is_static else p.TR_LIFETIME_DYNAMIC))]
        for t in self.tracks:
            _update(p, t)
        self.batch.add_frame(unassigned)
        labels = None
        if len(self.batch.effective_data) > 0 and len(self.tracks) < p.TR_MAX_TRACKS:
            data = self.batch.effective_data
            labels = DBSCAN(eps=p.DB_EPS, min_samples=p.DB_MIN_SAMPLES_MIN, metric=self._metric).fit_predict(data)
            found = sorted(set(labels) - {-1})
            if found:
                self.batch.clear()
            for lab in found:
                self.tracks.append(_new_track(p, np.array([data[i] for i in range(len(labels)) if labels[i] == lab])))
            labels = labels.astype(np.int32)
        return assoc, labels

    @property
    def n_tracks(self):
        return len(self.tracks)


def py_normalize(p: Params, det) -> np.ndarray:
    """Utils.normalize_data + point_transform_to_standard_axis (Utils.py:294-434) with the reference's computational shape: a
    Python loop over the points, two 4 x 4 products per vector, rows appended one at a time.  det: dict of equal-length
    sequences x, y, z, doppler, peakVal.  Returns (N', 8) float64."""
    rows = np.vstack((det["x"], det["y"], det["z"], det["doppler"], det["peakVal"])).T
    shift = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, p.S_HEIGHT], [0, 0, 0, 1]])
    a = np.radians(p.S_TILT)
    rot = np.array([[1, 0, 0, 0], [0, np.cos(a), -np.sin(a), 0], [0, np.sin(a), np.cos(a), 0], [0, 0, 0, 1]])
    out = np.empty((0, 8), dtype="float")
    for k in range(len(rows)):
        x, y, z, dop, peak = rows[k]
        r = math.sqrt(x ** 2 + y ** 2 + z ** 2)
        vel = (0, dop, 0) if r == 0 else (dop * x / r, dop * y / r, dop * z / r)
        pos_t = np.dot(shift, np.dot(rot, np.array([x, y, z, 1])))
        vel_t = np.dot(shift, np.dot(rot, np.array([vel[0], vel[1], vel[2], 0])))
        row = np.append(np.array([pos_t[0], pos_t[1], pos_t[2], vel_t[0], vel_t[1], vel_t[2]]), (dop, peak))
        if row[2] <= 2.5 and row[2] > 0 and row[1] > 0:
            out = np.append(out, [row], axis=0)
    return out


def py_feature_maps(p: Params, sc: "PyScene"):
    """The feature side of TrackBuffer.estimate_posture (Tracking.py:718-730): relative_coordinates + format_single_frame
    (Utils.py:437-520) per eligible track, in the reference's shape -- a Python loop over the points for the shift, per frame
    column pick, intensity normalisation, pad / cut to 64 rows, argsort by x.  Returns (maps[B, ring, 8, 8, 5] float64, owners)."""
    ring = p.FB_FRAMES_BATCH + 1
    maps, owners = [], []
    for j, t in enumerate(sc.tracks):
        if len(t.batch.effective_data) <= p.MODEL_MIN_INPUT:
            continue
        ref = [t.centroid[0], t.centroid[1], 0, 0, 0, 0, 0, 0]
        grid = np.zeros((ring, 64, 5))
        for k, frame in enumerate(t.batch.buffer):
            rel = np.array([pt - ref for pt in frame])
            sel = rel[:, [0, 1, 2, -2, -1]]
            sel[:, 4] = (sel[:, 4] - p.INTENSITY_MU) / p.INTENSITY_STD
            sel = np.concatenate((sel, np.zeros((64 - len(sel), 5))), axis=0) if len(sel) < 64 else sel[:64]
            grid[k] = sel[np.argsort(sel[:, 0])]
        maps.append(grid.reshape((64, 5)).reshape((8, 8, 5)) if ring == 1 else grid.reshape((ring, 8, 8, 5)))
        owners.append(j)
    return np.array(maps), owners


def py_estimate_posture(p: Params, sc: "PyScene", model):
    """TrackBuffer.estimate_posture (Tracking.py:705-734): `model` is anything with a Keras-style .predict."""
    maps, owners = py_feature_maps(p, sc)
    if len(maps) > 0:
        kp = model.predict(maps)
        for i, j in enumerate(owners):
            sc.tracks[j].keypoints = kp[i]
    return len(owners)


def _worker(args):
    params_kw, frames, counts, dts = args
    out = []
    for s in range(frames.shape[1]):
        sc = PyScene(Params(**params_kw))
        for f in range(frames.shape[0]):
            c = int(counts[f, s])
            if c:
                sc.track(frames[f, s, :c].astype(np.float64), float(dts[f, s]))
        out.append(sc.n_tracks)
    return out


def run_batch_multiprocess(params_kw, pts, cnt, dts, procs):
    """Scenes sharded evenly over `procs` worker processes.  pts[F,S,N,8].  Returns wall seconds."""
    import multiprocessing as mp
    import time

    S = pts.shape[1]
    procs = max(1, min(int(procs), S))
    shards = np.array_split(np.arange(S), procs)
    jobs = [(params_kw, pts[:, sh], cnt[:, sh], dts[:, sh]) for sh in shards if len(sh)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    if procs == 1:
        res = [_worker(jobs[0])]
    else:
        with ctx.Pool(len(jobs)) as pool:
            res = pool.map(_worker, jobs)
    el = time.perf_counter() - t0
    return el, [x for r in res for x in r]


def _window_proc(idx, q, params_kw, frames, counts, dts, warm, barrier, posture_weights=None):
    import time
    try:  # one BLAS thread per process: the processes ARE the parallelism (numpy is already imported in the parent)
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    scenes = [PyScene(Params(**params_kw)) for _ in range(frames.shape[1])]
    model = None
    if posture_weights is not None:   # estimate_posture after every track() (offline_main.py:57-60): Keras' fp32 on this core
        import torch
        from .mars_torch import MarsTorchCPU
        model = MarsTorchCPU(posture_weights, torch.float32, threads=1)
    rows = 0

    def run(f0, f1):
        nonlocal rows
        for s, sc in enumerate(scenes):
            for f in range(f0, f1):
                c = int(counts[f, s])
                if c:
                    sc.track(frames[f, s, :c].astype(np.float64), float(dts[f, s]))
                    if model is not None:
                        rows += py_estimate_posture(sc.p, sc, model)

    try:
        run(0, warm)
        barrier.wait()
        rows = 0
        t0 = time.perf_counter()
        run(warm, frames.shape[0])
        t1 = time.perf_counter()
        q.put((idx, t0, t1, [sc.n_tracks for sc in scenes], rows))
    except BaseException as exc:  # a dead worker must not leave the others at the barrier
        barrier.abort()
        q.put((idx, None, None, repr(exc), 0))


def run_window_multiprocess(params_kw, pts, cnt, dts, procs, warm, posture_weights=None, want_rows=False):
    """The bench's CPU baseline: scenes sharded evenly over `procs` processes (started, and the first `warm`
    frames of every scene tracked, BEFORE the clock starts; a barrier lines the processes up), then frames
    warm.. timed.  Wall = last finish - first start on the shared monotonic clock.  pts[F,S,N,8].
    `posture_weights` (Keras-layout dict): estimate_posture after every track(), the CNN on torch's CPU operators.
    Returns (wall seconds, final track counts by scene[, feature tensors pushed through the CNN in the window])."""
    import multiprocessing as mp

    S = pts.shape[1]
    procs = max(1, min(int(procs), S))
    shards = [sh for sh in np.array_split(np.arange(S), procs) if len(sh)]
    ctx = mp.get_context("fork")
    barrier = ctx.Barrier(len(shards))
    q = ctx.Queue()
    ps = [ctx.Process(target=_window_proc, args=(i, q, params_kw, pts[:, sh], cnt[:, sh], dts[:, sh], int(warm), barrier, posture_weights))
          for i, sh in enumerate(shards)]
    for p in ps:
        p.start()
    res = [q.get() for _ in ps]
    for p in ps:
        p.join()
    bad = [r for r in res if r[1] is None]
    if bad:
        raise RuntimeError(f"cpu baseline worker failed: {bad[0][3]}")
    res.sort(key=lambda r: r[0])
    wall = max(r[2] for r in res) - min(r[1] for r in res)
    counts = [x for r in res for x in r[3]]
    return (wall, counts, sum(r[4] for r in res)) if want_rows else (wall, counts)
