"""Seeded synthetic radar scenes (SURVEY.md §8d "Synthetic inputs").

The reference ships no data (its `.gitignore` excludes every log/dataset), so
every benchmark/parity input is generated here.  One scene = K walking targets
plus uniform clutter, already in the *normalised* 8-column layout that
`TrackBuffer.track` consumes (reference `Utils.normalize_data` output,
Utils.py:342-434): x, y, z, vx, vy, vz, doppler, peakVal.

All values are rounded to fp32-representable numbers so that fp32 storage ->
fp64 promotion is exact on every path (GPU, oracle, reference).
"""
from __future__ import annotations

import numpy as np

FRAME_DT = 0.1  # s, radar frame period (reference config_cases/our_config_8.5m.cfg:29)


def _place_targets(rng, k, min_sep=1.5):
    pts = []
    tries = 0
    while len(pts) < k:
        c = np.array([rng.uniform(-2.0, 2.0), rng.uniform(1.5, 6.0)])
        tries += 1
        if all(np.hypot(*(c - p)) >= min_sep for p in pts) or tries > 2000:
            pts.append(c)
    return np.array(pts).reshape(k, 2)


def make_scene(scene_id: int, n_frames: int, n_pts: int, n_targets: int,
               ragged: bool = False, presence=None, dt_seq=None):
    """Returns (points[F, n_pts, 8] float32, counts[F] int32, dt[F] float64).

    `counts[f]` valid rows per frame (== n_pts unless `ragged`); rows past the
    count are zero.  `presence[F, K]` (bool) switches a target off for some
    frames (its points are scattered as clutter instead); `dt_seq[F]` replaces
    the constant 100 ms frame period.
    """
    rng = np.random.default_rng(int(scene_id))
    k = int(n_targets)
    f = int(n_frames)
    per_t = int(np.floor(0.9 * n_pts / k)) if k > 0 else 0
    n_clutter = n_pts - per_t * k
    c0 = _place_targets(rng, k) if k > 0 else np.zeros((0, 2))
    vel = rng.normal(0.0, 0.5, size=(k, 2))
    dt = np.full(f, FRAME_DT, dtype=np.float64) if dt_seq is None else np.asarray(dt_seq, dtype=np.float64)
    t = (np.cumsum(dt) - dt[0])[:, None, None]
    centre = c0[None] + vel[None] * t  # (F, K, 2)
    out = np.zeros((f, n_pts, 8), dtype=np.float64)
    if k > 0:
        tp = np.empty((f, k, per_t, 8))
        tp[..., 0:2] = centre[:, :, None, :] + rng.normal(0.0, 0.15, size=(f, k, per_t, 2))
        tp[..., 2] = rng.uniform(0.05, 1.8, size=(f, k, per_t))
        tp[..., 3:5] = vel[None, :, None, :] + rng.normal(0.0, 0.05, size=(f, k, per_t, 2))
        tp[..., 5] = rng.normal(0.0, 0.05, size=(f, k, per_t))
        if presence is not None:
            gone = ~np.asarray(presence, dtype=bool).reshape(f, k)
            g = np.broadcast_to(gone[:, :, None], (f, k, per_t))
            tp[..., 0][g] = rng.uniform(-3.0, 3.0, size=int(g.sum()))
            tp[..., 1][g] = rng.uniform(0.2, 8.0, size=int(g.sum()))
            tp[..., 2][g] = rng.uniform(0.05, 2.4, size=int(g.sum()))
            for col in (3, 4, 5):
                tp[..., col][g] = rng.normal(0.0, 0.05, size=int(g.sum()))
        out[:, : k * per_t] = tp.reshape(f, k * per_t, 8)
    cl = out[:, k * per_t:]
    cl[..., 0] = rng.uniform(-3.0, 3.0, size=(f, n_clutter))
    cl[..., 1] = rng.uniform(0.2, 8.0, size=(f, n_clutter))
    cl[..., 2] = rng.uniform(0.05, 2.4, size=(f, n_clutter))
    cl[..., 3:6] = rng.normal(0.0, 0.05, size=(f, n_clutter, 3))
    out[..., 6] = rng.normal(0.0, 0.3, size=(f, n_pts))
    out[..., 7] = rng.gamma(1.0, 30.0, size=(f, n_pts))
    # keep every point inside the scene filter of normalize_data (z in (0, 2.5], y > 0)
    out[..., 1] = np.maximum(out[..., 1], 0.05)
    perm = np.argsort(rng.random(size=(f, n_pts)), axis=1)
    out = np.take_along_axis(out, perm[:, :, None], axis=1)
    pts32 = out.astype(np.float32)
    counts = np.full(f, n_pts, dtype=np.int32)
    if ragged:
        counts = rng.integers(n_pts // 2, n_pts + 1, size=f).astype(np.int32)
        for i in range(f):
            pts32[i, counts[i]:] = 0.0
    return pts32, counts, dt


def make_pair_scene(scene_id: int, n_frames: int, n_pts: int, n_pairs: int = 1, sep: float = 0.8, static: bool = False):
    """Scenes for `seek_inner_clusters` (reference Tracking.py:409-448): `n_pairs` pairs of people walking side by
    side, `sep` metres apart -- one DBSCAN cluster at DB_EPS, two at DB_INNER_EPS -- plus 10 % uniform clutter.
    `static` pairs stand still (cluster status STATIC).  Returns (points[F, n_pts, 8] float32, counts[F], dt[F])."""
    rng = np.random.default_rng(int(scene_id))
    f, k = int(n_frames), int(n_pairs)
    per = int(0.45 * n_pts / k)
    c0 = _place_targets(rng, k, min_sep=2.2)
    vel = np.zeros((k, 2)) if static else rng.normal(0.0, 0.3, size=(k, 2))
    out = np.zeros((f, n_pts, 8), dtype=np.float64)
    for i in range(f):
        row = 0
        for p in range(k):
            c = c0[p] + vel[p] * FRAME_DT * i
            for off in (-sep / 2, sep / 2):
                blk = out[i, row: row + per]
                blk[:, 0] = c[0] + off + rng.normal(0.0, 0.08, per)
                blk[:, 1] = c[1] + rng.normal(0.0, 0.08, per)
                blk[:, 2] = rng.uniform(0.6, 1.4, per)
                blk[:, 3:5] = vel[p] + rng.normal(0.0, 0.02 if static else 0.05, (per, 2))
                blk[:, 5] = rng.normal(0.0, 0.02 if static else 0.05, per)
                row += per
        cl = out[i, row:]
        m = n_pts - row
        cl[:, 0] = rng.uniform(-3.0, 3.0, m)
        cl[:, 1] = rng.uniform(0.2, 8.0, m)
        cl[:, 2] = rng.uniform(0.05, 2.4, m)
        cl[:, 3:6] = rng.normal(0.0, 0.05, (m, 3))
        out[i, :, 6] = rng.normal(0.0, 0.3, n_pts)
        out[i, :, 7] = rng.gamma(1.0, 30.0, n_pts)
        out[i] = out[i][rng.permutation(n_pts)]
    out[..., 1] = np.maximum(out[..., 1], 0.05)
    return out.astype(np.float32), np.full(f, n_pts, dtype=np.int32), np.full(f, FRAME_DT, dtype=np.float64)


def make_batch(scene_ids, n_frames: int, n_pts: int, n_targets: int, ragged: bool = False):
    """Stack scenes: points[F, S, n_pts, 8] float32, counts[F, S] int32, dt[F, S] float64."""
    ps, cs, ds = [], [], []
    for sid in scene_ids:
        p, c, d = make_scene(sid, n_frames, n_pts, n_targets, ragged)
        ps.append(p); cs.append(c); ds.append(d)
    return (np.ascontiguousarray(np.stack(ps, axis=1)),
            np.ascontiguousarray(np.stack(cs, axis=1)),
            np.ascontiguousarray(np.stack(ds, axis=1)))
