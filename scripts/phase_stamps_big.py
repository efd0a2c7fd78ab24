#!/usr/bin/env python3
"""Diagnostic: phase shares of apply_DBscan on 1536-point clouds (the generic strided build of k_dbscan_big / k_chain: a scene
without tracks clusters its whole ring), STAMPS build.  Clutter + a few faint targets so that the in-kernel screen does not
end the call.  Shares only."""
import os, sys
import numpy as np
os.environ["MMW_LIB_NAME"] = "libmmw_hip_stamps.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch
S, N, F = 256, 512, 6
rng = np.random.default_rng(3)
pts = np.zeros((F, S, N, 8))
pts[..., 0] = rng.uniform(-6, 6, size=(F, S, N)); pts[..., 1] = rng.uniform(0.3, 7.5, size=(F, S, N)); pts[..., 2] = rng.uniform(0.05, 2.4, size=(F, S, N))
for s in range(S):   # a faint target: 14 points a frame -> a cluster only in the whole ring
    for f in range(F):
        idx = rng.choice(N, size=14, replace=False)
        pts[f, s, idx, 0:2] = np.array([1.0, 3.0]) + rng.normal(0, 0.06, size=(14, 2))
        pts[f, s, idx, 2] = rng.uniform(0.6, 1.4, size=14)
cnt = np.full((F, S), N, np.int32); dts = np.full((F, S), 0.1)
sb = SceneBatch(_lib.default_config(tr_max_tracks=4, chain_side_stream=-1), S, N)
WHICH = int(os.environ.get("FRAME", "2"))   # 1: 256 trees of 1024 points, a cluster each; 2: ~55 trees of 1536 points, none
for f in range(F):
    if f == WHICH:
        sb.synchronize(); sb.stats_reset()
    sb.step_host(pts[f], cnt[f], dts[f])
    if f == WHICH:
        break
sb.synchronize()
out = np.zeros(32, dtype=np.uint64)
sb._chk(sb.L.mmw_stats_get_ext(sb.h, out.ctypes.data))
names_d = ["stage (+ in-kernel screen)", "tree build (rest)", "centroids+radii", "queries", "labelling", "-", "  build: min/max", "  build: split dim+keys", "  build: rank scan", "  build: partition"]
calls = float(out[3]) - float(out[31])
totd = float(out[20:30].sum())
print(f"apply_DBscan calls that built a tree: {calls:.0f} (of {float(out[3]):.0f}), mean U {float(out[4]) / max(float(out[3]), 1):.0f}, mean cycles per cloud {totd / max(calls, 1):.0f}")
for i, nme in enumerate(names_d):
    print(f"  {nme:28s} {float(out[20 + i]) / max(calls, 1):11.0f} cyc  {100 * float(out[20 + i]) / max(totd, 1):5.1f} %")
