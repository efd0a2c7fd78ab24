#!/usr/bin/env python3
"""Diagnostic: where k_track / k_dbscan spend their cycles, from the STAMPS build
(MMW_LIB_NAME=libmmw_hip_stamps.so).  Shares only -- never quote this build's run time."""
import os
import sys

import numpy as np

os.environ["MMW_LIB_NAME"] = "libmmw_hip_stamps.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmwave_msc_amd import _lib  # noqa: E402
from mmwave_msc_amd.batch import SceneBatch  # noqa: E402

S, N, T, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 512, 8, int(os.environ.get("FRAMES", "60"))
RESET_AT = int(os.environ.get("RESET_AT", "10"))   # frame at which the counters are cleared (FRAMES=160 RESET_AT=100: the window in which scenes that lost tracks re-cluster)
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=16)
sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
bp = sb.alloc(S * N * 64); bn = sb.alloc(S * 4); bd = sb.alloc(S * 8)
for f in range(F):
    if f == RESET_AT:
        sb.stats_reset()
    bp.upload(pts[f].astype(np.float64)); bn.upload(cnt[f]); bd.upload(dts[f])
    sb.step_dev(bp.ptr, bn.ptr, bd.ptr)
sb.synchronize()
out = np.zeros(32, dtype=np.uint64)
sb._chk(sb.L.mmw_stats_get_ext(sb.h, out.ctypes.data))
names_t = ["stage points", "gate (records as SGPR operands)", "gating", "class split", "centroid/minmax/spread", "dispersion D",
           "track-ring rows+barrier", "maintenance", "update", "global ring append", "DBSCAN screens + push"]
frames = float(out[2])
tot = float(out[8:19].sum())
print(f"k_track: {frames:.0f} scene-frames, mean cycles/WG {tot / frames:.0f}")
for i, nme in enumerate(names_t):
    print(f"  {nme:28s} {float(out[8 + i]) / frames:9.0f} cyc  {100 * float(out[8 + i]) / tot:5.1f} %")
names_d = ["stage", "tree build (rest)", "centroids+radii", "queries", "labelling", "-", "  build: min/max", "  build: split dim+keys", "  build: rank scan", "  build: partition"]
calls = float(out[3])
print(f"k_dbscan early exits (screen or in-kernel): {float(out[31]):.0f} of {calls:.0f} calls; clusters found {float(out[7]):.0f}")
calls -= float(out[31])  # the phase stamps below only exist for calls that built the BallTree
totd = float(out[20:30].sum())
if calls and totd:
    print(f"k_dbscan: {calls:.0f} calls, mean U {float(out[4]) / calls:.0f}, mean cycles/WG {totd / calls:.0f}")
    for i, nme in enumerate(names_d):
        print(f"  {nme:28s} {float(out[20 + i]) / calls:9.0f} cyc  {100 * float(out[20 + i]) / totd:5.1f} %")
