#!/usr/bin/env python3
"""Diagnostic: phase shares of the BallTree DBSCAN on the start-up frame (4096 clouds of 512 points), from the STAMPS
build (MMW_LIB_NAME=libmmw_hip_stamps.so).  Shares only -- never quote this build's run time."""
import os
import sys

import numpy as np

os.environ["MMW_LIB_NAME"] = "libmmw_hip_stamps.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmwave_msc_amd import _lib  # noqa: E402
from mmwave_msc_amd.batch import SceneBatch  # noqa: E402

S, N, T, F = int(os.environ.get("COLD_S", "4096")), 512, 8, 2
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=16)
sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
bp = sb.alloc(S * N * 64); bn = sb.alloc(S * 4); bd = sb.alloc(S * 8)
for rep in range(2):
    sb.reset(); sb.stats_reset()
    bp.upload(pts[0].astype(np.float64)); bn.upload(cnt[0]); bd.upload(dts[0])
    sb.step_dev(bp.ptr, bn.ptr, bd.ptr)
    sb.synchronize()
out = np.zeros(32, dtype=np.uint64)
sb._chk(sb.L.mmw_stats_get_ext(sb.h, out.ctypes.data))
names_d = ["stage", "tree build (rest)", "centroids+radii", "queries", "labelling", "-", "  build: min/max", "  build: split dim+keys", "  build: rank scan", "  build: partition"]
calls = float(out[3]) - float(out[31])
totd = float(out[20:30].sum())
print(f"k_dbscan: {calls:.0f} calls, mean U {float(out[4]) / max(float(out[3]), 1):.0f}, mean cycles/WG {totd / calls:.0f}; clusters {float(out[7]):.0f}")
for i, nme in enumerate(names_d):
    print(f"  {nme:28s} {float(out[20 + i]) / calls:9.0f} cyc  {100 * float(out[20 + i]) / totd:5.1f} %")
