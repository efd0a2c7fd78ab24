#!/usr/bin/env python3
"""Diagnostic: frames 0..3 of fresh scenes (every scene clusters its ring), twice; meant to run under
`rocprofv3 --kernel-trace` (scripts/cold_trace.sh prints the launches of the second pass in order)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmwave_msc_amd import _lib  # noqa: E402
from mmwave_msc_amd.batch import SceneBatch  # noqa: E402

S, N, T, F = int(os.environ.get("COLD_S", "4096")), 512, 8, 4
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=int(os.environ.get("COLD_WORKERS", "1")))
print("points per frame: mean", cnt.mean(axis=1), "share > 256:", (cnt > 256).mean(axis=1))
sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
bp = [sb.alloc(S * N * 64) for _ in range(F)]; bn = [sb.alloc(S * 4) for _ in range(F)]; bd = [sb.alloc(S * 8) for _ in range(F)]
for f in range(F):
    bp[f].upload(pts[f].astype(np.float64)); bn[f].upload(cnt[f]); bd[f].upload(dts[f])
for rep in range(2):
    sb.reset(); sb.stats_reset()
    sb.synchronize()
    for f in range(F):
        sb.step_dev(bp[f].ptr, bn[f].ptr, bd[f].ptr)
    sb.synchronize()
st = sb.stats()
print("dbscan calls", int(st[3]), "mean U", float(st[4]) / max(float(st[3]), 1), "clusters", int(st[7]))
