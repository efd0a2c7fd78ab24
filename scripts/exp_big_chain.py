#!/usr/bin/env python3
"""Diagnostic: the lone big-cloud BallTree chain (k_dbscan_big with a handful of scenes, i.e. one workgroup per CU
and nothing else on the chip), per start-up frame: duration from the launch-attached events and, with
MMW_LIB_NAME=libmmw_hip_stamps.so, the per-phase cycle shares.  usage: exp_big_chain.py [scenes [points [targets]]] -- 8 512 8 is the big chain, 8 128 2 the small one (k_post worker)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmwave_msc_amd import _lib  # noqa: E402
from mmwave_msc_amd.batch import SceneBatch  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8
F = 6
stamps = "stamps" in os.environ.get("MMW_LIB_NAME", "")
ids = np.arange(S) * T + (T - 1)  # T targets per scene
pts, cnt, dts = bench.generate(ids, F, N, T, workers=1)
sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
bp = sb.alloc(S * N * 64); bn = sb.alloc(S * 4); bd = sb.alloc(S * 8)
names_d = ["stage", "tree build (rest)", "centroids+radii", "queries", "labelling", "-", "  build: min/max", "  build: split dim+keys",
           "  build: rank scan", "  build: partition"]
for f in range(F):
    sb.stats_reset()
    sb.profile_reset()
    sb.profile(True)
    bp.upload(pts[f].astype(np.float64)); bn.upload(cnt[f]); bd.upload(dts[f])
    sb.step_dev(bp.ptr, bn.ptr, bd.ptr)
    sb.synchronize()
    sb.profile(False)
    line = f"frame {f}: " + "  ".join(f"{nm} {sb.profile_get(k)[0] * 1e3:.1f}us" for nm, k in (("predict", 5), ("track", 0), ("post", 6), ("big", 1)))
    out = np.zeros(32, dtype=np.uint64)
    sb._chk(sb.L.mmw_stats_get_ext(sb.h, out.ctypes.data))
    line += f" | dbscan calls {int(out[3])} mean U {float(out[4]) / max(float(out[3]), 1):.0f} clusters {int(out[7])}"
    print(line)
    if stamps:
        calls = float(out[3]) - float(out[31])
        totd = float(out[20:30].sum())
        if calls and totd:
            print("   " + "  ".join(f"{nm.strip()} {float(out[20 + i]) / calls:.0f}" for i, nm in enumerate(names_d) if nm != "-"))
