#!/usr/bin/env python3
"""Diagnostic: raw clock probes of ONE k_track workgroup (STAMPS build), printed as a timeline per wave."""
import os, sys
import numpy as np
os.environ["MMW_LIB_NAME"] = "libmmw_hip_stamps.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import bench  # noqa: E402
from mmwave_msc_amd import _lib  # noqa: E402
from mmwave_msc_amd.batch import SceneBatch  # noqa: E402

S, N, T, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 512, 8, 30
POP = sys.argv[2] if len(sys.argv) > 2 else "full"
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=16, population=POP)
sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
bp = sb.alloc(S * N * 64); bn = sb.alloc(S * 4); bd = sb.alloc(S * 8)
for f in range(F):
    bp.upload(pts[f].astype(np.float64)); bn.upload(cnt[f]); bd.upload(dts[f])
    sb.step_dev(bp.ptr, bn.ptr, bd.ptr)
sb.synchronize()
out = np.zeros(256 + 8192, dtype=np.uint64)
fn = sb.L.mmw_diag_probes
fn.argtypes = [C.c_void_p, C.c_void_p]
fn(sb.h, out.ctypes.data)
o = out[:256].reshape(4, 64).astype(np.int64)
t0 = o[o > 0].min()
for w in range(4):
    print("T,counts:", None) if False else None
    print("wave", w, {i: int(o[w, i] - t0) for i in range(64) if o[w, i] > 0})
