#!/usr/bin/env python3
"""Diagnostic (STAMPS build): start / end of every k_track workgroup of the LAST frame, in 100 MHz ticks and shader cycles."""
import os, sys
import numpy as np
os.environ["MMW_LIB_NAME"] = "libmmw_hip_stamps.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import bench  # noqa: E402
from mmwave_msc_amd import _lib  # noqa: E402
from mmwave_msc_amd.batch import SceneBatch  # noqa: E402

S, N, T, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 512), 512, 8, 30
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=16)
sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
bp = sb.alloc(S * N * 64); bn = sb.alloc(S * 4); bd = sb.alloc(S * 8)
for f in range(F):
    bp.upload(pts[f].astype(np.float64)); bn.upload(cnt[f]); bd.upload(dts[f])
    sb.step_dev(bp.ptr, bn.ptr, bd.ptr)
sb.synchronize()
ntr = sb.num_tracks()
out = np.zeros(256 + 8192, dtype=np.uint64)
fn = sb.L.mmw_diag_probes
fn.argtypes = [C.c_void_p, C.c_void_p]
fn(sb.h, out.ctypes.data)
w = out[256:256 + 4 * min(S, 2048)].reshape(-1, 4).astype(np.int64)
t0 = w[:, 0].min()
start, end = (w[:, 0] - t0) / 100.0, (w[:, 2] - t0) / 100.0   # us
cyc = w[:, 3] - w[:, 1]
print(f"S={S}: first start 0, last start {start.max():.2f} us, last end {end.max():.2f} us")
print("  duration us: median %.2f  p90 %.2f  max %.2f ; cycles median %d max %d ; clock GHz median %.2f" % (
    np.median(end - start), np.percentile(end - start, 90), (end - start).max(), np.median(cyc), cyc.max(), np.median(cyc / np.maximum(end - start, 1e-3)) / 1e3))
order = np.argsort(-(end))[:8]
for b in order:
    print(f"  block {b}: start {start[b]:.2f} end {end[b]:.2f} dur {end[b]-start[b]:.2f} us cycles {cyc[b]}")
