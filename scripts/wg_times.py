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

S, N, T, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 512), (int(sys.argv[2]) if len(sys.argv) > 2 else 512), 8, 30
CSS = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # mmw_config.chain_side_stream (-1: no DBSCAN workers beside k_track)
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=16)
sb = SceneBatch(_lib.default_config(tr_max_tracks=T, chain_side_stream=CSS), S, N)
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
sh = 0
while (S >> sh) > 2048:
    sh += 1
nst = S >> sh
w = out[256:256 + 4 * nst].reshape(-1, 4).astype(np.int64)
t0 = w[:, 0].min()
start, end = (w[:, 0] - t0) / 100.0, (w[:, 2] - t0) / 100.0   # us
dur = end - start
cyc = w[:, 3] - w[:, 1]
print(f"S={S}: every {1 << sh}th workgroup stamped; first start 0, last start {start.max():.2f} us, last end {end.max():.2f} us")
print("  duration us: median %.2f  p10 %.2f  p90 %.2f  max %.2f  mean %.2f ; cycles median %d max %d ; clock GHz median %.2f" % (
    np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), dur.max(), dur.mean(), np.median(cyc), cyc.max(), np.median(cyc / np.maximum(dur, 1e-3)) / 1e3))
span = end.max()
slots = 256 * (5 if N > 256 else 4)
print(f"  sum of durations x {1 << sh} = {dur.sum() * (1 << sh):.0f} us = {dur.sum() * (1 << sh) / slots:.1f} us on {slots} slots; launch {span:.1f} us -> slot utilisation {dur.sum() * (1 << sh) / slots / span:.2f}")
# resident workgroups over time (sampled ones, scaled)
edges = np.linspace(0, span, 25)
res = [(1 << sh) * int(((start <= t) & (end > t)).sum()) for t in edges[:-1]]
print("  resident workgroups at", [f"{t:.0f}" for t in edges[:-1]])
print("                        ", res)
# by dispatch index: eighths
for k in range(8):
    sl = slice(k * nst // 8, (k + 1) * nst // 8)
    print(f"  blocks {sl.start << sh:5d}..{sl.stop << sh:5d}: start {start[sl].min():7.2f}..{start[sl].max():7.2f}  duration mean {dur[sl].mean():6.2f}  end max {end[sl].max():7.2f}")
