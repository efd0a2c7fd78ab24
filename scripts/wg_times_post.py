#!/usr/bin/env python3
"""Diagnostic (make STAMPS=1 DIAGFLAGS=-DMMW_STAMPS_POST): start / end of the workgroups of the LAST frame's k_post -- the 256 worker
blocks (block 0 waits for the claimed clouds), block 256 (next frame's schedule) and the update blocks behind them."""
import os, sys
import numpy as np
os.environ["MMW_LIB_NAME"] = "libmmw_hip_stamps.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import bench  # noqa: E402
from mmwave_msc_amd import _lib  # noqa: E402
from mmwave_msc_amd.batch import SceneBatch  # noqa: E402

S, N, T, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 4096), 512, 8, 30
POP = sys.argv[2] if len(sys.argv) > 2 else "full"   # "full": K = T targets in every scene (bench.py's headline); "mixed": 1 + s mod T
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=16, population=POP)
sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
bp = sb.alloc(S * N * 64); bn = sb.alloc(S * 4); bd = sb.alloc(S * 8)
for f in range(F):
    bp.upload(pts[f].astype(np.float64)); bn.upload(cnt[f]); bd.upload(dts[f])
    sb.step_dev(bp.ptr, bn.ptr, bd.ptr)
sb.synchronize()
print("tracks:", int(sb.num_tracks().sum()), "side workers:", sb.side_workers(), "kalman layout:", sb.kalman_layout())
out = np.zeros(256 + 8192, dtype=np.uint64)
fn = sb.L.mmw_diag_probes
fn.argtypes = [C.c_void_p, C.c_void_p]
fn(sb.h, out.ctypes.data)
w = out[256:256 + 4 * 2048].reshape(-1, 4).astype(np.int64)
ok = w[:, 0] > 0
t0 = w[ok, 0].min()
start, end = (w[:, 0] - t0) / 100.0, (w[:, 2] - t0) / 100.0   # us
G0 = min(S, 256)
wk, up = np.arange(2048) < G0, (np.arange(2048) > G0) & ok   # (block G0 = the schedule sort)
print(f"workers: start {start[wk].min():.2f}..{start[wk].max():.2f}  end median {np.median(end[wk]):.2f}  max {end[wk].max():.2f} (block {int(np.argmax(np.where(wk, end, -1)))})")
print(f"  block 0 (waits for the claimed clouds): {start[0]:.2f} -> {end[0]:.2f};  block {G0} (schedule sort): {start[G0]:.2f} -> {end[G0]:.2f}")
live = up & ((end - start) > 1.0)
print(f"update blocks: {int(up.sum())} stamped, {int(live.sum())} with work; start {start[up].min():.2f}..{start[up].max():.2f}; "
      f"of those with work: duration median {np.median((end - start)[live]):.2f} max {(end - start)[live].max():.2f}, last end {end[live].max():.2f}")
hist, edges = np.histogram(start[live], bins=8)
print("  start histogram of the update blocks with work (us):", [f"{edges[i]:.1f}:{hist[i]}" for i in range(8)])
print(f"launch: first start 0 -> last end {end[ok].max():.2f} us")
