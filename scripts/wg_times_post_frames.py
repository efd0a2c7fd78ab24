#!/usr/bin/env python3
"""Diagnostic (make STAMPS=1 DIAGFLAGS=-DMMW_STAMPS_POST): per FRAME, when k_post's update blocks were done and when block 0 (which
holds the launch until every cloud the side stream claimed is finished) left -- what a context too small to hide a BallTree chain
under its Kalman update waits for.  usage: wg_times_post_frames.py [scenes] [full|mixed] [frames]"""
import os, sys
import numpy as np
os.environ["MMW_LIB_NAME"] = "libmmw_hip_stamps.so"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import bench  # noqa: E402
from mmwave_msc_amd import _lib  # noqa: E402
from mmwave_msc_amd.batch import SceneBatch  # noqa: E402

S, N, T = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 512, 8
POP = sys.argv[2] if len(sys.argv) > 2 else "full"
F = int(sys.argv[3]) if len(sys.argv) > 3 else 60
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=16, population=POP)
sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N)
bp = sb.alloc(S * N * 64); bn = sb.alloc(S * 4); bd = sb.alloc(S * 8)
fn = sb.L.mmw_diag_probes
fn.argtypes = [C.c_void_p, C.c_void_p]
dbn = sb.alloc(S * 4)
rows = []
G0 = min(S, 256)
for f in range(F):
    bp.upload(pts[f].astype(np.float64)); bn.upload(cnt[f]); bd.upload(dts[f])
    sb.step_dev(bp.ptr, bn.ptr, bd.ptr, dbn_ptr=dbn.ptr)
    sb.synchronize()
    out = np.zeros(256 + 8192, dtype=np.uint64)
    fn(sb.h, out.ctypes.data)
    w = out[256:256 + 4 * 2048].reshape(-1, 4).astype(np.int64)
    ok = w[:, 0] > 0
    t0 = w[ok, 0].min()
    end = (w[:, 2] - t0) / 100.0
    up = (np.arange(2048) > G0) & ok
    n_db = dbn.download((S,), np.int32)
    rows.append((f, end[up].max() if up.any() else 0.0, end[0], int((n_db >= 0).sum()), int((n_db > 0).sum())))
print("frame: update blocks done (us) | block 0 leaves (us) | scenes whose apply_DBscan ran | ... that found clusters")
for r in rows[8:]:
    print(f"  {r[0]:3d}: {r[1]:6.1f} | {r[2]:6.1f} | {r[3]:4d} | {r[4]:3d}")
a = np.array([r[2] for r in rows[10:]]); u = np.array([r[1] for r in rows[10:]])
print(f"frames 10..: block 0 leaves at median {np.median(a):.1f} us, mean {a.mean():.1f}; update done median {np.median(u):.1f}; frames where block 0 is last by > 5 us: {(a > u + 5).sum()} of {len(a)}")
