#!/usr/bin/env python3
"""Diagnostic: apply_DBscan on the golden clouds, LDS classes (<= 1920 points) against the global-memory path above them."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
zs, zh = np.load(os.path.join(G, "dbscan.npz")), np.load(os.path.join(G, "dbscan_huge.npz"))
sb = SceneBatch(_lib.default_config(fb_frames_batch=3), 1, 1024)
for z, sizes in ((zs, [481, 961, 1536]), (zh, [int(v) for v in zh["sizes"]])):
    for n in sizes:
        pts = np.zeros((1, n, 8)); pts[0] = z[f"pts_{n}"]
        nn = np.array([n], np.int32)
        b_p = sb.buf("db_pts", pts.nbytes).upload(pts); b_n = sb.buf("db_n", 4).upload(nn)
        b_l = sb.buf("db_lab", n * 4); b_c = sb.buf("db_ncl", 4)
        for rep in range(3):
            sb.synchronize(); t0 = time.perf_counter()
            sb._chk(sb.L.mmw_dbscan(sb.h, b_p.ptr, b_n.ptr, n, sb.cfg.db_eps, 35, b_l.ptr, b_c.ptr))
            sb.synchronize(); dt = time.perf_counter() - t0
        print(f"n = {n:5d}: {dt * 1e6:8.0f} us  ({'global-memory slab' if n > 1920 else 'LDS'}), clusters {int(b_c.download((1,), np.int32)[0])}")
