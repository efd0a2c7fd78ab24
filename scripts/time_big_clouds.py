import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch
S, N, F = 256, 512, 6
rng = np.random.default_rng(3)
pts = np.zeros((F, S, N, 8))
pts[..., 0] = rng.uniform(-6, 6, size=(F, S, N)); pts[..., 1] = rng.uniform(0.3, 7.5, size=(F, S, N)); pts[..., 2] = rng.uniform(0.05, 2.4, size=(F, S, N))
for s in range(S):
    for f in range(F):
        idx = rng.choice(N, size=14, replace=False)
        pts[f, s, idx, 0:2] = np.array([1.0, 3.0]) + rng.normal(0, 0.06, size=(14, 2))
        pts[f, s, idx, 2] = rng.uniform(0.6, 1.4, size=14)
cnt = np.full((F, S), N, np.int32); dts = np.full((F, S), 0.1)
for rep in range(2):
    sb = SceneBatch(_lib.default_config(tr_max_tracks=4, chain_side_stream=-1), S, N)
    bp = [sb.alloc(S * N * 64) for _ in range(3)]; bn = sb.alloc(S * 4); bd = sb.alloc(S * 8)
    for f in range(3): bp[f].upload(pts[f])
    bn.upload(cnt[0]); bd.upload(dts[0])
    for f in range(3):
        sb.synchronize(); t0 = time.perf_counter()
        sb.step_dev(bp[f].ptr, bn.ptr, bd.ptr)
        sb.synchronize(); dt = time.perf_counter() - t0
        if rep: print(f"frame {f}: {dt * 1e6:.0f} us")
    st = sb.stats(); print("calls", int(st[3]), "clusters", int(st[7]))
    sb.close()
