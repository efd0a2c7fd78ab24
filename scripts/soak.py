#!/usr/bin/env python3
"""Soak: many frames of a side-worker context (SOAK_S >= 1536 scenes, or chain_side_stream = 1), then mmw_check (a worker that gave up a wait it must not
give up is a sticky error) and the final state against the C oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch
from oracle import c_oracle as co

S, N, T, F = int(os.environ.get("SOAK_S", "2048")), 512, 8, int(os.environ.get("SOAK_F", "400"))
pts, cnt, dts = bench.generate(np.arange(S), F, N, T, workers=16)
ob = co.OracleBatch(co.default_config(tr_max_tracks=T), S, N)
t0 = time.perf_counter()
co.batch_run_f32(ob, pts, cnt, dts, 0)
print(f"oracle: {time.perf_counter() - t0:.1f} s", flush=True)
dev = torch.device("cuda:0")
sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N, device=0)
d_cnt = torch.from_numpy(cnt).to(dev); d_dt = torch.from_numpy(dts).to(dev)
a = torch.empty((S, N), dtype=torch.int32, device=dev); lab = torch.empty((S, sb.UM), dtype=torch.int32, device=dev)
dbn = torch.empty((S,), dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st); sb.follow_torch_stream(st)
t0 = time.perf_counter()
for f in range(F):
    p = torch.from_numpy(pts[f]).to(dev).double()
    sb.step_dev(p.data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr(), a.data_ptr(), lab.data_ptr(), dbn.data_ptr())
torch.cuda.synchronize()
print(f"gpu: {F} frames in {time.perf_counter() - t0:.2f} s, side workers {sb.side_workers()}", flush=True)
sb.check()
ntr = sb.num_tracks(); trk = sb.tracks(cap=max(int(ntr.max()), 1))
bad = 0
for s in range(S):
    want = ob.scenes[s].tracks()
    ok = ntr[s] == len(want) and all(np.array_equal(trk[s, : ntr[s]][n], want[n]) for n in ("x", "P", "centroid", "lifetime", "point_num", "ring_n"))
    bad += 0 if ok else 1
print("scenes differing from the oracle:", bad)
sys.exit(1 if bad else 0)
