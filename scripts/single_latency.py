"""Where a single scene's estimate_posture / track_raw spend their time (host wall clock, stream synchronised between the
pieces): python scripts/single_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_single as bs
from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
from mmwave_msc_amd.tracking import BatchedData, TrackBuffer
from mmwave_msc_amd.utils import OfflineManager

path = bs.write_experiment("/tmp/mmw_single_lat/A")
model = MarsCNN.from_keras_weights(random_keras_weights(0, 3)).to("cuda:0")
tb, batch = TrackBuffer(max_pts=256), BatchedData()
src = OfflineManager(path)
T = {k: [] for k in ("track_raw", "features", "conv", "head", "scatter", "sync", "posture_call")}
first, t_prev = True, 0.0
while not src.is_finished():
    ok, _, det = src.get_data()
    if not ok:
        continue
    tb.dt = 0.1 if first else det["posix"][0] / 1000 - t_prev
    first, t_prev = False, det["posix"][0] / 1000
    t0 = time.perf_counter(); kept = tb.track_raw(det, batch); T["track_raw"].append(time.perf_counter() - t0)
    if not kept:
        continue
    t0 = time.perf_counter(); tb.estimate_posture(model); T["posture_call"].append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    P = tb._posture
    if P is None:
        continue
    sb = tb._sb
    with torch.cuda.stream(P["stream"]), torch.no_grad():
        t0 = time.perf_counter(); n = sb.features_dev(P["feat"].data_ptr(), P["owner"].data_ptr(), P["feat"].shape[0]); t1 = time.perf_counter()
        if n:
            act = model._hip_convs(P["feat"][:n]); torch.cuda.synchronize(); t2 = time.perf_counter()
            kp = model.forward_small(P["feat"][:n]); torch.cuda.synchronize(); t3 = time.perf_counter()
            sb.set_keypoints_dev(kp.data_ptr(), P["owner"].data_ptr(), n); t4 = time.perf_counter()
            torch.cuda.synchronize(); t5 = time.perf_counter()
            T["features"].append(t1 - t0); T["conv"].append(t2 - t1); T["head"].append(t3 - t2); T["scatter"].append(t4 - t3); T["sync"].append(t5 - t4)
for k, v in T.items():
    if v:
        print(f"{k:14s} median {np.median(v[10:]) * 1e6:8.1f} us   min {np.min(v[10:]) * 1e6:8.1f}   n {len(v)}")
