"""One scene, the model attached (TrackBuffer.attach_posture_model): wall clock of track_raw (= normalize + track + posture, one
round trip) per frame, and the same loop unattached.  python scripts/single_latency_attached.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_single as bs
from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
from mmwave_msc_amd.tracking import BatchedData, TrackBuffer

path = bs.write_experiment("/tmp/mmw_single_lat/A")
model = MarsCNN.from_keras_weights(random_keras_weights(0, 3)).to("cuda:0")
for attach in (False, True, False, True):
    tb, batch = TrackBuffer(max_pts=256), BatchedData()
    if attach:
        assert tb.attach_posture_model(model)

    def nt(det, dt):
        tb.dt = dt
        return tb.track_raw(det, batch)

    def posture():
        tb.estimate_posture(model)
        torch.cuda.synchronize()

    wall, lt, lp, frames = bs._loop(path, nt, posture)
    tot = (lt + lp)[10:] * 1e6
    print(f"attached={attach}: track_raw median {np.median(lt[10:]) * 1e6:7.1f} us  posture {np.median(lp[10:]) * 1e6:7.1f} us  "
          f"sum median {np.median(tot):7.1f}  p95 {np.percentile(tot, 95):7.1f}  frames {frames}  cap {tb._sb.track_cap}")
    tb.close()
