"""mmwave_msc_amd -- MI355X (gfx950) implementation of the per-frame point-cloud
hot path of AsteriosPar/mmWave_MSc: DBSCAN clustering, Mahalanobis gating /
association, per-track Kalman predict/update (src/Tracking.py) and the MARS
feature map + 19-keypoint CNN (src/Utils.py, src/train.py), behind the
reference's own call surface (TrackBuffer / BatchedData / constants /
offline_main frame iterator).

The compute path is hand-written HIP behind a C-ABI (include/mmw.h,
mmwave_msc_amd/libmmw_hip.so).  There is no CPU fallback.
"""
__version__ = "0.1.0"
