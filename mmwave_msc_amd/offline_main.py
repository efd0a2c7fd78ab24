"""Headless counterpart of the reference's offline loop (src/offline_main.py:21-65):

    read frame -> normalize_data -> TrackBuffer.track -> TrackBuffer.estimate_posture

The reference version owns a Qt application and a visualiser and runs at import; this
one is a function, takes the experiment path as an argument and reports through a
callback instead of `visual.update` (Visualizer.py is out of scope).
"""
from __future__ import annotations

import os
from typing import Callable, Optional

from . import constants as const
from .tracking import BatchedData, TrackBuffer
from .utils import OfflineManager, normalize_data

SLEEPTIME = 0.1  # radar frame period, config_cases/our_config_8.5m.cfg "frameCfg" (offline_main.py:26)


def offline_main(experiment_path: str, model=None, on_frame: Optional[Callable] = None,
                 max_frames: Optional[int] = None, max_pts: int = 512, device: int = 0) -> TrackBuffer:
    """Replays a logged experiment.  `on_frame(trackbuffer, detObj, frame_no)` replaces
    `visual.update(trackbuffer, detObj)`.  `model`: anything with `.predict`, or the path of a Keras `.h5` /
    `.npz` weight file (offline_main.py:33 `load_model(const.P_MODEL_PATH)`).  Returns the TrackBuffer."""
    if not os.path.exists(experiment_path):
        raise ValueError(f"No experiment file found in the path: {experiment_path}")
    if isinstance(model, (str, os.PathLike)):
        from .mars import MarsCNN
        model = MarsCNN.load(os.fspath(model)).to(f"cuda:{device}")
    sensor_data = OfflineManager(experiment_path)
    trackbuffer = TrackBuffer(max_pts=max_pts, device=device)
    if model is not None and hasattr(model, "has_small_path"):
        trackbuffer.attach_posture_model(model)   # this loop estimates after every tracked frame: one round trip per frame
    batch = BatchedData()
    first_iter = True
    seen = 0
    while not sensor_data.is_finished():
        data_ok, frame_no, det = sensor_data.get_data()
        if not data_ok:
            continue
        if first_iter:
            trackbuffer.dt = SLEEPTIME
            first_iter = False
        else:
            trackbuffer.dt = det["posix"][0] / 1000 - trackbuffer.t
        trackbuffer.t = det["posix"][0] / 1000
        # (the reference: effective_data = normalize_data(det); if effective_data.shape[0] != 0: trackbuffer.track(effective_data,
        #  batch) -- offline_main.py:53-57.  Both calls exist here too (utils.normalize_data, TrackBuffer.track); the loop uses
        #  their fused form, one round trip to the GPU per frame instead of two)
        if trackbuffer.track_raw(det, batch) != 0:
            if model is not None:
                trackbuffer.estimate_posture(model)
        if on_frame is not None:
            on_frame(trackbuffer, det, frame_no)
        seen += 1
        if max_frames is not None and seen >= max_frames:
            break
    return trackbuffer


if __name__ == "__main__":
    import sys

    tb = offline_main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(const.P_LOG_PATH, "mmWave", "A21"),
                      model=const.P_MODEL_PATH if os.path.exists(const.P_MODEL_PATH) else None)
    print("tracks at end:", len(tb.effective_tracks))
