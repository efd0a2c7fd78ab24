"""bench.py's `single_scene` leg: BASELINE.json configs[0] -- ONE logged scene through the offline loop (offline_main.py:21-65:
CSV shards -> normalize_data -> TrackBuffer.track -> TrackBuffer.estimate_posture), ~200 points per frame, 2 targets -- on the
GPU drop-in (`mmwave_msc_amd.offline_main`, `tracking.TrackBuffer`) and, beside it on this host's cores, on the
reference-shaped Python port (oracle/py_tracker.py + oracle/mars_torch.py).  The reference runs this loop at 10 frames/s in
real time (offline_main.py:26); what is measured here is how long a frame takes.

The experiment is synthetic (the reference ships no logs): two walking targets + clutter (mmwave_msc_amd/synth.py), turned into
raw radar rows whose doppler is the radial velocity (bench_ingest.raw_rows_from_normalised), written as the two CSV shards
DataLogging.py:60-82 would have written.  The oracle is used as the CPU baseline and as the checker of the GPU loop's final
state."""
import os
import tempfile
import time

import numpy as np

N_FRAMES, N_PTS, N_TARGETS, SEED = 120, 200, 2, 4242


def write_experiment(root, n_frames=N_FRAMES, n_pts=N_PTS, n_targets=N_TARGETS, seed=SEED):
    """Two CSV shards `<root>/1.csv`, `<root>/2.csv`: frame, x, y, z, doppler, peakVal, posix_ms -- one row per detected point."""
    from bench_ingest import raw_rows_from_normalised
    from mmwave_msc_amd import constants as const
    from mmwave_msc_amd.synth import make_scene
    pts, cnt, dts = make_scene(seed, n_frames, n_pts, n_targets, ragged=True)
    ang = np.radians(const.S_TILT)
    raw = raw_rows_from_normalised(pts, float(np.cos(ang)), float(np.sin(ang)), float(const.S_HEIGHT))
    os.makedirs(root, exist_ok=True)
    t_ms, half = 1_700_000_000_000, n_frames // 2
    for shard, (f0, f1) in enumerate(((0, half), (half, n_frames)), start=1):
        with open(os.path.join(root, f"{shard}.csv"), "w") as fh:
            for f in range(f0, f1):
                stamp = t_ms + int(round(1000 * float(np.sum(dts[: f + 1]))))
                for i in range(int(cnt[f])):
                    x, y, z, dop, peak = (float(v) for v in raw[f, i])
                    fh.write(f"{f + 1},{x!r},{y!r},{z!r},{dop!r},{peak!r},{stamp}\n")
    return root


def _loop(path, normalize_track, posture, after_frame=None):
    """the reference's loop (offline_main.py:36-62) over the package's CSV iterator; returns per-frame seconds of the two calls"""
    from mmwave_msc_amd.utils import OfflineManager
    src = OfflineManager(path)
    lat_t, lat_p, first, t_prev, frames = [], [], True, 0.0, 0
    t_loop = time.perf_counter()
    while not src.is_finished():
        ok, _, det = src.get_data()
        if not ok:
            continue
        dt = 0.1 if first else det["posix"][0] / 1000 - t_prev
        first, t_prev = False, det["posix"][0] / 1000
        t0 = time.perf_counter()
        kept = normalize_track(det, dt)
        t1 = time.perf_counter()
        if kept:
            posture()
        t2 = time.perf_counter()
        lat_t.append(t1 - t0)
        lat_p.append(t2 - t1)
        frames += 1
        if after_frame is not None:
            after_frame()
    return time.perf_counter() - t_loop, np.array(lat_t), np.array(lat_p), frames


def cpu_single_scene(path, weights):
    """oracle/py_tracker.py (reference-shaped numpy + sklearn) + oracle/mars_torch.py (fp32 CNN on torch's CPU operators), one core."""
    import torch
    from oracle.mars_torch import MarsTorchCPU
    from oracle.py_tracker import Params, PyScene, py_estimate_posture, py_normalize
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    p = Params(TR_MAX_TRACKS=4)
    sc = PyScene(p)
    model = MarsTorchCPU(weights, torch.float32, threads=1)

    def nt(det, dt):
        rows = py_normalize(p, det)
        if len(rows):
            sc.track(rows, dt)
        return len(rows)

    wall, lt, lp, frames = _loop(path, nt, lambda: py_estimate_posture(p, sc, model))
    skip = min(10, frames // 4)
    return {"value": round(frames / wall, 2), "unit": "frames/s", "cores": 1, "kind": "port",
            "track_ms_per_frame_median": round(float(np.median(lt[skip:])) * 1e3, 3),
            "posture_ms_per_frame_median": round(float(np.median(lp[skip:])) * 1e3, 3),
            "sample": f"oracle/py_tracker.py (py_normalize + PyScene.track + py_estimate_posture, the reference's computational shape) with "
                      f"oracle/mars_torch.py (fp32, 1 thread) on the same {frames}-frame CSV experiment, {wall:.1f} s wall"}, sc.n_tracks


def gpu_single_scene(path, weights, device=0):
    """`mmwave_msc_amd.offline_main` (whole loop, frames/s) and the same loop with the two calls timed (latency per frame)."""
    import torch
    from mmwave_msc_amd.mars import MarsCNN
    from mmwave_msc_amd.offline_main import offline_main
    from mmwave_msc_amd.tracking import BatchedData, TrackBuffer
    dev = torch.device("cuda", device)
    model = MarsCNN.from_keras_weights(weights).to(dev)
    seen = []
    tb = offline_main(path, model=model, on_frame=lambda *_: seen.append(time.perf_counter()), max_pts=256, device=device)   # warms everything
    tb.close()
    seen.clear()
    t0 = time.perf_counter()
    tb = offline_main(path, model=model, on_frame=lambda *_: seen.append(time.perf_counter()), max_pts=256, device=device)
    torch.cuda.synchronize()
    loop_s = time.perf_counter() - t0
    frames = len(seen)
    nt_final = len(tb.effective_tracks)
    finals = tb._sb.tracks(cap=max(nt_final, 1))[0, :nt_final].copy()
    tb.close()
    # the two calls timed (the posture one until its keypoints are in the track records: stream synchronised)
    tb2, batch = TrackBuffer(max_pts=256, device=device), BatchedData()
    attached = tb2.attach_posture_model(model)   # as offline_main does: estimate_posture rides in track_raw's round trip

    def nt(det, dt):
        tb2.dt = dt
        return tb2.track_raw(det, batch)

    def posture():
        tb2.estimate_posture(model)
        torch.cuda.synchronize()

    _, lt, lp, _ = _loop(path, nt, posture)
    tb2.close()
    skip = min(10, frames // 4)
    res = {"value": round(frames / loop_s, 1), "unit": "frames/s", "frames": frames,
           "loop": "mmwave_msc_amd.offline_main: OfflineManager (CSV) -> TrackBuffer.track_raw (normalize_data + track + the attached "
                   "model's estimate_posture: features, MARS CNN in fp32 and keypoint scatter on the device, ONE round trip) -> "
                   "estimate_posture (finds its work done)",
           "posture_in_track_round_trip": bool(attached),
           "loop_us_per_frame": round(loop_s / frames * 1e6, 1),
           "track_us_per_frame_median": round(float(np.median(lt[skip:])) * 1e6, 1),
           "posture_us_per_frame_median": round(float(np.median(lp[skip:])) * 1e6, 1),
           "track_plus_posture_us_median": round(float(np.median((lt + lp)[skip:])) * 1e6, 1),
           "track_us_p95": round(float(np.percentile(lt[skip:], 95)) * 1e6, 1)}
    return res, finals


def oracle_final_state(path):
    """the same loop on oracle/c (normalize + track): the final track state the GPU loop must equal bit for bit"""
    from oracle import c_oracle as co
    cfg = co.default_config()
    sc = co.OracleScene(cfg, 256)

    def nt(det, dt):
        raw = np.vstack((det["x"], det["y"], det["z"], det["doppler"], det["peakVal"])).T.astype(np.float64)
        rows = co.normalize(cfg, raw)
        if len(rows):
            sc.track(rows, dt)
        return len(rows)

    _loop(path, nt, lambda: None)
    return sc.tracks()


def single_scene_cpu(workdir=None):
    """CPU side (before the GPU is touched): writes the experiment, runs the port and the checker.  Returns a dict for gpu side."""
    from mmwave_msc_amd.mars import random_keras_weights
    root = os.path.join(workdir or tempfile.mkdtemp(prefix="mmw_single_"), "A_synth")
    write_experiment(root)
    weights = random_keras_weights(0, 3)
    cpu, ntr = cpu_single_scene(root, weights)
    return {"path": root, "weights": weights, "cpu": cpu, "cpu_tracks": ntr, "want": oracle_final_state(root)}


def single_scene_gpu(prep, device=0):
    res, finals = gpu_single_scene(prep["path"], prep["weights"], device)
    want = prep["want"]
    ok = len(want) == len(finals)
    if ok:
        for name in ("x", "P", "centroid", "spread_est", "group_disp_est", "lifetime", "point_num", "is_static", "ring_n"):
            ok = ok and bool(np.array_equal(finals[name], want[name]))
    res["config"] = (f"1 scene, {N_FRAMES} frames in 2 CSV shards, <= {N_PTS} points per frame, {N_TARGETS} targets, TR_MAX_TRACKS=4 "
                     f"(constants.py defaults); BASELINE.json configs[0]")
    res["parity"] = {"tracks": int(len(want)), "final_state_bit_equal_vs_oracle": bool(ok)}
    res["cpu_baseline"] = prep["cpu"]
    res["speedup_vs_cpu_baseline"] = round(res["value"] / prep["cpu"]["value"], 1)
    res["realtime_requirement"] = "10 frames/s (one IWR1443 frame every 100 ms, offline_main.py:26)"
    return res
