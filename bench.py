#!/usr/bin/env python3
"""Benchmark of the hot path: scene-frames/s of `TrackBuffer.track` (DBSCAN + gating /
association + Kalman) over S concurrent synthetic scenes per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one radar frame for every scene of the rank (one `mmw_step`).  Inputs for
all W+K frames are resident in HBM before the timed region.  One JSON line on rank 0.

Workload (config.workload): BASELINE.json configs[2] at one GPU -- 4096 scenes x 512
points, TR_MAX_TRACKS = 8 -- with scene s holding 1 + (s mod 8) walking targets, so that
7/8 of the scenes keep calling apply_DBscan every frame (a scene whose track list is
full never clusters again; see DESIGN.md §5 for why the population is mixed).
Weak scaling: every rank owns `--scenes` scenes; no data-path collective; one RCCL
all-gather of the track table closes the timed region when N > 1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mmwave_msc_amd.synth import make_scene  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def _gen_one(args):
    sid, frames, n_pts, n_targets = args
    return make_scene(sid, frames, n_pts, n_targets)


def generate(scene_ids, frames, n_pts, tracks, workers):
    """points[F,S,N,8] float32, counts[F,S] int32, dt[F,S] float64 for the given global scene ids."""
    jobs = [(int(s), frames, n_pts, 1 + int(s) % tracks) for s in scene_ids]
    if workers > 1 and len(jobs) > 8:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(_gen_one, jobs, chunksize=max(1, len(jobs) // (workers * 4)))
    else:
        res = [_gen_one(j) for j in jobs]
    pts = np.stack([r[0] for r in res], axis=1)
    cnt = np.stack([r[1] for r in res], axis=1)
    dts = np.stack([r[2] for r in res], axis=1)
    return np.ascontiguousarray(pts), np.ascontiguousarray(cnt), np.ascontiguousarray(dts)


def cpu_legs(pts, cnt, dts, tracks, cores, py_scenes, py_frames, c_scenes):
    """CPU baselines on a bounded sample of the SAME workload, before the GPU is touched.
    Returns (cpu_baseline dict, native dict, oracle final states for the parity check)."""
    from oracle import c_oracle as co
    from oracle.py_tracker import run_batch_multiprocess

    F, S = pts.shape[0], pts.shape[1]
    out = {}
    # (i) reference-faithful Python restatement, scenes sharded over processes
    procs = max(1, min(cores, py_scenes))
    ns = min(S, py_scenes)
    nf = min(F, py_frames)
    el, _ = run_batch_multiprocess({"TR_MAX_TRACKS": tracks}, pts[:nf, :ns], cnt[:nf, :ns], dts[:nf, :ns], procs)
    out["cpu_baseline"] = {
        "value": round(ns * nf / el, 2), "unit": "scene-frames/s", "cores": procs, "kind": "port",
        "sample": f"oracle/py_tracker.py (numpy + per-point inv/det + sklearn DBSCAN with the Python metric, as the "
                  f"reference): first {ns} scenes x first {nf} frames of this workload, {procs} processes, {el:.1f} s wall",
    }
    # (ii) plain-C oracle, OpenMP over scenes, all frames (also yields the parity reference)
    cfg = co.default_config(tr_max_tracks=tracks)
    nc = min(S, c_scenes)
    ob = co.OracleBatch(cfg, nc, pts.shape[2])
    threads = max(1, min(cores, co.max_threads(), nc))
    sub_pts = np.ascontiguousarray(pts[:, :nc])
    sub_cnt, sub_dt = np.ascontiguousarray(cnt[:, :nc]), np.ascontiguousarray(dts[:, :nc])
    t0 = time.perf_counter()
    co.batch_run_f32(ob, sub_pts, sub_cnt, sub_dt, threads)
    elc = time.perf_counter() - t0
    out["cpu_baseline_native"] = {
        "value": round(nc * F / elc, 1), "unit": "scene-frames/s", "cores": threads, "kind": "port",
        "sample": f"oracle/c (plain C, OpenMP, each thread runs whole scenes): first {nc} scenes x all {F} frames, {threads} threads, {elc:.2f} s wall",
    }
    finals = [sc.tracks() for sc in ob.scenes]
    return out, finals


PROF_EVERY = 4  # hipEvent-timed steps inside the timed region: one in PROF_EVERY
OTHER_KERNELS = (5, 1, 6)  # _lib.K_PREDICT, K_DBSCAN, K_POST
# algorithmic bytes per track of the two Kalman kernels: k_predict reads the 1232-byte record
# prefix, writes P and x (720) and the gate record (352); the update half of k_post reads the prefix and
# writes P and x
PREDICT_BYTES_PER_TRACK = 1232 + 720 + 352
UPDATE_BYTES_PER_TRACK = 1232 + 720


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scenes", type=int, default=4096, help="scenes per GPU")
    ap.add_argument("--pts", type=int, default=512)
    ap.add_argument("--tracks", type=int, default=8, help="TR_MAX_TRACKS")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--py-scenes", type=int, default=64)
    ap.add_argument("--py-frames", type=int, default=20)
    ap.add_argument("--c-scenes", type=int, default=1024)
    ap.add_argument("--no-posture", action="store_true", help="skip the (untimed-for-value) feature-map + CNN leg")
    ap.add_argument("--gen-workers", type=int, default=-1,
                    help="processes for scene generation (-1 = auto; use 1 under rocprofv3: its preloaded tool initialises "
                         "the GPU before main(), and forking after that hangs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    S, N, K, W = args.scenes, args.pts, args.steps, args.warmup
    F = K + W
    cores = os.cpu_count() or 1

    # ---- host-side generation and CPU legs: nothing below touches the GPU yet ----
    t_gen = time.perf_counter()
    ids = np.arange(rank * S, rank * S + S)
    workers = args.gen_workers if args.gen_workers > 0 else max(1, min(32, cores // max(world, 1)))
    pts, cnt, dts = generate(ids, F, N, args.tracks, workers=workers)
    t_gen = time.perf_counter() - t_gen
    cpu, finals = {}, None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu, finals = cpu_legs(pts, cnt, dts, args.tracks, cores, args.py_scenes, args.py_frames, args.c_scenes)

    # ---- GPU ----
    import torch
    import torch.distributed as dist

    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.dist import all_gather_tables, SUMMARY_WORDS

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)
    sb = SceneBatch(_lib.default_config(tr_max_tracks=args.tracks), S, N, device=local_rank)
    # one real stream for torch and the context: uploads, the CNN of the posture leg and the mmw_* calls on device
    # tensors are then ordered by the stream itself (torch's default stream would read as "context's own stream")
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    sb.follow_torch_stream(side)
    d_pts = torch.empty((F, S, N, 8), dtype=torch.float64, device=dev)
    for f in range(F):
        d_pts[f] = torch.from_numpy(pts[f]).to(dev).double()
    d_cnt = torch.from_numpy(cnt).to(dev)
    d_dt = torch.from_numpy(dts).to(dev)
    d_assoc = torch.empty((S, N), dtype=torch.int32, device=dev)
    d_lab = torch.empty((S, sb.UM), dtype=torch.int32, device=dev)
    d_dbn = torch.empty((S,), dtype=torch.int32, device=dev)
    slots = args.tracks
    d_table = torch.zeros((S * slots, SUMMARY_WORDS), dtype=torch.int32, device=dev)

    def step(f):
        sb.step_dev(d_pts[f].data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr(),
                    d_assoc.data_ptr(), d_lab.data_ptr(), d_dbn.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()

    for f in range(W):
        step(f)
    torch.cuda.synchronize()
    sb.check()
    sb.stats_reset()
    sb.profile_reset()
    sb.profile(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(W, F):
        # A HIP-event pair costs the stream ~10 us of idle time, so the live kernel durations come from a
        # sample of the timed launches: every PROF_EVERY-th step times k_track and, in rotation, one of the
        # other three kernels of the step
        if (f - W) % PROF_EVERY == 0:
            sb.profile(True, kernels=(_lib.K_TRACK, OTHER_KERNELS[((f - W) // PROF_EVERY) % len(OTHER_KERNELS)]))
        else:
            sb.profile(False)
        step(f)
    sb.profile(True)
    sb.track_table_dev(d_table.data_ptr(), slots, scene_base=rank * S)
    gathered = all_gather_tables(d_table)
    torch.cuda.synchronize()
    barrier()
    el = time.perf_counter() - t0
    sb.profile(False)
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    sb.check()
    stats = sb.stats()
    prof = {k: sb.profile_get(k) for k in (_lib.K_TRACK, _lib.K_DBSCAN, _lib.K_TABLE, _lib.K_PREDICT, _lib.K_POST)}

    # ---- secondary: the posture leg on the final state (features kernel -> MARS CNN -> keypoints).
    #      Not part of `value`; reported so configs[3]/[4] (end-to-end) have a measured number. ----
    posture = None
    if not args.no_posture:
        try:
            from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
            cap = S * min(sb.track_cap, 2 * args.tracks)
            d_feat = torch.empty((cap, sb.ring, 8, 8, 5), dtype=torch.float32, device=dev)
            d_owner = torch.empty((cap, 2), dtype=torch.int32, device=dev)
            model = MarsCNN.from_keras_weights(random_keras_weights(0, sb.ring)).to(dev)

            def posture_iter():
                nrow = sb.features_dev(d_feat.data_ptr(), d_owner.data_ptr(), cap)
                with torch.no_grad():
                    kp = model(d_feat[:nrow])
                sb.set_keypoints_dev(kp.data_ptr(), d_owner.data_ptr(), nrow)
                return nrow

            for _ in range(2):
                nrow = posture_iter()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            reps = 5
            for _ in range(reps):
                nrow = posture_iter()
            torch.cuda.synchronize()
            tp = (time.perf_counter() - tp) / reps
            flop = 25187328.0 if sb.ring == 3 else 2837504.0
            posture = {"tracks": int(nrow), "ms_per_iter": round(tp * 1e3, 3), "samples_per_s": round(nrow / tp, 1),
                       "cnn_tflops_fp32": round(nrow * flop / tp / 1e12, 2), "mfma_fp32_peak_tflops": 157.3,
                       "note": "features kernel + torch-ROCm CNN (fp32, random Keras-layout weights) + keypoint scatter on the final state; "
                               "with this leg every frame the step would take ms_per_step + ms_per_iter"}
        except Exception as exc:  # never lose the headline line over the secondary leg
            posture = {"error": repr(exc)[:200]}

    parity = None
    if finals is not None:
        ntr = sb.num_tracks()
        trk = sb.tracks(cap=max(int(ntr.max()), 1))
        ok = True
        for s, want in enumerate(finals):
            got = trk[s, : ntr[s]]
            ok &= len(want) == ntr[s]
            if not ok:
                break
            for name in ("x", "P", "centroid", "spread_est", "group_disp_est", "lifetime", "point_num", "is_static", "ring_n"):
                ok &= bool(np.array_equal(got[name], want[name]))
        parity = {"scenes_checked": len(finals), "frames": F, "bit_equal_vs_oracle": bool(ok)}

    if rank == 0:
        total_sf = S * world * K
        # per-kernel device time from the sampled HIP-event pairs (one step in PROF_EVERY); algorithmic bytes
        # from the device counters, which cover all K steps
        n_samp = max(prof[_lib.K_TRACK][1], 1)
        step_ms = {k: prof[k][0] / max(prof[k][1], 1)
                   for k in (_lib.K_PREDICT, _lib.K_TRACK, _lib.K_DBSCAN, _lib.K_POST)}
        tracks_in = float(stats[5])  # sum over scene-frames of the tracks entering track()
        step_bytes = {
            _lib.K_PREDICT: tracks_in * PREDICT_BYTES_PER_TRACK / K,
            _lib.K_TRACK: float(stats[0]) / K,
            _lib.K_DBSCAN: float(stats[1]) / K,
            _lib.K_POST: tracks_in * UPDATE_BYTES_PER_TRACK / K,
        }
        dom = max(step_ms, key=step_ms.get)
        dom_ms, dom_bytes = step_ms[dom], step_bytes[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(tpath):
            try:
                tj = json.load(open(tpath))
                key = f"{S}x{N}x{args.tracks}"
                ent = tj.get(key, {}).get(_lib.load().mmw_kernel_name(dom).decode())
                # FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, bytes per launch, from a
                # separate rocprofv3 --pmc run of this workload (profiles/README.md)
                traffic = ent.get("hbm_bytes_per_launch_fetch_x2") if isinstance(ent, dict) else ent
            except Exception:
                traffic = None
        line = {
            "metric": "scene_frames_per_sec", "value": round(total_sf / el, 1), "unit": "scene-frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(el / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{S} scenes/GPU x {N} pts x TR_MAX_TRACKS={args.tracks}, DBSCAN+gating+KF (TrackBuffer.track), "
                            f"targets per scene = 1 + (scene_id mod {args.tracks}); BASELINE.json configs[2]",
                "scenes_per_gpu": S, "points_per_frame": N, "max_tracks": args.tracks, "frames_resident": F,
                "parallelism": f"scenes sharded over {world} GPU(s), weak; all-gather of track table once per run",
            },
            "roofline": {
                "kernel": _lib.load().mmw_kernel_name(dom).decode(), "bound": "hbm",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                "algorithmic_bytes_per_launch": round(dom_bytes, 1),
                "avg_launch_ms": round(dom_ms, 5), "launches_timed": n_samp,
            },
            # all four launches of a step together, per SURVEY.md §8(d): algorithmic bytes of the step / step time
            "roofline_whole_step": {"bound": "hbm", "achieved": round(sum(step_bytes.values()) / (el / K) / 1e9, 2), "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": round(sum(step_bytes.values()) / (el / K) / 1e9 / HBM_PEAK_GBS, 6),
                                    "algorithmic_bytes_per_step": round(sum(step_bytes.values()), 1)},
            "kernels": {
                name: {"avg_ms": round(step_ms[k], 5), "alg_bytes_per_launch": round(step_bytes[k], 1)}
                for k, name in ((_lib.K_PREDICT, "k_predict"), (_lib.K_TRACK, "k_track"),
                                (_lib.K_DBSCAN, "k_dbscan_big"), (_lib.K_POST, "k_post"))
            },
            "work": {"dbscan_calls_per_step": round(float(stats[3]) / K, 1), "mean_U": round(float(stats[4]) / max(float(stats[3]), 1), 1),
                     "gate_evals_per_step": round(float(stats[6]) / K, 1), "tracks_per_scene": round(tracks_in / max(float(stats[2]), 1), 2),
                     "clusters_found_per_step": round(float(stats[7]) / K, 2)},
            "host": {"cores": cores, "gen_s": round(t_gen, 1)},
        }
        line.update(cpu)
        if parity is not None:
            line["parity"] = parity
        if posture is not None:
            line["posture_leg"] = posture
        if "cpu_baseline" in cpu:
            line["speedup_vs_cpu_baseline"] = round(line["value"] / cpu["cpu_baseline"]["value"], 1)
            line["speedup_vs_cpu_native"] = round(line["value"] / cpu["cpu_baseline_native"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    sb.close()
    _ = gathered


if __name__ == "__main__":
    main()
