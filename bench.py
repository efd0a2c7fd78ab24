#!/usr/bin/env python3
"""Benchmark of the hot path: scene-frames/s of `TrackBuffer.track` (DBSCAN + gating /
association + Kalman) over S concurrent synthetic scenes.

    python bench.py --gpus N --steps K --warmup W [--scaling strong|weak]

N > 1 needs no launcher: when WORLD_SIZE is unset the script starts one child process per GPU
itself (before anything in this process touches the GPU) and prints rank 0's line; under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` the ranks it is
given are used as they are.

A "step" = one radar frame for every scene of the job (one `mmw_step` per rank).  Inputs for
all W+K frames are resident in HBM before the timed region.  One JSON line on rank 0.

Workload: BASELINE.json configs[2] -- 4096 scenes x 512 points, TR_MAX_TRACKS = 8 -- with every
scene holding K = T = 8 walking targets, SURVEY.md §8(d)'s population (`--population full`, the
default since round 5: 7.75 tracks per scene, 16 M gate evaluations per step).  The population of
rounds 1-4 -- scene s holds 1 + (s mod 8) targets, so that 7/8 of the scenes keep calling
apply_DBscan every frame, 4.5 tracks per scene -- is the second leg of the line (`mixed_population`;
`--population mixed` makes it the headline again and K = T the second leg, `full_tracks`).
Scaling: "strong" (default for N > 1) shards the 4096 scenes of configs[2]/[4] over the ranks;
"weak" gives every rank `--scenes` scenes.  No data-path collective either way; one RCCL
all-gather of the track table closes the timed region when N > 1.

Beside `value` (tracker only, configs[1]/[2]) the line carries `e2e`: the same scenes stepped
through track -> features -> MARS CNN -> keypoints EVERY frame (configs[3]/[4], Tracking.py:
705-734 after every track()), with its own scene-frames/s, the CNN's MFMA roofline and the
feature kernel's HBM rate; `e2e_parity`: configs[3] (256 scenes x 256 points x 4 tracks) end to
end against the oracle (ints bit-equal, keypoints <= 1e-4); `cold_start`: frames 0..3 of fresh
scenes, where every scene clusters its whole ring (the BallTree DBSCAN kernels' own number).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mmwave_msc_amd.shard import shard_range  # noqa: E402  (no torch, no GPU)
from mmwave_msc_amd.synth import make_scene  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
MFMA_FP32_PEAK_TF = 157.3   # dense fp32 matrix-core peak (MI355X_MICROARCH.md)
VALU_FP64_PEAK_TF = 78.6    # fp64 vector peak with FMA (the tracker is built with -ffp-contract=off: half of it is reachable)
CNN_FLOP = {3: 25187328.0, 1: 2837504.0}   # per sample: define_CNN_3D / define_CNN


def _gen_one(args):
    sid, frames, n_pts, n_targets = args
    return make_scene(sid, frames, n_pts, n_targets)


def generate(scene_ids, frames, n_pts, tracks, workers, population="mixed"):
    """points[F,S,N,8] float32, counts[F,S] int32, dt[F,S] float64 for the given global scene ids.  population "mixed":
    scene s holds 1 + (s mod T) targets (the headline workload); "full": every scene holds T targets (SURVEY.md
    §8(d)'s K = T: every track spawns in the first frames and apply_DBscan is never called again)."""
    jobs = [(int(s), frames, n_pts, tracks if population == "full" else 1 + int(s) % tracks) for s in scene_ids]
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


def effective_cores():
    """CPUs this process may actually use: the smallest of the logical CPU count, the affinity mask and the cgroup
    CPU quota (a container on a 256-thread host may be capped at 16 CPUs' worth of time: 256 busy processes then
    only throttle each other)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]      # cgroup v2: "max 100000" or "1600000 100000"
        if a != "max":
            quota = float(a) / float(b)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def cpu_legs(pts, cnt, dts, tracks, cores, W, py_scenes_per_core, py_frames, c_scenes):
    """CPU baselines on a bounded sample of the SAME workload and the SAME frame window as the GPU (frames
    W.. of every sampled scene; frames 0..W-1 advance the state untimed), before the GPU is touched.
    Returns (dict of the two legs, oracle final states for the parity check)."""
    from oracle import c_oracle as co
    from oracle.py_tracker import run_window_multiprocess

    F, S = pts.shape[0], pts.shape[1]
    out = {}
    # (i) reference-shaped Python restatement: one process per host core, BLAS threads pinned to 1, pool up and
    #     warm-up frames done before the clock starts (a barrier separates the two phases in every worker)
    procs = max(1, min(cores, S))
    ns = min(S, procs * max(1, py_scenes_per_core))
    nf = max(1, min(F - W, py_frames))
    el, _ = run_window_multiprocess({"TR_MAX_TRACKS": tracks}, pts[:W + nf, :ns], cnt[:W + nf, :ns], dts[:W + nf, :ns], procs, W)
    out["cpu_baseline"] = {
        "value": round(ns * nf / el, 2), "unit": "scene-frames/s", "cores": procs, "kind": "port",
        "sample": f"oracle/py_tracker.py (numpy + per-point inv/det + sklearn DBSCAN with the Python metric, as the reference): "
                  f"scenes 0..{ns - 1} of this workload, frames {W}..{W + nf - 1} timed after frames 0..{W - 1} untimed "
                  f"(the GPU's window starts at frame {W} too), {procs} processes x 1 BLAS thread, pool started before the clock, {el:.1f} s wall",
    }
    # (ii) plain-C oracle, OpenMP over scenes, same window (also yields the parity reference over all F frames)
    cfg = co.default_config(tr_max_tracks=tracks)
    nc = min(S, c_scenes)
    ob = co.OracleBatch(cfg, nc, pts.shape[2])
    threads = max(1, min(cores, co.max_threads(), nc))
    sub_pts = np.ascontiguousarray(pts[:, :nc])
    sub_cnt, sub_dt = np.ascontiguousarray(cnt[:, :nc]), np.ascontiguousarray(dts[:, :nc])
    if W > 0:
        co.batch_run_f32(ob, sub_pts[:W], sub_cnt[:W], sub_dt[:W], threads)
    t0 = time.perf_counter()
    co.batch_run_f32(ob, sub_pts[W:], sub_cnt[W:], sub_dt[W:], threads)
    elc = time.perf_counter() - t0
    out["cpu_baseline_native"] = {
        "value": round(nc * (F - W) / elc, 1), "unit": "scene-frames/s", "cores": threads, "kind": "port",
        "sample": f"oracle/c (plain C, OpenMP, each thread runs whole scenes): scenes 0..{nc - 1}, frames {W}..{F - 1} timed after "
                  f"frames 0..{W - 1} untimed, {threads} threads, {elc:.2f} s wall",
    }
    finals = [sc.tracks() for sc in ob.scenes]
    return out, finals


# hipEvent-timed launches inside the timed region.  An event pair holds the stream for 10-20 us (a trace of the sampled steps,
# scripts/trace_chain_start.sh: 6.6 + 4.5 + 7 us of gaps around the timed kernels), so the sample is thin: k_track -- the roofline
# kernel -- in one step of PROF_EVERY, and in the step half way between two of those ONE of the other three kernels in rotation;
# never two pairs in a step.  (Measured on one box, 40 steps: a pair of kernels in every 4th step cost 3.3 % of the headline --
# 0.2040 ms per step against 0.1973 with a single sample; this scheme costs ~1 %.)
PROF_EVERY = int(os.environ.get("MMW_BENCH_PROF_EVERY", "10"))
OTHER_KERNELS = (5, 1, 6)  # _lib.K_PREDICT, K_DBSCAN, K_POST


def profiled_kernels(i, n_steps):
    """Which kernels of timed step i (of n_steps) carry a HIP-event pair: () for most steps.  A short run samples more
    densely, so that each of the four kernels is seen at least once."""
    period = max(2, min(PROF_EVERY, n_steps // 4))
    if i % period == 0:
        return (0,)   # _lib.K_TRACK
    if i % period == period // 2:
        return (OTHER_KERNELS[(i // period) % len(OTHER_KERNELS)],)
    return ()
# algorithmic bytes per track of the two Kalman kernels: k_predict reads the 1232-byte record
# prefix, writes P and x (720) and the gate record (352); the update half of k_post reads the prefix and
# writes P and x
PREDICT_BYTES_PER_TRACK = 1232 + 720 + 352
UPDATE_BYTES_PER_TRACK = 1232 + 720
# fp64 operations per unit of work, counted from the kernels' arithmetic: one gate evaluation =
# 6 subtractions + 36 mul + 30 add (y'C^-1 y, dense) + 6 mul + 5 add (outer dot) + 1 add (log det) ...
FLOP_PER_GATE = 84.0
# k_predict per track: x' = F x and F P F' + Q over F's non-zero terms (~900), C = P[:6,:6] + diag + gd (72),
# pivoted 6x6 LU + inverse + log det (~580); update half of k_post: K = P H' S^-1 (648), S^-1 (~580), x += K y (114),
# Joseph form (I-KH) P (I-KH)' (2916) + K R K' (1620)
FLOP_PER_TRACK_PREDICT = 1550.0
FLOP_PER_TRACK_UPDATE = 5900.0
# per point of a frame: 6 column-sum adds + 12 min/max compares + centred 21-entry dispersion products (6 sub, 21 mul, 21 add)
FLOP_PER_POINT_STATS = 66.0


def population_label(population, T):
    return (f"EVERY scene holds {T} targets (SURVEY.md §8d: K = T)" if population == "full"
            else f"targets per scene = 1 + (scene_id mod {T}) (the headline population of rounds 1-4)")


def workload_label(S_total, N, T, world, scaling, S_rank, population="full"):
    tag = ""
    if (S_total, N, T) == (4096, 512, 8):
        tag = "; BASELINE.json configs[2]"
    elif (S_total, N, T) == (256, 256, 4):
        tag = "; BASELINE.json configs[1]"
    return (f"{S_total} scenes x {N} pts x TR_MAX_TRACKS={T}, DBSCAN+gating+KF (TrackBuffer.track), "
            f"{population_label(population, T)}{tag}")


def self_launch(args, argv):
    """One child per GPU, started before this process has touched the GPU; rank 0's stdout is passed through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for raw in procs[0].stdout:
            line = raw.decode(errors="replace")
            # the JSON line is the contract; library chatter that lands on rank 0's stdout goes to stderr
            out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            out.write(line)
            out.flush()
        for p in procs:
            p.wait()
            rc = rc or p.returncode
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def dry_run(rank, world, lo, hi, S_total, scaling, slots):
    """The N > 1 plumbing without a GPU: process group (gloo), this rank's shard, the all-gather of a track table whose
    rows carry their global scene id, and the rank-0 line."""
    import torch
    import torch.distributed as dist

    from mmwave_msc_amd._lib import SUMMARY_DTYPE
    from mmwave_msc_amd.dist import ShardedTracker, tensor_to_summaries

    if world > 1:
        dist.init_process_group(backend="gloo")

    class TableOnly:   # stands in for the rank-local SceneBatch: a track table whose rows carry their global scene id
        def __init__(self, cfg, n_local, max_pts, device):
            self.S = n_local

        def track_table_host(self, n_slots, scene_base=0):
            tab = np.zeros((self.S, n_slots), dtype=SUMMARY_DTYPE)
            tab["scene"] = (scene_base + np.arange(self.S))[:, None]
            tab["slot"] = np.arange(n_slots)[None, :]
            return tab

    n_arg = S_total if scaling == "strong" else S_total // world
    st = ShardedTracker(None, n_arg, 1, scaling=scaling, batch_factory=TableOnly)
    assert (st.lo, st.hi, st.n_total) == (lo, hi, S_total), (st.lo, st.hi, st.n_total, lo, hi, S_total)
    glob = tensor_to_summaries(st.gather_table(slots), slots)
    ok = glob.shape[0] == S_total and bool(np.array_equal(glob["scene"][:, 0], np.arange(S_total)))
    if world > 1:
        t = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item() == 1.0)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "n_ranks_seen": dist.get_world_size() if world > 1 else 1, "scaling": scaling,
                          "scenes_total": S_total, "scenes_rank0": hi - lo, "gathered_table_rows": int(glob.shape[0]), "gather_ok": ok}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scenes", type=int, default=4096, help="scenes of the job (strong) / per GPU (weak)")
    ap.add_argument("--pts", type=int, default=512)
    ap.add_argument("--tracks", type=int, default=8, help="TR_MAX_TRACKS")
    ap.add_argument("--scaling", choices=("strong", "weak"), default=None,
                    help="strong (default): --scenes is the whole job, sharded over the GPUs (BASELINE configs[2]/[4]); "
                         "weak: every GPU owns --scenes scenes")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--py-scenes-per-core", type=int, default=4)
    ap.add_argument("--py-frames", type=int, default=20)
    ap.add_argument("--c-scenes", type=int, default=4096)
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (tracker + features + CNN every frame) leg")
    ap.add_argument("--no-e2e-parity", action="store_true", help="skip the configs[3] end-to-end parity leg")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-start (frames 0..3) leg")
    ap.add_argument("--chain-side-stream", type=int, default=0, choices=(-1, 0, 1, 2, 3),
                    help="mmw_config.chain_side_stream: 0 = the library's choice (on above 512 scenes), 1 = on, -1 = off, "
                         "2 = on without the concurrency probe (counter collection serialises kernels)")
    ap.add_argument("--no-shards", action="store_true", help="skip the shard legs (one rank's share of the 2/4/8-GPU job on this GPU)")
    ap.add_argument("--population", choices=("full", "mixed"), default="full",
                    help="targets per scene of the headline workload: full (default) = every scene holds TR_MAX_TRACKS targets (SURVEY.md "
                         "§8(d): K = T), mixed = scene s holds 1 + (s mod TR_MAX_TRACKS) (the headline of rounds 1-4); the other one is the second leg")
    ap.add_argument("--no-full", action="store_true", help="skip the second population's leg")
    ap.add_argument("--rows", choices=("f32", "f64"), default="f64",
                    help="how the resident frames are stored in HBM: f64 (default: 64 B per point, mmw_step -- the reference's float64 arrays, "
                         "what rounds 1-3 timed) or f32 (32 B per point, mmw_step_f32 promotes them to fp64 as they are loaded -- exact, the "
                         "synthetic rows are fp32-representable; every output is bit-equal; the step is not faster for it, k_track is bound by "
                         "its chains, not by bytes: profiles/NOTEBOOK.md)")
    ap.add_argument("--no-single", action="store_true", help="skip the single-scene leg (configs[0]: the offline loop on one synthetic CSV experiment: bench_single.py)")
    ap.add_argument("--no-ingest", action="store_true", help="skip the host-fed legs (frames from pinned host memory every step: bench_ingest.py)")
    ap.add_argument("--fused-step", type=int, default=0, choices=(-1, 0, 1),
                    help="mmw_config.fused_step: 0 = the library's choice (k_scene for contexts whose scenes are all resident), 1 = on, -1 = off")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / shard / gather plumbing only, on the CPU with gloo (tests/test_dist_gloo.py): no GPU work, no metric")
    ap.add_argument("--gen-workers", type=int, default=-1,
                    help="processes for scene generation (-1 = auto; use 1 under rocprofv3: its preloaded tool initialises "
                         "the GPU before main(), and forking after that hangs)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    scaling = args.scaling or "strong"
    N, K, W = args.pts, args.steps, args.warmup
    if scaling == "strong":
        S_total = args.scenes
        lo, hi = shard_range(S_total, rank, world)
    else:
        S_total = args.scenes * world
        lo, hi = rank * args.scenes, (rank + 1) * args.scenes
    S = hi - lo
    if S < 1:
        raise SystemExit(f"rank {rank}: no scenes to own ({S_total} scenes over {world} ranks)")
    F = K + W
    cores = effective_cores()
    single = world == 1 and rank == 0

    if args.dry_run:
        return dry_run(rank, world, lo, hi, S_total, scaling, args.tracks)

    # ---- host-side generation and CPU legs: nothing below touches the GPU yet ----
    t_gen = time.perf_counter()
    ids = np.arange(lo, hi)
    workers = args.gen_workers if args.gen_workers > 0 else max(1, min(32, cores // max(world, 1)))
    pts, cnt, dts = generate(ids, F, N, args.tracks, workers=workers, population=args.population)
    other_pop = "mixed" if args.population == "full" else "full"
    t_gen = time.perf_counter() - t_gen
    cpu, finals = {}, None
    if single and not args.no_cpu:
        cpu, finals = cpu_legs(pts, cnt, dts, args.tracks, cores, W, args.py_scenes_per_core, args.py_frames, args.c_scenes)
    full_host, cpu_full, finals_full = None, {}, None
    if single and not args.no_full:
        full_host = generate(ids, F, N, args.tracks, workers=workers, population=other_pop)
        if not args.no_cpu:
            cpu_full, finals_full = cpu_legs(*full_host, args.tracks, cores, W, args.py_scenes_per_core, args.py_frames, args.c_scenes)
    def cpu_e2e_leg(p_, c_, d_, tracks_, what):
        """the end-to-end loop on the CPU port: oracle/py_tracker.py + reference-shaped relative_coordinates / format_single_frame
        + oracle/mars_torch.py (fp32, torch's CPU operators), one process per host core, on a bounded sample of the leg's scenes
        and the GPU leg's own frame window start"""
        from oracle.py_tracker import run_window_multiprocess
        from mmwave_msc_amd.mars import random_keras_weights
        procs = max(1, min(cores, p_.shape[1]))
        ns = min(p_.shape[1], procs * 6)
        w_ = min(W, p_.shape[0] // 2)
        nf = max(1, min(p_.shape[0] - w_, 24))
        el_, _, rows_ = run_window_multiprocess({"TR_MAX_TRACKS": tracks_}, p_[:w_ + nf, :ns], c_[:w_ + nf, :ns], d_[:w_ + nf, :ns], procs, w_,
                                                posture_weights=random_keras_weights(0, 3), want_rows=True)
        return {"value": round(ns * nf / el_, 2), "unit": "scene-frames/s", "cores": procs, "kind": "port", "samples_per_s": round(rows_ / el_, 1),
                "sample": f"oracle/py_tracker.py (track + py_estimate_posture: the reference's relative_coordinates / format_single_frame shape) + "
                          f"oracle/mars_torch.py (Keras' fp32 on torch's CPU operators, 1 thread per process): scenes 0..{ns - 1} of {what}, frames "
                          f"{w_}..{w_ + nf - 1} timed after frames 0..{w_ - 1} untimed, {procs} processes, {el_:.1f} s wall"}

    e2e_cpu = None
    if single and not args.no_cpu and not args.no_e2e:
        e2e_cpu = cpu_e2e_leg(pts, cnt, dts, args.tracks, "this workload")
    e2e_ref = None
    if single and not args.no_e2e_parity:
        from bench_e2e import oracle_reference  # CPU side of the configs[3] leg (oracle), before the GPU is initialised
        e2e_ref = oracle_reference(workers)
        if not args.no_cpu:
            e2e_ref["cpu_baseline"] = cpu_e2e_leg(e2e_ref["pts"], e2e_ref["cnt"], e2e_ref["dts"], e2e_ref["par"]["T"], "the configs[3] scenes")
    single_prep = None
    if single and not args.no_single:
        try:
            from bench_single import single_scene_cpu   # configs[0]: the CSV experiment, the CPU port and the checker
            single_prep = single_scene_cpu()
        except Exception as exc:
            single_prep = {"error": repr(exc)[:300]}

    # ---- GPU ----
    import torch
    import torch.distributed as dist

    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.dist import ShardedTracker

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
    n_ranks_seen = dist.get_world_size() if world > 1 else 1
    # this rank's share of the job: the product's own object (mmwave_msc_amd/dist.py) -- rank-local SceneBatch + the all-gather
    shard = ShardedTracker(_lib.default_config(tr_max_tracks=args.tracks, chain_side_stream=args.chain_side_stream, fused_step=args.fused_step),
                           args.scenes, N, scaling=scaling, device=local_rank, rank=rank, world=world)
    assert (shard.lo, shard.hi, shard.n_total) == (lo, hi, S_total)
    sb = shard.sb
    # one real stream for torch and the context: uploads, the CNN of the posture leg and the mmw_* calls on device
    # tensors are then ordered by the stream itself (torch's default stream would read as "context's own stream")
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    sb.follow_torch_stream(side)
    rows32 = args.rows == "f32"
    row_dtype = torch.float32 if rows32 else torch.float64

    def resident(host_frames):
        """the frames of a leg in HBM before its clock starts: as they are (fp32 rows) or promoted to fp64"""
        d = torch.empty(host_frames.shape, dtype=row_dtype, device=dev)
        for f_ in range(host_frames.shape[0]):
            t_ = torch.from_numpy(host_frames[f_]).to(dev)
            d[f_] = t_ if rows32 else t_.double()
        return d

    def stepper(ctx):
        return ctx.step_dev_f32 if rows32 else ctx.step_dev

    d_pts = resident(pts)
    d_cnt = torch.from_numpy(cnt).to(dev)
    d_dt = torch.from_numpy(dts).to(dev)
    d_assoc = torch.empty((S, N), dtype=torch.int32, device=dev)
    d_lab = torch.empty((S, sb.UM), dtype=torch.int32, device=dev)
    d_dbn = torch.empty((S,), dtype=torch.int32, device=dev)
    slots = args.tracks

    def step(f):
        stepper(sb)(d_pts[f].data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr(),
                    d_assoc.data_ptr(), d_lab.data_ptr(), d_dbn.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    # ---- cold start: frames 0..3 of fresh scenes (every scene clusters its whole ring: the BallTree kernels) ----
    cold = None
    if not args.no_cold and F >= 4:
        for f in range(4):   # an untimed pass first: code objects, LDS configuration and the side stream exist afterwards
            step(f)
        torch.cuda.synchronize()
        sb.reset()
        sb.profile_reset()
        sb.stats_reset()
        sb.profile(True)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for f in range(4):
            step(f)
        torch.cuda.synchronize()
        tc = time.perf_counter() - tc
        sb.profile(False)
        st = sb.stats()
        pk = {k: sb.profile_get(k) for k in (_lib.K_DBSCAN, _lib.K_POST, _lib.K_TRACK, _lib.K_PREDICT)}
        db_ms = pk[_lib.K_DBSCAN][0]
        cold = {"frames": 4, "ms_total": round(tc * 1e3, 3), "ms_per_frame": [],
                "k_dbscan_big_ms": round(db_ms, 3), "k_post_ms": round(pk[_lib.K_POST][0], 3), "k_track_ms": round(pk[_lib.K_TRACK][0], 3),
                "dbscan_calls": int(st[3]), "mean_U": round(float(st[4]) / max(float(st[3]), 1), 1), "clusters_found": int(st[7]),
                "dbscan_alg_bytes": int(st[1]),
                "roofline_dbscan": {"kernel": "k_dbscan_big", "bound": "hbm", "achieved": round(float(st[1]) / max(db_ms, 1e-9) / 1e6, 2),
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(float(st[1]) / max(db_ms, 1e-9) / 1e6 / HBM_PEAK_GBS, 6),
                                    "note": "BallTree DBSCAN is a latency chain per cloud (build levels, queries, label propagation), "
                                            "not a stream: algorithmic bytes (68 B per clustered point + spawned records) over kernel time"}}
        sb.reset()
        torch.cuda.synchronize()
        # per-frame wall times of a second cold pass (each frame synchronised: includes launch latency)
        for f in range(4):
            t1 = time.perf_counter()
            step(f)
            torch.cuda.synchronize()
            cold["ms_per_frame"].append(round((time.perf_counter() - t1) * 1e3, 3))
        sb.reset()
        torch.cuda.synchronize()

    for f in range(W):
        step(f)
    shard.gather_table(slots)   # (untimed: allocates the table and the collective's buffers; the timed gather below reuses them)
    torch.cuda.synchronize()
    sb.check()
    sb.stats_reset()
    sb.profile_reset()
    sb.profile(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(W, F):
        # the live kernel durations come from a thin sample of the timed launches (profiled_kernels above)
        pk = profiled_kernels(f - W, F - W)
        if pk:
            sb.profile(True, kernels=pk)
        else:
            sb.profile(False)
        step(f)
    sb.profile(True)
    gathered = shard.gather_table(slots, reuse_out=True)   # (read once below: the cached output buffer will do)
    torch.cuda.synchronize()
    barrier()
    el = time.perf_counter() - t0
    sb.profile(False)
    el = max_over_ranks(el)
    sb.check()
    stats = sb.stats()
    side_workers = {0: "off", 1: "on", 2: "unchecked"}[sb.side_workers()]   # the DBSCAN chain workers on their side streams
    prof = {k: sb.profile_get(k) for k in (_lib.K_TRACK, _lib.K_DBSCAN, _lib.K_TABLE, _lib.K_PREDICT, _lib.K_POST)}
    gathered_rows = int(gathered.shape[0])

    def final_state_equal(ctx, want_states):
        """final track state of the context's first len(want_states) scenes against the oracle's, bit for bit"""
        ntr_ = ctx.num_tracks()
        trk_ = ctx.tracks(cap=max(int(ntr_.max()), 1))
        for s_, want in enumerate(want_states):
            if len(want) != ntr_[s_]:
                return False
            got = trk_[s_, : ntr_[s_]]
            for name in ("x", "P", "centroid", "spread_est", "group_disp_est", "lifetime", "point_num", "is_static", "ring_n"):
                if not np.array_equal(got[name], want[name]):
                    return False
        return True

    def timed_window(ctx, p_dev, c_dev, t_dev, n_ctx):
        """W untimed + K timed steps of a context over the first n_ctx scenes of the resident frames (the same window as the
        headline: frames W..F-1 timed); returns (seconds, per-kernel average ms from the sampled HIP-event pairs)."""
        a_, l_, b_ = d_assoc[:n_ctx], d_lab[:n_ctx], d_dbn[:n_ctx]

        def st_(f):
            stepper(ctx)(p_dev[f][:n_ctx].data_ptr(), c_dev[f][:n_ctx].data_ptr(), t_dev[f][:n_ctx].data_ptr(),
                         a_.data_ptr(), l_.data_ptr(), b_.data_ptr())
        for f in range(W):
            st_(f)
        torch.cuda.synchronize()
        ctx.check()
        ctx.stats_reset()
        ctx.profile_reset()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for f in range(W, F):
            pk = profiled_kernels(f - W, F - W)
            if pk:
                ctx.profile(True, kernels=pk)
            else:
                ctx.profile(False)
            st_(f)
        torch.cuda.synchronize()
        el_ = time.perf_counter() - t1
        ctx.profile(False)
        ctx.check()
        pk_ = {name: ctx.profile_get(k) for k, name in ((_lib.K_PREDICT, "k_predict"), (_lib.K_TRACK, "k_track"),
                                                        (_lib.K_DBSCAN, "k_dbscan_big"), (_lib.K_POST, "k_post"))}
        return el_, {name: round(v[0] / max(v[1], 1), 5) for name, v in pk_.items()}

    parity = None
    if finals is not None:
        parity = {"scenes_checked": len(finals), "frames": F, "bit_equal_vs_oracle": bool(final_state_equal(sb, finals))}

    STEP_NAMES = {1: "k_scene + k_post workers", 2: "k_track (with _predict_all) + k_post", 4: "k_predict + k_track + k_post + k_dbscan_big"}

    # ---- shard legs: what ONE rank of the 2- / 4- / 8-GPU strong-scaling job runs (rank 0's scenes 0..S/G-1, same
    #      generator, same frame window), on this GPU; projected whole-node value = G x S_shard x K / t_shard ----
    shards = None
    if single and not args.no_shards and scaling == "strong" and S >= 8:
        shards = {}
        for G in (2, 4, 8):
            n_sh = shard_range(S_total, 0, G)[1]
            ctx = SceneBatch(_lib.default_config(tr_max_tracks=args.tracks, chain_side_stream=args.chain_side_stream, fused_step=args.fused_step),
                             n_sh, N, device=local_rank)
            ctx.follow_torch_stream(side)
            el_s, k_s = timed_window(ctx, d_pts, d_cnt, d_dt, n_sh)
            ent = {"scenes": n_sh, "ms_per_step": round(el_s / K * 1e3, 4), "scene_frames_per_s": round(n_sh * K / el_s, 1),
                   "projected_whole_node": round(G * n_sh * K / el_s, 1), "step_kernels": STEP_NAMES[ctx.step_kind()], "kernels_avg_ms": k_s}
            if finals is not None:
                nchk = min(n_sh, len(finals))
                ent["parity"] = {"scenes_checked": nchk, "frames": F, "bit_equal_vs_oracle": bool(final_state_equal(ctx, finals[:nchk]))}
            shards[f"{G}_gpus"] = ent
            ctx.close()
        shards["note"] = ("one rank's share of the G-GPU job (scenes 0..S/G-1) stepped on this one GPU; projected_whole_node = G x that "
                          "rank's scene-frames/s (no data-path collective; the once-per-run all-gather of 324 B per track is not in it)")

    # ---- the other population (headline K = T: every scene holds T targets, all tracks spawn in the first frames and apply_DBscan
    #      is hardly called again; mixed: scene s holds 1 + (s mod T), 0.55x the gate / Kalman work, 7/8 of the scenes cluster
    #      every frame) ----
    full = None
    if full_host is not None:
        fp_, fc_, fd_ = full_host
        d_fp = resident(fp_)
        d_fc, d_fd = torch.from_numpy(fc_).to(dev), torch.from_numpy(fd_).to(dev)
        ctx = SceneBatch(_lib.default_config(tr_max_tracks=args.tracks, chain_side_stream=args.chain_side_stream, fused_step=args.fused_step),
                         S, N, device=local_rank)
        ctx.follow_torch_stream(side)
        el_f, k_f = timed_window(ctx, d_fp, d_fc, d_fd, S)
        stf = ctx.stats()
        kt_ms = k_f["k_track"]
        full = {"workload": f"{S} scenes x {N} pts x TR_MAX_TRACKS={args.tracks}, " + population_label(other_pop, args.tracks),
                "value": round(S * K / el_f, 1), "unit": "scene-frames/s", "ms_per_step": round(el_f / K * 1e3, 4), "kernels_avg_ms": k_f,
                "step_kernels": STEP_NAMES[ctx.step_kind()],
                "gate_evals_per_step": round(float(stf[6]) / K, 1), "tracks_per_scene": round(float(stf[5]) / max(float(stf[2]), 1.0), 2),
                "dbscan_calls_per_step": round(float(stf[3]) / K, 1),
                "roofline": {"kernel": "k_track", "bound": "hbm", "achieved": round(float(stf[0]) / K / max(kt_ms, 1e-9) / 1e6, 2), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(float(stf[0]) / K / max(kt_ms, 1e-9) / 1e6 / HBM_PEAK_GBS, 6),
                             "algorithmic_bytes_per_launch": round(float(stf[0]) / K, 1), "avg_launch_ms": kt_ms}}
        if finals_full is not None:
            full["parity"] = {"scenes_checked": len(finals_full), "frames": F, "bit_equal_vs_oracle": bool(final_state_equal(ctx, finals_full))}
        full.update(cpu_full)
        if "cpu_baseline" in cpu_full:
            full["speedup_vs_cpu_baseline"] = round(full["value"] / cpu_full["cpu_baseline"]["value"], 1)
            full["speedup_vs_cpu_native"] = round(full["value"] / cpu_full["cpu_baseline_native"]["value"], 1)
        ctx.close()
        del d_fp

    # ---- end to end: the same scenes again from frame 0, estimate_posture after every track() ----
    e2e = None
    if not args.no_e2e:
        try:
            from bench_e2e import e2e_leg
            e2e = e2e_leg(sb, step, W, F, S, world, barrier, max_over_ranks, dev, side)
        except Exception as exc:  # never lose the headline line over the second leg
            e2e = {"error": repr(exc)[:300]}
    e2e_par = None
    if e2e_ref is not None:
        try:
            from bench_e2e import e2e_parity_leg
            e2e_par = e2e_parity_leg(e2e_ref, local_rank)
        except Exception as exc:
            e2e_par = {"error": repr(exc)[:300]}

    if isinstance(e2e, dict) and e2e_cpu is not None and "error" not in e2e:
        e2e["cpu_baseline"] = e2e_cpu
        e2e["speedup_vs_cpu_baseline"] = round(e2e["value"] / e2e_cpu["value"], 1)
    if isinstance(e2e_par, dict) and e2e_ref is not None and "cpu_baseline" in e2e_ref and "error" not in e2e_par:
        e2e_par["cpu_baseline"] = e2e_ref["cpu_baseline"]
        e2e_par["speedup_vs_cpu_baseline"] = round(e2e_par["scene_frames_per_sec"] / e2e_ref["cpu_baseline"]["value"], 1)
    # ---- configs[0]: one scene through the offline loop (CSV -> normalize_data -> track -> estimate_posture) ----
    single_leg = None
    if single_prep is not None:
        if "error" in single_prep:
            single_leg = single_prep
        else:
            try:
                from bench_single import single_scene_gpu
                single_leg = single_scene_gpu(single_prep, local_rank)
            except Exception as exc:
                single_leg = {"error": repr(exc)[:300]}

    # ---- host-fed: the same frames from pinned host memory every step (H2D one frame ahead; fp32 rows, fp64 rows, raw rows
    #      through mmw_normalize_f32; and the end-to-end form) ----
    ingest = None
    if single and not args.no_ingest:
        try:
            from bench_ingest import ingest_leg
            e2e_ms = e2e.get("ms_per_step") if isinstance(e2e, dict) else None
            ingest = ingest_leg(sb, pts, cnt, dts, d_cnt, d_dt, (d_assoc, d_lab, d_dbn), W, F, S, world, barrier, max_over_ranks, dev, side,
                                resident_ms=el / K * 1e3, e2e_resident_ms=e2e_ms, with_e2e=not args.no_e2e)
        except Exception as exc:  # never lose the headline line over a further leg
            ingest = {"error": repr(exc)[:300]}

    if rank == 0:
        total_sf = S_total * K
        # per-kernel device time from the sampled HIP-event pairs (profiled_kernels: a thin sample); algorithmic bytes
        # from the device counters, which cover all K steps
        n_samp = max(prof[_lib.K_TRACK][1], 1)
        step_ms = {k: prof[k][0] / max(prof[k][1], 1)
                   for k in (_lib.K_PREDICT, _lib.K_TRACK, _lib.K_DBSCAN, _lib.K_POST)}
        tracks_in = float(stats[5])  # sum over scene-frames of the tracks entering track()
        gate_evals = float(stats[6])
        step_bytes = {
            _lib.K_PREDICT: tracks_in * PREDICT_BYTES_PER_TRACK / K,
            _lib.K_TRACK: float(stats[0]) / K,
            _lib.K_DBSCAN: float(stats[1]) / K,
            _lib.K_POST: tracks_in * UPDATE_BYTES_PER_TRACK / K,
        }
        dom = max(step_ms, key=step_ms.get)
        dom_ms, dom_bytes = step_ms[dom], step_bytes[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # roofline.traffic = HBM bytes per launch from the PMC counters.  Counter collection needs its own rocprofv3 --pmc passes
        # (it serialises the kernels; this process cannot read the counters): `scripts/gpu_round.sh pmc` runs them on THIS
        # command and scripts/make_traffic_json.py records the source hash of the build they ran on.  The number is quoted
        # only when that hash is the running library's (mmw_version()); for any other build the line says null.
        traffic, traffic_src = None, "null: no PMC passes of this build (profiles/traffic.json absent or from other sources)"
        issue = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(tpath):
            try:
                tj = json.load(open(tpath))
                lib_hash = _lib.load().mmw_version().decode().rsplit("src:", 1)[-1]
                key = f"{S}x{N}x{args.tracks}"
                ent = tj.get(key, {}).get(_lib.load().mmw_kernel_name(dom).decode())
                if tj.get("src_hash") == lib_hash and isinstance(ent, dict) and tj.get("population", "mixed") != args.population:
                    traffic_src = (f"null: profiles/traffic.json was collected on the {tj.get('population', 'mixed')} population, this run is "
                                   f"--population {args.population}")
                elif tj.get("src_hash") == lib_hash and isinstance(ent, dict):
                    # FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, bytes per launch
                    traffic = ent.get("hbm_bytes_per_launch_fetch_x2")
                    traffic_src = (f"profiles/traffic.json[{key}]: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on "
                                   f"this build (src:{lib_hash}); {tj.get('source', '')}")
                elif tj.get("src_hash") != lib_hash:
                    traffic_src = (f"null: profiles/traffic.json was collected on src:{tj.get('src_hash')}, this library is src:{lib_hash} "
                                   f"(run scripts/gpu_round.sh pmc + scripts/make_traffic_json.py on this build)")
                # the third roof: vector-instruction ISSUE.  SQ_INSTS_VALU per launch (its own --pmc pass on this build, scripts/pmc_sq.sh)
                # over this run's timed duration, against the rate at which the chip issues fp64 vector instructions by wall clock
                # (scripts/ubench/ubench_f64: 2.05 ns per wave instruction and SIMD) -- for kernels that are neither bandwidth- nor
                # flop-bound this is the roof that says how much is left (an upper bound: a third of the instructions are 32-bit)
                vi, ceil = tj.get("valu_insts_per_launch", {}).get(key), tj.get("fp64_issue_ceiling")
                if tj.get("src_hash") == lib_hash and tj.get("population", "mixed") == args.population and isinstance(vi, dict) and ceil:
                    peak_gips = ceil["simds"] / ceil["ns_per_wave_instruction_and_simd"]   # G wave-instructions / s
                    issue = {"bound": "valu_issue_fp64", "unit": "G wave-instructions/s", "peak": round(peak_gips, 1), "kernels": {}}
                    for kid, nm in ((_lib.K_TRACK, "k_track"), (_lib.K_POST, "k_post"), (_lib.K_PREDICT, "k_predict")):
                        if nm in vi and step_ms[kid] > 0:
                            a = vi[nm] / (step_ms[kid] * 1e-3) / 1e9
                            issue["kernels"][nm] = {"valu_insts_per_launch": vi[nm], "achieved": round(a, 1), "frac": round(a / peak_gips, 4)}
                    issue["source"] = tj["valu_insts_per_launch"].get("source", "") + "; ceiling: " + ceil.get("source", "")
            except Exception:
                traffic = None
        # B_trk = 64N + 4N + 4U + 2*T*1200 + 64*U_new per scene-frame, with the run's own means
        sf = max(float(stats[2]), 1.0)
        mean_T = tracks_in / sf
        mean_U = float(stats[4]) / max(float(stats[3]), 1.0)
        db_frac = float(stats[3]) / sf      # apply_DBscan calls per scene-frame
        u_new = mean_U / sb.ring            # unassigned rows appended per frame ~ a ring-th of the clustered cloud
        row_b = 32.0 if rows32 else 64.0   # (SURVEY.md §8(d) prices the points at 64 B: fp64 rows; stored as fp32 they are 32)
        b_trk = row_b * N + 4.0 * N + 4.0 * mean_U * db_frac + 2.0 * mean_T * 1200.0 + 64.0 * u_new
        # fp64 operations of the step's tracker arithmetic, counted from the device work counters
        flop_track = gate_evals / K * FLOP_PER_GATE + (S * N) * FLOP_PER_POINT_STATS
        flop_step = flop_track + tracks_in / K * (FLOP_PER_TRACK_PREDICT + FLOP_PER_TRACK_UPDATE)
        line = {
            "metric": "scene_frames_per_sec", "value": round(total_sf / el, 1), "unit": "scene-frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(el / K * 1e3, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": workload_label(S_total, N, args.tracks, world, scaling, S, args.population),
                "population": args.population,
                "scenes_total": S_total, "scenes_per_gpu": S, "points_per_frame": N, "max_tracks": args.tracks, "frames_resident": F,
                "rows": ("fp32 in HBM (32 B per point), promoted to fp64 in registers by mmw_step_f32: exact, outputs bit-equal to the fp64 entry's"
                         if rows32 else "fp64 in HBM (64 B per point), mmw_step"),
                "parallelism": f"scenes sharded over {world} GPU(s), {scaling} scaling, no data-path collective; "
                               f"all-gather of the track table once per run",
                "n_ranks_seen": n_ranks_seen, "gathered_table_rows": gathered_rows,
            },
            "roofline": {
                "kernel": _lib.load().mmw_kernel_name(dom).decode(), "bound": "hbm",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(dom_bytes, 1),
                "avg_launch_ms": round(dom_ms, 5), "launches_timed": n_samp,
            },
            # all four launches of a step together, per SURVEY.md §8(d): algorithmic bytes of the step / step time
            "roofline_whole_step": {"bound": "hbm", "achieved": round(sum(step_bytes.values()) / (el / K) / 1e9, 2), "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": round(sum(step_bytes.values()) / (el / K) / 1e9 / HBM_PEAK_GBS, 6),
                                    "algorithmic_bytes_per_step": round(sum(step_bytes.values()), 1)},
            # the same step priced with SURVEY.md §8(d)'s own byte formula (one read + one write of a 1200-byte record
            # per track, no per-track ring rows): the kernels move more than that (4 launches re-read the records)
            "roofline_survey_bytes": {"bound": "hbm", "bytes_per_scene_frame": round(b_trk, 1),
                                      "achieved": round(b_trk * S / (el / K) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": round(b_trk * S / (el / K) / 1e9 / HBM_PEAK_GBS, 6)},
            "roofline_issue": issue,
            # the other roof SURVEY.md §8(d) names: fp64 vector arithmetic of k_track / of the whole step
            "roofline_valu": {"bound": "valu_fp64", "kernel": "k_track", "flop_per_launch": round(flop_track, 1),
                              "achieved": round(flop_track / (step_ms[_lib.K_TRACK] * 1e-3) / 1e12, 3) if step_ms[_lib.K_TRACK] > 0 else None,
                              "peak": VALU_FP64_PEAK_TF, "unit": "TFLOP/s",
                              "frac": round(flop_track / (step_ms[_lib.K_TRACK] * 1e-3) / 1e12 / VALU_FP64_PEAK_TF, 6) if step_ms[_lib.K_TRACK] > 0 else None,
                              "whole_step_tflops": round(flop_step / (el / K) / 1e12, 3),
                              "note": "peak counts FMA as 2; the tracker is compiled with -ffp-contract=off (fixed operation order), so half the peak is reachable"},
            "kernels": {
                name: {"avg_ms": round(step_ms[k], 5), "alg_bytes_per_launch": round(step_bytes[k], 1)}
                for k, name in ((_lib.K_PREDICT, "k_predict"), (_lib.K_TRACK, "k_track"),
                                (_lib.K_DBSCAN, "k_dbscan_big"), (_lib.K_POST, "k_post"))
            },
            "work": {"dbscan_calls_per_step": round(float(stats[3]) / K, 1), "mean_U": round(mean_U, 1),
                     "gate_evals_per_step": round(gate_evals / K, 1), "tracks_per_scene": round(mean_T, 2),
                     "clusters_found_per_step": round(float(stats[7]) / K, 2)},
            "side_workers": side_workers,
            "step_kernels": STEP_NAMES[sb.step_kind()],
            "host": {"cores": cores, "logical_cpus": os.cpu_count(), "gen_s": round(t_gen, 1),
                     "note": "cores = min(logical CPUs, affinity mask, cgroup CPU quota): what the CPU baselines can really use"},
        }
        line.update(cpu)
        if parity is not None:
            line["parity"] = parity
        if shards is not None:
            line["shards"] = shards
        if full is not None:
            line["full_tracks" if other_pop == "full" else "mixed_population"] = full
        if cold is not None:
            line["cold_start"] = cold
        if ingest is not None:
            line["ingest"] = ingest
        if single_leg is not None:
            line["single_scene"] = single_leg
        if e2e is not None:
            line["e2e"] = e2e
        if e2e_par is not None:
            line["e2e_parity"] = e2e_par
        if "cpu_baseline" in cpu:
            line["speedup_vs_cpu_baseline"] = round(line["value"] / cpu["cpu_baseline"]["value"], 1)
            line["speedup_vs_cpu_native"] = round(line["value"] / cpu["cpu_baseline_native"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    shard.close()


if __name__ == "__main__":
    sys.exit(main())
