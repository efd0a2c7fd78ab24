"""End-to-end legs of bench.py (BASELINE.json configs[3]/[4]): `TrackBuffer.track` followed by
`TrackBuffer.estimate_posture` EVERY frame (reference offline_main.py:57-60, Tracking.py:705-734),
batched over the scenes of a context and pipelined by `mmwave_msc_amd.posture.PosturePipeline`.

* `e2e_leg`          -- the bench workload again from frame 0 with the CNN in the loop: scene-frames/s,
                        MFMA roofline of the CNN, HBM rate of the feature kernel.
* `oracle_reference` -- CPU side of the configs[3] parity leg (256 scenes x 256 points x 4 tracks): the C oracle's
                        tracker + feature map every frame and the fp64 numpy CNN (oracle/, test infrastructure: used
                        here only as the checker).
* `e2e_parity_leg`   -- the GPU side of it and the comparison: integers / fp64 state bit-equal, keypoints <= 1e-4.
"""
import time

import numpy as np

MFMA_FP32_PEAK_TF = 157.3
MFMA_FP16_PEAK_TF = 2500.0   # dense fp16 / bf16 matrix-core peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
CNN_FLOP = {3: 25187328.0, 1: 2837504.0}
KP_TOL = 1e-4            # max|d| <= 1e-4 (m) and rel 1e-4 on the 57 outputs
E2E_PAR = dict(S=256, N=256, T=4, F=10, seed0=9000)
MARK = 12345.0


def e2e_leg(sb, step, W, F, S, world, barrier, max_over_ranks, dev, stream_a, modes=None):
    import torch

    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
    from mmwave_msc_amd.posture import PosturePipeline

    cap = S * min(sb.track_cap, 2 * sb.cfg.tr_max_tracks)
    weights = random_keras_weights(0, sb.ring)
    # (schedule, Dense-1 arithmetic): the default build of the product, then the two alternatives beside it
    all_modes = (("serial", "f16x3"), ("serial", "f32"), ("overlap", "f16x3"))
    run_modes = tuple(modes) if modes else all_modes
    models = {a: MarsCNN.from_keras_weights(weights, arith=a).to(dev) for a in sorted({a for _, a in run_modes})}
    out = {}
    for mode, arith in run_modes:
        model = models[arith]
        pipe = PosturePipeline(sb, model, cap, tracker_stream=stream_a, overlap=(mode == "overlap"), time_cnn=True)
        sb.reset()
        sb.profile(False)
        for f in range(W):
            step(f)
            pipe.after_step()
        pipe.drain()
        sb.check()
        sb.stats_reset()
        sb.profile_reset()
        pipe.reset_counters()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(W, F):
            # the feature kernel's own HIP-event pair on a sample of the frames (a pair idles the stream ~10 us)
            sb.profile(True, kernels=(_lib.K_FEATURES,)) if (f - W) % 4 == 0 else sb.profile(False)
            step(f)
            pipe.after_step()
        pipe.drain()
        torch.cuda.synchronize()
        barrier()
        el = max_over_ranks(time.perf_counter() - t0)
        pipe.close()   # (outside the clock: hands the tracker's side-stream workers back for the next mode's constructor to see)
        sb.profile(False)
        sb.check()
        K = F - W
        rows = pipe.rows_total
        ext = sb.stats_ext()
        f_ms, f_cnt = sb.profile_get(_lib.K_FEATURES)
        feat_bytes_per_launch = float(ext[30]) / max(K, 1)
        f_avg = f_ms / max(f_cnt, 1)
        cnn_ms = pipe.cnn_ms()
        flop = CNN_FLOP[3 if sb.ring == 3 else 1]
        split = arith == "f16x3" and model.use_hip_conv
        # matrix-core work actually issued: with split operands every multiply-add is three fp16 partial products
        issued = 3.0 * flop if split else flop
        peak = MFMA_FP16_PEAK_TF if split else MFMA_FP32_PEAK_TF
        res = {
            "value": round(S * world * K / el, 1), "unit": "scene-frames/s", "ms_per_step": round(el / K * 1e3, 4),
            "samples_per_step": round(rows / max(K, 1), 1), "samples_per_s": round(rows / el, 1), "streams": pipe.overlap_note,
            "cnn_ms_per_step": round(cnn_ms, 4) if cnn_ms is not None else None,
            "cnn_arith": ("f32 results from fp16 matrix cores: every fp32 operand split hi + 2^-11 lo' (fp16 halves), each product = 3 "
                          "exact partial products, fp32 accumulation (k_mars_conv16 + k_mars_dense1, this package's kernels); closer to the fp64 oracle than "
                          "fp32 arithmetic") if split else "fp32 on the fp32 matrix cores throughout (k_mars_conv + fp32 GEMM)",
            "roofline_cnn": {"bound": "mfma", "dtype": "f16 x3 (fp32-exact split)" if split else "f32",
                             "achieved": round(rows * issued / el / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                             "frac": round(rows * issued / el / 1e12 / peak, 6),
                             "flop_per_sample_algorithmic": flop, "flop_per_sample_issued": issued,
                             "algorithmic_tflops": round(rows * flop / el / 1e12, 2),
                             "cnn_only_algorithmic_tflops": round(rows / max(K, 1) * flop / (cnn_ms * 1e-3) / 1e12, 2) if cnn_ms else None,
                             "fp32_mfma_peak": MFMA_FP32_PEAK_TF,
                             "note": "achieved = matrix-core flops ISSUED over the whole end-to-end step time (tracker and feature kernel "
                                     "included); algorithmic_tflops = 25.19 MFLOP per sample over the same time, comparable with the fp32 "
                                     "matrix-core peak, the roof of the reference's own arithmetic"},
            "roofline_features": {"kernel": "k_features", "bound": "hbm", "achieved": round(feat_bytes_per_launch / max(f_avg, 1e-9) / 1e6, 2),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(feat_bytes_per_launch / max(f_avg, 1e-9) / 1e6 / HBM_PEAK_GBS, 6),
                                  "algorithmic_bytes_per_launch": round(feat_bytes_per_launch, 1), "avg_launch_ms": round(f_avg, 5),
                                  "launches_timed": int(f_cnt)},
        }
        out[(mode, arith)] = res
    modes = {"serial": "one stream: track(f), features(f), CNN(f), keypoints(f), track(f+1), ... (the CNN owns the whole chip)",
             "overlap": "two streams: track(f+1) + features(f+1) beside CNN(f), keypoints scattered by track creation ordinal"}
    if run_modes != all_modes:   # a caller's own selection (bench_ingest.py): the first one, as measured
        best = dict(out[run_modes[0]])
        best["mode"] = run_modes[0][0] + " -- " + modes[run_modes[0][0]]
        best["steps"] = F - W
        return best
    first, second = ("serial", "overlap") if out[("serial", "f16x3")]["value"] >= out[("overlap", "f16x3")]["value"] else ("overlap", "serial")
    best = dict(out[(first, "f16x3")])
    best["mode"] = first + " -- " + modes[first]
    best["other_schedule"] = {"mode": second + " -- " + modes[second],
                              **{k: out[(second, "f16x3")][k] for k in ("value", "ms_per_step", "cnn_ms_per_step", "streams")}}
    best["fp32_dense1"] = {"mode": "serial", **{k: out[("serial", "f32")][k] for k in ("value", "ms_per_step", "cnn_ms_per_step", "cnn_arith")}}
    best["config"] = (f"{S * world} scenes, track -> features -> MARS CNN (random Keras-layout weights) -> keypoints every frame; "
                      f"BASELINE.json configs[4] shape at {world} GPU(s)")
    best["steps"] = F - W
    return best


# ---------------------------------------------------------------------------------------------------------------
def oracle_reference(workers=1, par=None, procs=1):
    """configs[3] on the CPU oracle: per frame track -> features -> (CNN deferred) -> set_keypoints.  The CNN is a pure
    function of the feature tensor, so the oracle stores every tensor it would have fed to `model.predict`, tags the
    track with the tensor's index instead of real keypoints, and evaluates the fp64 CNN once at the end for the
    tensor each surviving track was tagged with last: identical to running it every frame, at a fraction of the cost.
    procs > 1: the scenes are cut into blocks that run in child processes started from scratch (scenes are independent; safe
    in a process that has already initialised the GPU) -- configs[4] at its full 4096 scenes in ~20 s on the GPU box's 16 cores."""
    from mmwave_msc_amd.marsweights import random_keras_weights

    p = dict(par or E2E_PAR)
    t0 = time.perf_counter()
    S = p["S"]
    procs = max(1, min(int(procs), S // 32 if S >= 64 else 1))
    if procs == 1:
        blocks = [_oracle_block((p, 0, S, workers))]
    else:
        # (child PROCESSES started from scratch -- `python bench_e2e.py --oracle-block ...` -- not forks of this one, which may
        #  hold a GPU context, and not multiprocessing's spawn, which re-imports the parent's __main__)
        import json
        import os
        import pickle
        import subprocess
        import sys
        import tempfile
        edges = [S * i // procs for i in range(procs + 1)]
        with tempfile.TemporaryDirectory() as tmp:
            jobs = []
            for i in range(procs):
                out = os.path.join(tmp, f"block{i}.pkl")
                # (one BLAS / OpenMP thread per child: the children ARE the parallelism)
                env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
                jobs.append((out, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--oracle-block",
                                                    json.dumps([p, edges[i], edges[i + 1], 1]), out], env=env)))
            blocks = []
            for out, job in jobs:
                if job.wait() != 0:
                    raise RuntimeError(f"oracle block process failed (exit code {job.returncode})")
                with open(out, "rb") as fh:
                    blocks.append(pickle.load(fh))
    cat = lambda k, ax: np.concatenate([b[k] for b in blocks], axis=ax)
    return {"par": p, "pts": cat("pts", 1), "cnt": cat("cnt", 1), "dts": cat("dts", 1),
            "finals": [f for b in blocks for f in b["finals"]], "want_kp": [k for b in blocks for k in b["want_kp"]],
            "weights": random_keras_weights(0, 3), "oracle_s": round(time.perf_counter() - t0, 1),
            "samples_cnn": sum(b["samples_cnn"] for b in blocks)}


def _oracle_block(args):
    """scenes [s0, s1) of oracle_reference's job (module level: a spawned worker imports it)"""
    from mmwave_msc_amd.marsweights import random_keras_weights   # (no torch in a worker)
    from oracle import c_oracle as co
    from oracle.mars_np import mars_forward_np
    import bench

    p, s0, s1, workers = args
    S, N, T, F = s1 - s0, p["N"], p["T"], p["F"]
    ids = np.arange(p["seed0"] + s0, p["seed0"] + s1)
    pts, cnt, dts = bench.generate(ids, F, N, T, workers)
    cfg = co.default_config(tr_max_tracks=T)
    scenes = [co.OracleScene(cfg, N) for _ in range(S)]
    store = []
    for f in range(F):
        for s, sc in enumerate(scenes):
            c = int(cnt[f, s])
            if c == 0:
                continue
            sc.track(pts[f, s, :c].astype(np.float64), float(dts[f, s]))
            feat, owner = sc.features()
            if len(owner):
                tag = np.zeros((len(owner), 57), dtype=np.float32)
                tag[:, 0] = np.arange(len(store), len(store) + len(owner), dtype=np.float32)
                tag[:, 1] = MARK
                store.extend(feat)
                sc.set_keypoints(tag, owner)
    assert len(store) < (1 << 24)
    finals = [sc.tracks() for sc in scenes]
    w = random_keras_weights(0, 3)
    need = sorted({int(r["keypoints"][0]) for fin in finals for r in fin if r["keypoints"][1] == MARK})
    kp_of = {}
    for i in range(0, len(need), 256):
        idx = need[i:i + 256]
        kp = mars_forward_np(w, np.stack([store[j] for j in idx]).astype(np.float64))
        kp_of.update(zip(idx, kp))
    default = np.array(list(cfg.default_posture), dtype=np.float64)
    want_kp = [np.stack([kp_of[int(r["keypoints"][0])] if r["keypoints"][1] == MARK else default for r in fin])
               if len(fin) else np.zeros((0, 57)) for fin in finals]
    return {"pts": pts, "cnt": cnt, "dts": dts, "finals": finals, "want_kp": want_kp, "samples_cnn": len(need)}


def e2e_parity_leg(ref, device):
    import torch

    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd.mars import MarsCNN
    from mmwave_msc_amd.posture import PosturePipeline

    p = ref.get("par") or E2E_PAR
    S, N, T, F = p["S"], p["N"], p["T"], p["F"]
    dev = torch.device("cuda", device)
    sb = SceneBatch(_lib.default_config(tr_max_tracks=T), S, N, device=device)
    model = MarsCNN.from_keras_weights(ref["weights"]).to(dev)
    pipe = PosturePipeline(sb, model, S * min(sb.track_cap, 2 * T))
    with torch.cuda.stream(pipe.A):
        d_pts = torch.from_numpy(ref["pts"]).to(dev).double()
        d_cnt = torch.from_numpy(ref["cnt"]).to(dev)
        d_dt = torch.from_numpy(ref["dts"]).to(dev)
    pipe.A.synchronize()

    def run():
        sb.reset()
        for f in range(F):
            sb.step_dev(d_pts[f].data_ptr(), d_cnt[f].data_ptr(), d_dt[f].data_ptr())
            pipe.after_step()
        pipe.drain()

    run()            # first pass: also warms hipBLASLt / the allocator
    sb.check()
    t0 = time.perf_counter()
    run()
    el = time.perf_counter() - t0
    sb.check()
    ntr = sb.num_tracks()
    trk = sb.tracks(cap=max(int(ntr.max()), 1))
    ints_ok, max_err, n_kp = True, 0.0, 0
    for s, want in enumerate(ref["finals"]):
        got = trk[s, : ntr[s]]
        ints_ok &= len(want) == int(ntr[s])
        if not ints_ok:
            break
        for name in ("x", "P", "centroid", "spread_est", "group_disp_est", "lifetime", "point_num", "is_static", "ring_n"):
            ints_ok &= bool(np.array_equal(got[name], want[name]))
        if len(want):
            wk = ref["want_kp"][s]
            err = np.abs(got["keypoints"].astype(np.float64) - wk) / np.maximum(1.0, np.abs(wk))
            max_err = max(max_err, float(err.max()))
            n_kp += len(want)
    pipe.close()
    sb.close()
    return {"config": f"{S} scenes x {N} pts x TR_MAX_TRACKS={T}, {F} frames, track+features+CNN+keypoints every frame; BASELINE.json {p.get('label', 'configs[3]')}",
            "tracker_state_bit_equal_vs_oracle": bool(ints_ok), "tracks_checked": int(n_kp),
            "keypoint_max_err": float(f"{max_err:.3e}"), "keypoint_tol": KP_TOL, "keypoints_ok": bool(ints_ok and max_err <= KP_TOL),
            "cnn_oracle": "oracle/mars_np.py (fp64 numpy restatement of train.py:71-106; parity unpinned: no Keras / MARS.h5 in the image)",
            "ms_per_step": round(el / F * 1e3, 4), "scene_frames_per_sec": round(S * F / el, 1), "oracle_s": ref["oracle_s"]}


if __name__ == "__main__":
    import sys
    if len(sys.argv) == 4 and sys.argv[1] == "--oracle-block":   # a worker of oracle_reference(procs > 1): CPU only
        import json
        import pickle
        a = json.loads(sys.argv[2])
        with open(sys.argv[3], "wb") as fh:
            pickle.dump(_oracle_block(tuple(a)), fh, protocol=pickle.HIGHEST_PROTOCOL)
