"""GPU-vs-GPU dual run (DIAGNOSTIC, not product): catch a non-deterministic result.

Rounds 4-5 saw three GPU-vs-oracle state mismatches in ~85 000 fuzz runs, all while six processes shared the GPU, none
reproducible alone.  The fuzz compares the GPU with the oracle once per frame and throws the evidence away; this harness
runs every case TWICE on the GPU at the same time -- two contexts of one configuration in one process, each on its own
stream, fed the same device-resident frames, every step queued without a host wait -- and compares the two bit for bit:
per frame the association vector, the DBSCAN labels and call pattern and the track table (k_table: alive / static /
point_num / lifetime / x[9] / centroid[6] of every list position), at the end every field of every track, the rings and the
error words.  Deterministic kernels make two equal runs; ANY difference is the bug, and the first differing frame / scene /
field is dumped with both values, the queue words (mmw_diag_queue) and the error bits, the oracle is asked which side is
wrong, and the case is re-run `--rerun` times to see whether it repeats.  No oracle in the loop: GPU speed.

One invocation = `--procs` worker processes (each its own pair of contexts, its own seeds: they ARE each other's noise)
for `--seconds`; the parent never touches the GPU.  Bisect axes:

  --sync-every-step      mmw_synchronize on both contexts after every frame instead of twelve steps queued
  --procs N              how many processes share the GPU
  MMW_LIB_NAME=libmmw_hip_<diag>.so   a diagnostic build (e.g. `make DIAG=vgate DIAGFLAGS=-DMMW_DIAG_VGATE`: the gate reads its
                         records by vector loads instead of through the scalar cache)

RAS / ECC counters of the box are logged before and after (amd-smi / rocm-smi, whatever the box lets an ordinary user read).
Output: one JSON line per worker + a summary under gpurun_out/dual_run_<tag>.json; mismatch dumps as .npz beside it."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def ras_snapshot():
    out = {}
    for name, cmd in (("amd-smi metric --ecc", ["amd-smi", "metric", "--ecc", "--json"]),
                      ("amd-smi metric --ecc-blocks", ["amd-smi", "metric", "--ecc-blocks", "--json"]),
                      ("rocm-smi --showrasinfo all", ["rocm-smi", "--showrasinfo", "all"]),
                      ("rocm-smi --showretiredpages", ["rocm-smi", "--showretiredpages"])):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
            out[name] = (r.stdout + r.stderr)[-4000:]
        except Exception as e:   # noqa: BLE001
            out[name] = f"unavailable: {e!r}"
    return out


def first_difference(name, a, b):
    import numpy as np
    if a.dtype.names:
        for fld in a.dtype.names:
            d = first_difference(f"{name}.{fld}", a[fld], b[fld])
            if d:
                return d
        return None
    av, bv = np.ascontiguousarray(a), np.ascontiguousarray(b)
    raw_a, raw_b = av.view(np.uint8).reshape(-1), bv.view(np.uint8).reshape(-1)
    if raw_a.shape == raw_b.shape and np.array_equal(raw_a, raw_b):
        return None
    if av.shape != bv.shape:
        return dict(field=name, shape_a=list(av.shape), shape_b=list(bv.shape))
    item = av.dtype.itemsize
    k = int(np.flatnonzero(raw_a != raw_b)[0]) // item
    idx = [int(v) for v in np.unravel_index(k, av.shape)] if av.ndim else []
    n_diff = int((av.reshape(-1).view(f"V{item}") != bv.reshape(-1).view(f"V{item}")).sum()) if item in (1, 2, 4, 8) else -1
    return dict(field=name, index=idx, a=repr(av.reshape(-1)[k]), b=repr(bv.reshape(-1)[k]), elements_differing=n_diff)


def worker(args, wid):
    import numpy as np
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch
    from mmwave_msc_amd._lib import SUMMARY_DTYPE
    from tests._fuzz import ARMS, draw_case, plant_nonfinite, scene_inputs
    from tests._layouts import LAYOUTS, layout_kwargs

    t_end = time.time() + args.seconds
    seed = args.seed0 + wid * 1_000_000
    st = dict(worker=wid, cases=0, case_runs=0, frames=0, mismatches=0, reruns=0, rerun_mismatches=0, create_failed=0,
              error_cases=0, layouts={}, first_mismatches=[])
    tag = f"{args.tag}_w{wid}"

    def run_pair(case, kw, S, N, F, inputs, A=None, B=None, sync_every=False):
        """Both contexts through the F frames; returns (per-frame outputs of A, of B, final state of A, of B)."""
        pts, cnt, dts = inputs
        slots = A.track_cap
        UM = A.UM
        p64 = np.ascontiguousarray(pts.astype(np.float64))
        b_p = A.buf("dr_pts", p64.nbytes).upload(p64)
        b_c = A.buf("dr_cnt", cnt.nbytes).upload(np.ascontiguousarray(cnt))
        b_d = A.buf("dr_dt", dts.nbytes).upload(np.ascontiguousarray(dts))
        tab_b = S * slots * SUMMARY_DTYPE.itemsize
        outs = []
        for X in (A, B):
            o = dict(assoc=X.buf("dr_assoc", F * S * N * 4), labels=X.buf("dr_labels", F * S * UM * 4), dbn=X.buf("dr_dbn", F * S * 4),
                     table=X.buf("dr_table", F * tab_b))
            # (poison: a kernel that skips a write must not look equal by leftovers of the previous case)
            o["assoc"].upload(np.full(F * S * N, -7, np.int32)); o["labels"].upload(np.full(F * S * UM, -7, np.int32))
            o["dbn"].upload(np.full(F * S, -7, np.int32)); o["table"].upload(np.zeros(F * tab_b, np.uint8))
            outs.append(o)
        A.synchronize(); B.synchronize()
        for f in range(F):
            for X, o in ((A, outs[0]), (B, outs[1])):
                X.step_dev(b_p.ptr + f * S * N * 64, b_c.ptr + f * S * 4, b_d.ptr + f * S * 8, o["assoc"].ptr + f * S * N * 4,
                           o["labels"].ptr + f * S * UM * 4, o["dbn"].ptr + f * S * 4)
                X.track_table_dev(o["table"].ptr + f * tab_b, slots, 0)
            if sync_every:
                A.synchronize(); B.synchronize()
        A.synchronize(); B.synchronize()
        def read_back():
            res = []
            for X, o in ((A, outs[0]), (B, outs[1])):
                r = dict(assoc=o["assoc"].download((F, S, N), np.int32), labels=o["labels"].download((F, S, UM), np.int32),
                         dbn=o["dbn"].download((F, S), np.int32), table=o["table"].download((F, S, slots), SUMMARY_DTYPE))
                ntr = X.num_tracks()
                ln, rn = X.batch_ring()
                r.update(n_tracks=ntr, tracks=X.tracks(cap=max(int(ntr.max()), 1)), ring_len=ln, ring_n=rn, errors=X.errors(), queue=X.diag_queue())
                res.append(r)
            return res
        res = read_back()
        res.append(read_back)     # (the same device state again, for a mismatch: is it the state or the read-back that differs?)
        return res

    def compare(ra, rb, S, F, cnt):
        """First difference in frame order, then in the final state; None when equal."""
        for f in range(F):
            for key in ("dbn", "assoc", "labels", "table"):
                for s in range(S):
                    a, b = ra[key][f, s], rb[key][f, s]
                    if key == "assoc":
                        n = max(int(cnt[f, s]), 0)
                        a, b = a[:n], b[:n]
                    elif key == "labels":
                        n = max(int(ra["dbn"][f, s]), 0)
                        a, b = a[:n], b[:n]
                    d = first_difference(key, a, b)
                    if d:
                        d.update(frame=f, scene=s)
                        return d
        for key in ("n_tracks", "errors", "ring_len", "ring_n", "tracks"):
            a, b = ra[key], rb[key]
            if key == "tracks":   # (records beyond a scene's track count are not state)
                for s in range(S):
                    d = first_difference("tracks", a[s, : ra["n_tracks"][s]], b[s, : rb["n_tracks"][s]])
                    if d:
                        d.update(frame="final", scene=s)
                        return d
                continue
            d = first_difference(key, a, b)
            if d:
                d.update(frame="final")
                return d
        return None

    prog = os.path.join(ROOT, "gpurun_out", f"dual_run_{tag}.progress")
    while time.time() < t_end and seed < args.seed_end + wid * 1_000_000:
        seed += 1
        if seed % 20 == 0:    # (where a worker that dies was: the last line of this file)
            with open(prog, "a") as fh:
                fh.write(f"{seed} {time.time():.1f} {st['cases']} {st['mismatches']}\n")
        arm = ARMS[seed % 3]
        layout = LAYOUTS[(seed // 3) % 4]
        case = draw_case(seed, arm=arm, max_scenes=args.max_scenes)
        kw = dict(case["cfg"])
        for k, v in layout_kwargs(layout).items():
            kw.setdefault(k, v)
        S, N, F = case["S"], case["N"], case["F"]
        try:
            A = SceneBatch(_lib.default_config(**kw), S, N)
            B = SceneBatch(_lib.default_config(**kw), S, N)
        except _lib.MmwError:
            st["create_failed"] += 1
            continue
        st["cases"] += 1
        key = f"{layout}:kind{A.step_kind()}:dense{A.kalman_layout()}"
        st["layouts"][key] = st["layouts"].get(key, 0) + 1
        for rep in range(args.reps):
            if rep:
                A.reset(); B.reset()
            icase = dict(case, seed=(seed * 64 + rep) % (1 << 31))
            inputs = scene_inputs(icase)
            if rep % 4 == 3:
                plant_nonfinite(icase, inputs[0], inputs[1], rate=0.3)
            ra, rb, again = run_pair(case, kw, S, N, F, inputs, A, B, sync_every=args.sync_every_step)
            st["case_runs"] += 2
            st["frames"] += 2 * F
            st["error_cases"] += int(ra["errors"].any())
            d = compare(ra, rb, S, F, inputs[1])
            if d is None:
                continue
            st["mismatches"] += 1
            # read the SAME device state a second and a third time: a difference that is gone was in the read-back, not in the state
            ra2, rb2 = again()
            ra3, rb3 = again()
            d["reread"] = dict(a_vs_b_again=compare(ra2, rb2, S, F, inputs[1]), a1_vs_a2=compare(ra, ra2, S, F, inputs[1]),
                               b1_vs_b2=compare(rb, rb2, S, F, inputs[1]), a2_vs_a3=compare(ra2, ra3, S, F, inputs[1]),
                               b2_vs_b3=compare(rb2, rb3, S, F, inputs[1]))
            sc_err = d.get("scene")
            d["scene_had_error_bits"] = bool(isinstance(sc_err, int) and (ra["errors"][sc_err] or rb["errors"][sc_err]))
            kind = "readback" if d["reread"]["a_vs_b_again"] is None else "state"
            d["kind"] = kind
            st[kind + "_mismatches"] = st.get(kind + "_mismatches", 0) + 1
            d.update(seed=seed, rep=rep, arm=arm, layout=key, S=S, N=N, cfg=kw, t=round(time.time(), 1),
                     queue_a=[int(v) for v in ra["queue"]], queue_b=[int(v) for v in rb["queue"]],
                     errors_a=[int(v) for v in ra["errors"]], errors_b=[int(v) for v in rb["errors"]])
            # which side is wrong?  (the oracle, CPU -- only now)
            try:
                from oracle import c_oracle as co
                sc = d["scene"]
                orc = co.OracleScene(co.default_config(**case["cfg"]), N)
                want = []
                for f in range(F):
                    c = int(inputs[1][f, sc])
                    if c != 0:
                        try:
                            want.append(orc.track(inputs[0][f, sc, : max(c, 0)].astype(np.float64), float(inputs[2][f, sc]))[0])
                        except Exception as e:   # noqa: BLE001
                            want.append(repr(e))
                            break
                    else:
                        want.append(None)
                ot = orc.tracks()
                common = [n for n in ot.dtype.names if n in ra["tracks"].dtype.names]
                sub = lambda t: np.ascontiguousarray(t[common])   # noqa: E731
                fa = first_difference("tracks", sub(ra["tracks"][sc, : ra["n_tracks"][sc]]), sub(ot[: ra["n_tracks"][sc]])) if orc.n_tracks == ra["n_tracks"][sc] else "n_tracks"
                fb = first_difference("tracks", sub(rb["tracks"][sc, : rb["n_tracks"][sc]]), sub(ot[: rb["n_tracks"][sc]])) if orc.n_tracks == rb["n_tracks"][sc] else "n_tracks"
                d["vs_oracle_final"] = dict(a=fa, b=fb)
            except Exception as e:   # noqa: BLE001
                d["vs_oracle_final"] = f"oracle unavailable: {e!r}"
            path = os.path.join(ROOT, "gpurun_out", f"dual_run_{tag}_seed{seed}_rep{rep}.npz")
            try:
                np.savez_compressed(path, pts=inputs[0], cnt=inputs[1], dts=inputs[2],
                                    **{f"a_{k}": v for k, v in ra.items()}, **{f"b_{k}": v for k, v in rb.items()})
                d["dump"] = os.path.basename(path)
            except Exception as e:   # noqa: BLE001
                d["dump"] = repr(e)
            # does it repeat?
            rr = 0
            for _ in range(args.rerun):
                A.reset(); B.reset()
                xa, xb, _ = run_pair(case, kw, S, N, F, inputs, A, B, sync_every=args.sync_every_step)
                st["reruns"] += 1
                rr += int(compare(xa, xb, S, F, inputs[1]) is not None)
            d["rerun_mismatches"] = rr
            st["rerun_mismatches"] += rr
            if len(st["first_mismatches"]) < 20:
                st["first_mismatches"].append(d)
            print("MISMATCH " + json.dumps(d, default=str), flush=True)
        A.close(); B.close()
    print("WORKER " + json.dumps(st, default=str), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--seed0", type=int, default=90_000_000)
    ap.add_argument("--seed-end", type=int, default=1 << 60, help="stop after this seed (worker 0's numbering)")
    ap.add_argument("--reps", type=int, default=8, help="input sets per configuration (the contexts are reset in between)")
    ap.add_argument("--rerun", type=int, default=10, help="repeat runs of a mismatching case")
    ap.add_argument("--max-scenes", type=int, default=8)
    ap.add_argument("--sync-every-step", action="store_true")
    ap.add_argument("--no-ras", action="store_true", help="skip the amd-smi / rocm-smi snapshots (the 15-second pytest form)")
    ap.add_argument("--tag", default="r06")
    ap.add_argument("--worker", type=int, default=None)
    args = ap.parse_args()
    if args.worker is not None:
        worker(args, args.worker)
        return 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    ras0 = {} if args.no_ras else ras_snapshot()
    t0 = time.time()
    base = [sys.executable, os.path.abspath(__file__), "--seconds", str(args.seconds), "--seed0", str(args.seed0), "--seed-end", str(args.seed_end), "--reps", str(args.reps),
            "--rerun", str(args.rerun), "--max-scenes", str(args.max_scenes), "--tag", args.tag] + (["--sync-every-step"] if args.sync_every_step else [])
    procs = [subprocess.Popen(base + ["--worker", str(w)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT) for w in range(args.procs)]
    workers, mismatches, noise = [], [], []
    for p in procs:
        out, _ = p.communicate()
        for line in out.splitlines():
            if line.startswith("WORKER "):
                workers.append(json.loads(line[7:]))
            elif line.startswith("MISMATCH "):
                mismatches.append(json.loads(line[9:]))
            elif line.strip() and "amdgpu.ids" not in line:
                noise.append(line[-300:])
        if p.returncode != 0:
            noise.append(f"worker exit code {p.returncode}")
    el = time.time() - t0
    summ = dict(tag=args.tag, lib=os.environ.get("MMW_LIB_NAME", "libmmw_hip.so"), procs=args.procs, seconds=round(el, 1),
                sync_every_step=bool(args.sync_every_step), reps=args.reps, max_scenes=args.max_scenes,
                cases=sum(w["cases"] for w in workers), case_runs=sum(w["case_runs"] for w in workers),
                frames=sum(w["frames"] for w in workers), mismatches=len(mismatches),
                readback_mismatches=sum(w.get("readback_mismatches", 0) for w in workers), state_mismatches=sum(w.get("state_mismatches", 0) for w in workers),
                reruns=sum(w["reruns"] for w in workers), rerun_mismatches=sum(w["rerun_mismatches"] for w in workers),
                error_cases=sum(w["error_cases"] for w in workers), create_failed=sum(w["create_failed"] for w in workers),
                workers_reporting=len(workers), layouts={}, mismatch_records=mismatches[:40], other_output=noise[:20],
                ras_before=ras0, ras_after={} if args.no_ras else ras_snapshot())
    for w in workers:
        for k, v in w["layouts"].items():
            summ["layouts"][k] = summ["layouts"].get(k, 0) + v
    summ["case_runs_per_s"] = round(summ["case_runs"] / max(el, 1e-9), 1)
    with open(os.path.join(ROOT, "gpurun_out", f"dual_run_{args.tag}.json"), "w") as fh:
        json.dump(summ, fh, indent=1, default=str)
    brief = {k: summ[k] for k in ("tag", "lib", "procs", "seconds", "sync_every_step", "cases", "case_runs", "case_runs_per_s", "frames",
                                   "mismatches", "readback_mismatches", "state_mismatches", "rerun_mismatches", "error_cases", "workers_reporting", "layouts")}
    print(json.dumps(brief))
    for m in mismatches[:5]:
        print("MISMATCH", json.dumps({k: m[k] for k in m if k not in ("cfg",)}, default=str)[:1500])
    for n in noise[:10]:
        print("NOTE", n)
    return 0


if __name__ == "__main__":
    sys.exit(main())
