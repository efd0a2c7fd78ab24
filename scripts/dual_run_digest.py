"""Digest of the dual-run windows under gpurun_out/ (scripts/dual_run.py) for profiles/: one record per window -- build, load,
case-runs, mismatches by kind with their first-difference records (configuration dropped), worker crashes, ECC totals."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = []
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "dual_run_r06*.json"))):
    d = json.load(open(f))

    def ecc(snap):
        try:
            return json.loads(snap["amd-smi metric --ecc"])["gpu_data"][0]["ecc"]
        except Exception:   # noqa: BLE001
            return None
    rec = {k: d[k] for k in ("tag", "lib", "procs", "seconds", "sync_every_step", "reps", "max_scenes", "cases", "case_runs", "case_runs_per_s", "frames",
                             "mismatches", "rerun_mismatches", "reruns", "error_cases", "workers_reporting", "layouts") if k in d}
    rec["readback_mismatches"], rec["state_mismatches"] = d.get("readback_mismatches"), d.get("state_mismatches")
    rec["worker_crashes"] = [n for n in d["other_output"] if "Memory access fault" in n or "exit code" in n]
    rec["mismatch_records"] = [{k: v for k, v in m.items() if k not in ("cfg", "queue_a", "queue_b")} | {"queue_a": m.get("queue_a", [])[:12], "queue_b": m.get("queue_b", [])[:12]}
                               for m in d["mismatch_records"]]
    rec["ecc_before"], rec["ecc_after"] = ecc(d["ras_before"]), ecc(d["ras_after"])
    out.append(rec)
json.dump(out, open(os.path.join(ROOT, "profiles", "r06_dual_run_windows.json"), "w"), indent=1, default=str)
for r in out:
    print(r["tag"], r["lib"], "procs", r["procs"], "case-runs", r["case_runs"], "mismatches", r["mismatches"], "crashes", len(r["worker_crashes"]) // 2, "ecc", r["ecc_after"])
