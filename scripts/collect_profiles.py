#!/usr/bin/env python3
"""After `scripts/gpu_round.sh tests smoke prof posture pmc pmc512 [bench ...]` + `scripts/pmc_sq.sh sq1 | grep ^sq1 > gpurun_out/sq1_lines.txt`
on the GPU box: copy the summaries under profiles/<tag>_* and rebuild profiles/traffic.json for the library's source hash.
    python scripts/collect_profiles.py r06j"""
import glob, json, os, re, shutil, subprocess, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g, p = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def newest(pattern):
    files = sorted(glob.glob(os.path.join(g, pattern), recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def cp(src, name):
    if src and os.path.exists(src):
        shutil.copy(src, os.path.join(p, f"{tag}_{name}"))
        print("copied", name)


cp(os.path.join(g, "pmc_summary.json"), "pmc_summary.json")
cp(os.path.join(g, "pmc512_summary.json"), "pmc512_summary.json")
cp(os.path.join(g, "sq1_lines.txt"), "pmc_sq1_counters.txt")
cp(newest("prof/**/*kernel_stats.csv"), "kernel_stats.csv")
cp(newest("prof_posture/**/*kernel_stats.csv"), "posture_kernel_stats.csv")
cp(os.path.join(g, "mfma_summary.json"), "posture_mfma_counters.json")
cp(os.path.join(g, "pytest_gpu.txt"), "pytest_gpu.txt")
cp(os.path.join(g, "smoke.txt"), "smoke.txt")
for name in ("bench_default.json", "bench_256.json", "bench_160frames.json"):
    src = os.path.join(g, name)
    if os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(os.path.join(g, "pmc_summary.json")) - 3600:
        cp(src, name)
vals = {}
for line in open(os.path.join(p, f"{tag}_pmc_sq1_counters.txt")):
    m = re.match(r"sq1 (k_\w+)(<[^>]*>)? (\{.*\}) lau", line)
    if m:
        vals[m.group(1)] = eval(m.group(3))["SQ_INSTS_VALU"]
valu = {"4096x512x8": {k: vals[k] for k in ("k_predict", "k_track", "k_post")},
        "source": f"scripts/pmc_sq.sh sq1 (rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU ..., its own pass) on this build (profiles/{tag}_pmc_sq1_counters.txt), "
                  "K = T population: wave-level vector instructions per launch"}
json.dump(valu, open("/tmp/valu.json", "w"))
out = subprocess.run([sys.executable, os.path.join(root, "scripts", "make_traffic_json.py"), os.path.join(p, f"{tag}_pmc_summary.json"),
                      os.path.join(p, f"{tag}_pmc512_summary.json"), "/tmp/valu.json"], capture_output=True, text=True, check=True).stdout
open(os.path.join(p, "traffic.json"), "w").write(out)
print("traffic.json for", json.loads(out)["src_hash"])
