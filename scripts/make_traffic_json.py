#!/usr/bin/env python3
"""profiles/traffic.json (what bench.py quotes as roofline.traffic) from the PMC summaries of scripts/gpu_round.sh pmc / pmc512:
    python scripts/make_traffic_json.py profiles/r03a_pmc_summary.json profiles/r03a_pmc512_summary.json > profiles/traffic.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_msc_amd._lib import source_hash  # noqa: E402

KEYS = {"k_track": "k_track", "k_scene": "k_track", "k_predict": "k_predict", "k_post": "k_post", "k_dbscan_big": "k_dbscan_big", "k_chain": "k_chain"}


def table(path):
    out = {}
    for name, v in json.load(open(path)).items():
        base = name.split("<")[0]
        if base not in KEYS:
            continue
        ent = {"hbm_bytes_per_launch_raw": v["hbm_bytes_per_launch_raw"], "hbm_bytes_per_launch_fetch_x2": v["hbm_bytes_per_launch_x2"],
               "fetch_raw": v["fetch_bytes_per_launch_raw"], "write": v["write_bytes_per_launch"], "kernel": name}
        out[KEYS[base]] = ent   # (bench.py asks by the profile id's name: k_scene is timed under K_TRACK)
    return out


big, small = sys.argv[1], sys.argv[2]
res = {"4096x512x8": table(big), "512x512x8": table(small),
       # the build the counters were collected on: bench.py quotes them only for a library built from the same sources
       "src_hash": source_hash(),
       "population": "full",   # bench.py's default headline population since round 5 (K = T); quoted only for runs of the same one
       "source": f"{big} / {small}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of bench.py --no-cpu --no-e2e --no-e2e-parity "
                 f"--no-cold --no-shards --no-full --no-ingest --no-single --steps 10 --warmup 10 (--chain-side-stream 2 at 4096 scenes, --scenes 512 for the shard)",
       "_note": "KiB -> bytes; read side given raw and x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads: MI355X_MICROARCH.md). "
                "Counter collection serialises the dispatches: with --chain-side-stream 2 the chain workers are launched all the same (k_chain has its "
                "entry) but, alone on the chip, find nothing to claim and leave after their idle polls -- the DBSCAN bytes they move in the benchmarked "
                "schedule are k_post's / k_dbscan_big's here; the step's total does not depend on who moves them, and k_track, the roofline kernel, "
                "runs the same code with or without their company."}
# (optional third argument: a JSON file {"4096x512x8": {"k_track": N, ...}} of SQ_INSTS_VALU per launch from scripts/pmc_sq.sh sq1 on the
#  same build -- bench.py's roofline_issue; the measured fp64 issue ceiling of the chip travels with it)
if len(sys.argv) > 3:
    res["valu_insts_per_launch"] = json.load(open(sys.argv[3]))
    res["fp64_issue_ceiling"] = {"ns_per_wave_instruction_and_simd": 2.05, "simds": 1024,
                                 "source": "scripts/ubench/ubench_f64 (profiles/r05a_ubench_f64.txt): independent v_fma_f64 chains, 8 waves per SIMD, wall clock"}
print(json.dumps(res, indent=1))
