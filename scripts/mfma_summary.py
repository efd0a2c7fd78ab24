#!/usr/bin/env python3
"""Per-kernel MFMA counters of the posture leg (rocprofv3 --pmc pass) -> JSON summary."""
import csv, glob, json, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][-70:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
out = {}
for k, d in acc.items():
    mops = d.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0) + d.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0)
    if mops <= 0:
        continue
    # MFMA_BUSY counts cycles per SIMD summed over SIMDs; GUI_ACTIVE is summed over the 8 XCDs
    gui = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    out[k] = {"dispatches": cnt[k], "mfma_mops_f16": d.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0), "mfma_mops_f32": d.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0), "mfma_busy_cycles": busy,
              "gpu_active_cycles": gui, "mfma_util_vs_1024_simds": (busy / (gui * 1024.0)) if gui else None}
print(json.dumps(out, indent=1))
