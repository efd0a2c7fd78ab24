#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs into per-kernel HBM bytes per launch.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB
(hbm_bytes = value * 1024); on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide
coalesced reads, so the read side is reported both raw and x2 ("fetch_x2"); which applies
depends on the access width (our tile loads are 8 B/lane, not the calibrated 16 B/lane), so
both are kept and DESIGN.md says which is quoted.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname, counter):
    out = defaultdict(list)
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                out[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return out


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        short = k.split("::")[-1]
        f = fetch.get(k, [])
        w = write.get(k, [])
        # skip warm-up launches: keep the last half
        f2, w2 = f[len(f) // 2:], w[len(w) // 2:]
        fb = sum(f2) / max(len(f2), 1) * 1024
        wb = sum(w2) / max(len(w2), 1) * 1024
        res[short] = {"launches": len(f), "fetch_bytes_per_launch_raw": fb, "fetch_bytes_per_launch_x2": 2 * fb,
                      "write_bytes_per_launch": wb, "hbm_bytes_per_launch_raw": fb + wb, "hbm_bytes_per_launch_x2": 2 * fb + wb}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
