"""The GPU-vs-GPU determinism harness (scripts/dual_run.py) as a short test: two contexts of one configuration in this process,
each on its own stream, the same device-resident frames, twelve steps queued without a host wait -- per-frame association,
labels, db_n and track tables and the final state must be equal bit for bit.  A few seconds here guard the harness and the
property; what it is FOR -- the two races it caught under six processes sharing the GPU, one fresh context in 10^4: null-stream
initialisation in mmw_create, a side-stream worker claiming another step's cloud -- needs `python scripts/dual_run.py --procs 6`
for minutes (profiles/r06_dual_run_windows.json: 664 k case-runs clean on the fixed build, two boxes)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_contexts_of_one_configuration_stay_equal_bit_for_bit():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dual_run.py"), "--seconds", "12", "--procs", "2", "--tag", "pytest",
                          "--seed0", "77000000", "--reps", "4", "--no-ras"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    brief = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert brief["workers_reporting"] == 2, out.stdout[-2000:]                  # (no worker died: a GPU memory fault kills the process)
    assert brief["mismatches"] == 0, out.stdout[-3000:]
    assert brief["case_runs"] >= 200 and len(brief["layouts"]) >= 4, brief     # every layout took part
