import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    from oracle.ref_import import have_reference

    if not have_reference():
        skip = pytest.mark.skip(reason="/root/reference not present on this machine")
        for it in items:
            if "reference" in it.keywords:
                it.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Which kernels the GPU parity tests really ran under (tests/_layouts.py): written next to the other GPU-box outputs
    and echoed at the end of the log."""
    import json
    mod = sys.modules.get("tests._layouts")
    if mod is None or not any(mod.REPORT["contexts"].values()):
        return
    rep = dict(mod.REPORT)
    reasons = {}
    for layout, why in rep["skipped"]:
        reasons[f"{layout}: {why}"] = reasons.get(f"{layout}: {why}", 0) + 1
    rep["skipped"] = reasons
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "layout_report.json"), "w") as fh:
            json.dump(rep, fh, indent=1, sort_keys=True)
    except OSError:
        pass
    tr = session.config.pluginmanager.get_plugin("terminalreporter")
    if tr is not None:
        tr.write_line("")
        tr.write_line("kernel layouts the parity tests ran under: " + json.dumps(rep, sort_keys=True))
