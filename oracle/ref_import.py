"""Import the read-only reference (`/root/reference/src`) in THIS container.

TEST INFRASTRUCTURE ONLY.  Used by `oracle/gen_golden.py` and by the
container-only pinning tests; never available on the GPU box
(`have_reference()` is False there and everything depending on it skips).

The reference imports `filterpy` (absent here) -> `oracle/filterpy_shim` is put
on sys.path first.  Modules that need PyQt5/keras/wakepy/pyserial
(offline_main, main, Visualizer, train, preprocessing) cannot be imported; the
ones on the hot path (constants, Utils, Tracking) can.
"""
import importlib
import os
import sys

REFERENCE_SRC = "/root/reference/src"
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "filterpy_shim")


def have_reference() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_SRC, "Tracking.py"))


def load_reference():
    """Returns (constants, Utils, Tracking) modules of the reference."""
    if not have_reference():
        raise RuntimeError("reference sources are not present on this machine")
    sys.dont_write_bytecode = True  # /root/reference is read-only
    for p in (_SHIM, REFERENCE_SRC):
        if p not in sys.path:
            sys.path.insert(0, p)
    const = importlib.import_module("constants")
    utils = importlib.import_module("Utils")
    tracking = importlib.import_module("Tracking")
    assert os.path.dirname(os.path.abspath(const.__file__)) == REFERENCE_SRC
    return const, utils, tracking
