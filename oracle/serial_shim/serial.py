"""Stand-in for pyserial, used ONLY by oracle/gen_golden.py to import the reference's ReadDataIWR1443 module
(test infrastructure; pyserial is not installed here).  The golden generator never opens a port: it bypasses
ReadIWR14xx.__init__ and feeds byte chunks through a fake Dataport object."""


class Serial:  # pragma: no cover - never instantiated
    def __init__(self, *a, **k):
        raise RuntimeError("oracle/serial_shim: no serial ports in the test environment")
