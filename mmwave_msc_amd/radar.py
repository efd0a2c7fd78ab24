"""Online input step: the IWR1443 UART stream into `normalize_data`'s input (reference src/ReadDataIWR1443.py).

`parse_config_file` restates `ReadIWR14xx.__parseConfigFile` (ReadDataIWR1443.py:203-262: radar .cfg ->
range / doppler scales); `UartFrameParser` is `ReadIWR14xx.read` (27-201) without the serial port: the caller
feeds whatever bytes arrived, the parser keeps the reference's byte buffer discipline (append if it fits, cut to
the LAST magic word, parse one packet, drop it) and returns the same `(dataOK, frameNumber, detObj)` triple.
The packet itself is decoded by the C-ABI's host function `mmw_parse_uart` (include/mmw.h).  Opening and
configuring the serial ports (pyserial, `__serialConfig`) stays with the application.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import _lib

MAX_BUFFER = 2 ** 15          # ReadDataIWR1443.py:24
NUM_TX_ANT = 3                # hard-coded there (214-215)


def parse_config_file(path: str) -> dict:
    """numDopplerBins, numRangeBins, rangeResolutionMeters, rangeIdxToMeters, dopplerResolutionMps, maxRange,
    maxVelocity (+ framePeriodicity) from the `profileCfg` and `frameCfg` lines of a radar .cfg."""
    start_freq = idle = ramp_end = slope = n_adc = rate = None
    chirp0 = chirp1 = loops = period = None
    with open(path) as fh:
        for line in fh:
            w = line.rstrip("\r\n").split(" ")
            if "profileCfg" in w[0]:
                start_freq, idle, ramp_end = int(float(w[2])), int(w[3]), float(w[5])
                slope, n_adc, rate = float(w[8]), int(w[10]), int(w[11])
            elif "frameCfg" in w[0]:
                chirp0, chirp1, loops, period = int(w[1]), int(w[2]), int(w[3]), float(w[5])
    pow2 = 1
    while n_adc > pow2:
        pow2 *= 2
    p = {}
    chirps = (chirp1 - chirp0 + 1) * loops
    p["numDopplerBins"] = chirps / NUM_TX_ANT
    p["numRangeBins"] = pow2
    p["rangeResolutionMeters"] = (3e8 * rate * 1e3) / (2 * slope * 1e12 * n_adc)
    p["rangeIdxToMeters"] = (3e8 * rate * 1e3) / (2 * slope * 1e12 * p["numRangeBins"])
    p["dopplerResolutionMps"] = 3e8 / (2 * start_freq * 1e9 * (idle + ramp_end) * 1e-6 * p["numDopplerBins"] * NUM_TX_ANT)
    p["maxRange"] = (300 * 0.9 * rate) / (2 * slope * 1e3)
    p["maxVelocity"] = 3e8 / (4 * start_freq * 1e9 * (idle + ramp_end) * 1e-6 * NUM_TX_ANT)
    p["framePeriodicity"] = period
    return p


_UartCfg = _lib.MmwUartCfg


def uart_cfg(config_parameters: dict) -> "_lib.MmwUartCfg":
    """struct mmw_uart_cfg from the reference's configParameters dict (rangeIdxToMeters, dopplerResolutionMps, numDopplerBins)."""
    return _lib.MmwUartCfg(float(config_parameters["rangeIdxToMeters"]), float(config_parameters["dopplerResolutionMps"]),
                           int(config_parameters["numDopplerBins"]), 0)


def find_tlv(buf) -> tuple:
    """mmw_find_tlv on a bytes-like object: (found, body_offset, n_obj, frame_number, packet_start, packet_len) -- the packet
    part of ReadIWR14xx.read (ReadDataIWR1443.py:47-113) without decoding an object: what the host does per packet when the
    GPU decodes the detected-points TLV itself (SceneBatch.normalize_tlv_dev)."""
    a = np.frombuffer(buf, dtype=np.uint8)
    off, n = C.c_int64(-1), C.c_int32(0)
    frame = C.c_uint32(0)
    start, plen = C.c_size_t(0), C.c_size_t(0)
    rc = _lib.load().mmw_find_tlv(a.ctypes.data, len(a), C.byref(off), C.byref(n), C.byref(frame), C.byref(start), C.byref(plen))
    if rc < 0:
        raise _lib.MmwError(rc, "mmw_find_tlv: bad arguments")
    return rc == 1, int(off.value), int(n.value), int(frame.value), int(start.value), int(plen.value)


def encode_tlv_bodies(raw: np.ndarray, counts: np.ndarray, qfmt: int, doppler_resolution_mps: float, stride: int = 0) -> np.ndarray:
    """A synthetic sensor: the detected-points TLV BODIES an IWR1443 would have sent for raw rows (x, y, z, doppler, peakVal) --
    u16 numObj, u16 xyzQFormat, then per object int16 rangeIdx (0), dopplerIdx = round(doppler / resolution), peakVal, and
    x, y, z = round(coordinate * 2^Q) (ReadDataIWR1443.py:107-150) -- one body per scene at a fixed stride (default: the smallest
    multiple of 16 that holds max_pts objects).  raw[..., N, 5], counts[...] -> uint8 [..., stride].  Used by the tests and by
    bench_ingest.py's `tlv` leg; decoding them (mmw_parse_uart / mmw_normalize_tlv) gives the QUANTISED rows, not `raw`."""
    raw = np.asarray(raw)
    lead, N = raw.shape[:-2], raw.shape[-2]
    if stride <= 0:
        stride = (4 + 12 * N + 15) // 16 * 16
    assert stride >= 4 + 12 * N and stride % 2 == 0
    out = np.zeros(lead + (stride,), dtype=np.uint8)
    words = out.view(np.uint16).reshape(lead + (stride // 2,))
    cnt = np.clip(np.asarray(counts), 0, N).astype(np.uint16)
    words[..., 0] = cnt
    words[..., 1] = qfmt
    obj = np.zeros(lead + (N, 6), dtype=np.int16)
    obj[..., 1] = np.clip(np.rint(raw[..., 3].astype(np.float64) / doppler_resolution_mps), -32768, 32767).astype(np.int16)
    obj[..., 2] = np.clip(np.rint(raw[..., 4].astype(np.float64)), -32768, 32767).astype(np.int16)
    obj[..., 3:6] = np.clip(np.rint(raw[..., 0:3].astype(np.float64) * float(2 ** qfmt)), -32768, 32767).astype(np.int16)
    valid = np.arange(N).reshape((1,) * len(lead) + (N,)) < cnt[..., None]
    obj[~valid] = 0
    words[..., 2: 2 + 6 * N] = obj.view(np.uint16).reshape(lead + (6 * N,))
    return out


def decode_tlv_bodies_numpy(bodies: np.ndarray, cfg: dict):
    """The reference's decode (ReadDataIWR1443.py:153-171, numpy-1.26 int16 wrap) of `encode_tlv_bodies`-shaped bodies, in numpy:
    (raw[..., N, 5] float64, counts[...]).  The checker's restatement -- the product decodes on the device."""
    lead, stride = bodies.shape[:-1], bodies.shape[-1]
    words = np.ascontiguousarray(bodies).view(np.uint16).reshape(lead + (stride // 2,))
    cnt = words[..., 0].astype(np.int32)
    q = np.ldexp(1.0, words[..., 1].astype(np.int32))
    N = (stride - 4) // 12
    obj = words[..., 2: 2 + 6 * N].reshape(lead + (N, 6)).view(np.int16)
    dop = obj[..., 1].copy()
    hi = dop > (cfg["numDopplerBins"] / 2 - 1)
    dop[hi] = (dop[hi].astype(np.int32) - 65535).astype(np.int16)
    raw = np.zeros(lead + (N, 5))
    raw[..., 0:3] = obj[..., 3:6] / q[..., None, None]
    raw[..., 3] = dop * cfg["dopplerResolutionMps"]
    raw[..., 4] = obj[..., 2]
    return raw, cnt


class UartFrameParser:
    def __init__(self, config_parameters: dict, max_obj: int = 1024):
        self.configParameters = dict(config_parameters)
        self.byteBuffer = np.zeros(MAX_BUFFER, dtype=np.uint8)
        self.byteBufferLength = 0
        self.max_obj = int(max_obj)
        self._cfg = _UartCfg(float(config_parameters["rangeIdxToMeters"]), float(config_parameters["dopplerResolutionMps"]),
                             int(config_parameters["numDopplerBins"]), 0)
        self._raw = np.zeros((self.max_obj, 5))
        self._rng = np.zeros(self.max_obj)

    def feed(self, data: bytes):
        """One call of `read()` with `data` as what the port delivered: (dataOK, frameNumber, detObj)."""
        L = _lib.load()
        vec = np.frombuffer(data, dtype=np.uint8)
        if self.byteBufferLength + len(vec) < MAX_BUFFER:       # (a chunk that does not fit is dropped, as there)
            self.byteBuffer[self.byteBufferLength: self.byteBufferLength + len(vec)] = vec
            self.byteBufferLength += len(vec)
        if self.byteBufferLength <= 16:
            return 0, 0, {}
        n = C.c_int32(0)
        frame = C.c_uint32(0)
        start, plen = C.c_size_t(0), C.c_size_t(0)
        rc = L.mmw_parse_uart(self.byteBuffer.ctypes.data, self.byteBufferLength, C.byref(self._cfg), self._raw.ctypes.data,
                              self._rng.ctypes.data, self.max_obj, C.byref(n), C.byref(frame), C.byref(start), C.byref(plen))
        if rc < 0:
            raise _lib.MmwError(rc, "mmw_parse_uart: more objects than max_obj or bad arguments")
        if start.value > 0:                                     # cut to the last magic word
            rest = self.byteBufferLength - start.value
            self.byteBuffer[:rest] = self.byteBuffer[start.value: self.byteBufferLength].copy()
            self.byteBufferLength = rest
        if plen.value == 0:                                     # no magic word, or the packet is not complete yet
            return 0, 0, {}
        det, ok, idx = {}, 0, 36
        if self._num_detected() > 0:
            idx = 44                                            # TLV type and length were read
            if rc == 1:
                k = n.value
                r = self._raw[:k]
                det = {"numObj": k, "range": self._rng[:k].copy(), "doppler": r[:, 3].copy(), "peakVal": r[:, 4].astype(np.int16),
                       "x": r[:, 0].copy(), "y": r[:, 1].copy(), "z": r[:, 2].copy(), "timestamp": round(time.time() * 1000)}
                ok, idx = 1, 48 + 12 * k
        if self.byteBufferLength > idx:                         # "remove already processed data" (191-197)
            total = plen.value
            rest = self.byteBufferLength - total
            self.byteBuffer[:rest] = self.byteBuffer[total: self.byteBufferLength].copy()
            self.byteBufferLength = rest
        return ok, int(frame.value), det

    def _num_detected(self) -> int:
        b = self.byteBuffer
        return int(b[28]) | int(b[29]) << 8 | int(b[30]) << 16 | int(b[31]) << 24
