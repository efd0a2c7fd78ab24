"""mmw_parse_uart / radar.UartFrameParser (reference src/ReadDataIWR1443.py:27-262).  The parser is host code
inside the HIP library, so loading it needs the GPU box (marked gpu); no kernel runs."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "uart.npz")


def _packet(frame, objs, qfmt=9, tlv_type=1, num_det=None):
    body = struct.pack("<HH", len(objs), qfmt) + b"".join(struct.pack("<6H", *[int(v) & 0xFFFF for v in o]) for o in objs)
    tlv = struct.pack("<II", tlv_type, len(body)) + body
    total = 36 + len(tlv)
    return bytes([2, 1, 4, 3, 6, 5, 8, 7]) + struct.pack("<IIIIIII", 0x01020304, total, 0xA1443, frame, 1, len(objs) if num_det is None else num_det, 1) + tlv


def test_buffer_discipline_matches_reference_recording():
    """Frame numbers, dataOK and the length of the byte buffer after every read, as recorded from the reference's
    read() (oracle/gen_golden.py gen_uart: the paths that run under numpy 2)."""
    from mmwave_msc_amd.radar import UartFrameParser
    g = np.load(GOLD, allow_pickle=True)
    cfg = g["cfg"]
    p = UartFrameParser({"rangeIdxToMeters": cfg[0], "dopplerResolutionMps": cfg[1], "numDopplerBins": cfg[2]})
    for i in range(int(g["n_chunks"])):
        ok, fn, det = p.feed(g[f"chunk{i}"].tobytes())
        assert ok == int(g[f"ok{i}"]) and fn == int(g[f"frame{i}"]), (i, ok, fn)
        assert p.byteBufferLength == int(g[f"buflen{i}"]), (i, p.byteBufferLength, int(g[f"buflen{i}"]))


def test_detected_points_decode():
    """The decode branch against a restatement of ReadDataIWR1443.py:118-175 with numpy-1.26 semantics (u16 words
    stored into int16 arrays wrap; indices above numDopplerBins/2 - 1 get 65535 subtracted, again in int16).
    Parity with the reference itself is UNPINNED here: under numpy 2 that branch raises OverflowError."""
    from mmwave_msc_amd.radar import UartFrameParser
    rng = np.random.default_rng(3)
    cfgp = {"rangeIdxToMeters": 0.0436, "dopplerResolutionMps": 0.1252, "numDopplerBins": 16.0}
    p = UartFrameParser(cfgp)
    for frame, (n, q) in enumerate([(5, 9), (1, 7), (64, 9), (200, 8)]):
        o = np.zeros((n, 6), dtype=np.int64)
        o[:, 0] = rng.integers(0, 256, n)
        o[:, 1] = rng.integers(-8, 16, n)
        o[:, 2] = rng.integers(0, 4000, n)
        o[:, 3:6] = rng.integers(-3000, 3000, size=(n, 3))
        ok, fn, det = p.feed(b"xx" + _packet(100 + frame, o, qfmt=q))
        assert ok == 1 and fn == 100 + frame and det["numObj"] == n
        dop = o[:, 1].astype(np.int16)
        hi = dop > (cfgp["numDopplerBins"] / 2 - 1)
        dop[hi] = (dop[hi].astype(np.int32) - 65535).astype(np.int16)
        assert np.array_equal(det["doppler"], dop * cfgp["dopplerResolutionMps"])
        for k, col in (("x", 3), ("y", 4), ("z", 5)):
            assert np.array_equal(det[k], o[:, col].astype(np.int16) / 2 ** q), k
        assert np.array_equal(det["peakVal"], o[:, 2].astype(np.int16))
        assert np.array_equal(det["range"], o[:, 0].astype(np.int16) * cfgp["rangeIdxToMeters"])
        # (the reference only drops a packet when MORE bytes than it parsed are buffered, ReadDataIWR1443.py:191:
        #  a packet that ends exactly at the end of the buffer stays until the next read cuts to a later magic word)
        assert p.byteBufferLength == 48 + 12 * n


def test_parse_config_file(tmp_path):
    from mmwave_msc_amd.radar import parse_config_file
    cfg = tmp_path / "radar.cfg"
    cfg.write_text("sensorStop\nprofileCfg 0 77 7 7 58 0 0 68 1 256 5500 0 0 30\nframeCfg 0 2 16 0 100 1 0\nsensorStart\n")
    p = parse_config_file(str(cfg))
    assert p["numDopplerBins"] == 16.0 and p["numRangeBins"] == 256 and p["framePeriodicity"] == 100.0
    assert abs(p["rangeIdxToMeters"] - (3e8 * 5500 * 1e3) / (2 * 68 * 1e12 * 256)) < 1e-15
    assert abs(p["dopplerResolutionMps"] - 3e8 / (2 * 77 * 1e9 * (7 + 58) * 1e-6 * 16 * 3)) < 1e-15
