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


def _tlv_case(rng, S, N, cfgp):
    """S chunks as a serial port might deliver them: most hold one complete detected-points packet (random object count, Q format,
    int16 fields incl. negative coordinates and doppler indices on both sides of the reference's wrap threshold), some none."""
    chunks, kinds = [], []
    for s in range(S):
        kind = int(rng.integers(0, 10))
        n = int(rng.integers(0, N + 1)) if kind != 3 else N
        o = np.zeros((n, 6), dtype=np.int64)
        o[:, 0] = rng.integers(0, 256, n)
        o[:, 1] = rng.integers(-40, 41, n)
        o[:, 2] = rng.integers(0, 4000, n)
        o[:, 3] = rng.integers(-1500, 1500, n)
        o[:, 4] = rng.integers(20, 3600, n)
        o[:, 5] = rng.integers(-900, 300, n)
        q = int(rng.choice([7, 8, 9, 9, 9]))
        lead = b"\x00" * int(2 * rng.integers(0, 4))            # (bodies 2-byte aligned, not always 4)
        if kind == 0:
            pkt = _packet(7 + s, o, qfmt=q, num_det=0)            # header announces no objects
        elif kind == 1:
            pkt = _packet(7 + s, o, qfmt=q, tlv_type=2)           # another TLV first
        elif kind == 2:
            pkt = _packet(7 + s, o, qfmt=q)[:-6] if n else _packet(7 + s, o, qfmt=q)   # the last object is not all there yet
        else:
            pkt = _packet(7 + s, o, qfmt=q)
        chunks.append(lead + pkt + (b"\x00\x00" if kind == 4 else b""))
        kinds.append(kind)
    return chunks, kinds


@pytest.mark.parametrize("N", [64, 200, 600])
def test_device_tlv_decode_and_normalize_equals_host_parse_then_normalize(N):
    """mmw_find_tlv + mmw_normalize_tlv (the GPU decodes the detected-points TLV of every scene's packet and normalises it in one
    kernel; ReadDataIWR1443.py:107-171, Utils.py:342-434) against mmw_parse_uart + mmw_normalize on the same bytes: the rows that
    reach track() and their counts bit-equal, scene by scene; scenes without a complete detected-points packet get n = 0.  Then both
    paths through mmw_step for three frames: identical association and track state."""
    import ctypes as C
    from mmwave_msc_amd import _lib, radar
    from mmwave_msc_amd.batch import SceneBatch
    rng = np.random.default_rng(40 + N)
    S = 48
    cfgp = {"rangeIdxToMeters": 0.0436, "dopplerResolutionMps": 0.1252, "numDopplerBins": 32.0}
    ucfg = radar.uart_cfg(cfgp)
    sb_dev = SceneBatch(_lib.default_config(db_min_samples=10), S, N)
    sb_host = SceneBatch(_lib.default_config(db_min_samples=10), S, N)
    L = _lib.load()
    n_rows = 0
    for frame in range(3):
        chunks, kinds = _tlv_case(rng, S, N, cfgp)
        # ---- host path: one mmw_parse_uart per packet, then mmw_normalize ----
        raw = np.zeros((S, N, 5))
        n_raw = np.zeros(S, np.int32)
        rcs = np.zeros(S, np.int32)
        for s, ch in enumerate(chunks):
            a = np.frombuffer(ch, dtype=np.uint8)
            rows = np.zeros((N, 5))
            n = C.c_int32(0)
            rc = L.mmw_parse_uart(a.ctypes.data, len(a), C.byref(ucfg), rows.ctypes.data, None, N, C.byref(n), None, None, None)
            assert rc in (0, 1), (s, rc)
            raw[s], n_raw[s], rcs[s] = rows, n.value, rc
        want_pts, want_n = sb_host.normalize_host(raw, n_raw)
        # ---- device path: the host only finds the bodies ----
        blob = b"".join(chunks)
        offs = np.full(S, -1, np.int64)
        base = 0
        for s, ch in enumerate(chunks):
            found, off, n_obj, _, _, _ = radar.find_tlv(ch)
            assert found == (rcs[s] == 1) and (not found or n_obj == n_raw[s]), (s, kinds[s], found, rcs[s])
            if found:
                assert n_obj <= N
                offs[s] = base + off
            base += len(ch)
        b_pk = sb_dev.buf("tlv_bytes", len(blob) + 16).upload(np.frombuffer(blob, dtype=np.uint8))
        b_of = sb_dev.buf("tlv_off", S * 8).upload(offs)
        b_out = sb_dev.buf("tlv_pts", S * N * 64)
        b_no = sb_dev.buf("tlv_n", S * 4)
        sb_dev.normalize_tlv_dev(b_pk.ptr, len(blob), b_of.ptr, ucfg, b_out.ptr, b_no.ptr)
        got_n = b_no.download((S,), np.int32)
        got_pts = b_out.download((S, N, 8), np.float64)
        assert np.array_equal(got_n, want_n), (frame, got_n, want_n)
        for s in range(S):
            assert np.array_equal(got_pts[s, : got_n[s]], want_pts[s, : want_n[s]]), (frame, s, kinds[s])
        n_rows += int(got_n.sum())
        # ---- both through TrackBuffer.track ----
        dt = np.full(S, 0.1)
        b_dt = sb_dev.buf("tlv_dt", S * 8).upload(dt)
        b_as = sb_dev.buf("tlv_assoc", S * N * 4)
        sb_dev.step_dev(b_out.ptr, b_no.ptr, b_dt.ptr, b_as.ptr)
        a_host, _, _ = sb_host.step_host(want_pts, want_n, dt)
        a_dev = b_as.download((S, N), np.int32)
        for s in range(S):
            assert np.array_equal(a_dev[s, : got_n[s]], a_host[s, : want_n[s]]), (frame, s)
    assert n_rows > S * N // 8
    nt_d, nt_h = sb_dev.num_tracks(), sb_host.num_tracks()
    assert np.array_equal(nt_d, nt_h)
    td, th = sb_dev.tracks(cap=max(int(nt_d.max()), 1)), sb_host.tracks(cap=max(int(nt_h.max()), 1))
    for name in ("x", "P", "centroid", "spread_est", "group_disp_est", "lifetime", "point_num", "ring_n"):
        assert np.array_equal(td[name], th[name]), name
    sb_dev.check(); sb_host.check()
    sb_dev.close(); sb_host.close()


def test_find_tlv_agrees_with_the_recorded_uart_session():
    """The byte chunks of the reference's recorded read() session (tests/golden/uart.npz).  The recording holds no DECODED packet
    -- under numpy 2 the reference's decode branch raises, so the generator could only record the other paths (no magic word,
    incomplete packet, no objects announced, another TLV first) --: on every buffer state of the session mmw_find_tlv reports the
    frame number the reference read from the header and no detected-points body, as mmw_parse_uart does."""
    from mmwave_msc_amd import radar
    g = np.load(GOLD, allow_pickle=True)
    cfg = g["cfg"]
    cfgp = {"rangeIdxToMeters": float(cfg[0]), "dopplerResolutionMps": float(cfg[1]), "numDopplerBins": float(cfg[2])}
    p = radar.UartFrameParser(cfgp)
    complete = 0
    for i in range(int(g["n_chunks"])):
        chunk = g[f"chunk{i}"].tobytes()
        before = (bytes(p.byteBuffer[: p.byteBufferLength]) + chunk) if p.byteBufferLength + len(chunk) < radar.MAX_BUFFER else bytes(p.byteBuffer[: p.byteBufferLength])
        ok, fn, det = p.feed(chunk)
        found, off, n_obj, frame, start, plen = radar.find_tlv(before)
        assert ok == int(g[f"ok{i}"]) == 0 and not found and off == -1 and n_obj == 0, i
        assert frame == fn == int(g[f"frame{i}"]), (i, frame, fn)
        complete += int(plen > 0)
    assert complete >= 3


def test_device_tlv_decode_refuses_bodies_outside_the_buffer_or_the_context():
    """mmw_normalize_tlv reads nothing outside packets[0 .. packets_bytes): an offset that is odd or beyond the buffer, a body
    whose announced objects run past its end, and a body that announces more than max_pts objects (mmw_parse_uart: MMW_E_ARG for
    those bytes) all give n_out = MMW_BAD_FRAME for THAT scene -- the mmw_step that follows raises its bad-count bit -- while the
    scenes beside it decode as usual."""
    import struct
    from mmwave_msc_amd import _lib, radar
    from mmwave_msc_amd.batch import SceneBatch
    N, S = 64, 6
    cfgp = {"rangeIdxToMeters": 0.0436, "dopplerResolutionMps": 0.1252, "numDopplerBins": 32.0}
    ucfg = radar.uart_cfg(cfgp)
    rng = np.random.default_rng(3)

    def body(n_announced, n_present, q=9):
        o = rng.integers(-200, 200, size=(n_present, 6)).astype("<i2")
        o[:, 4] = np.abs(o[:, 4]) + 300      # y > 0: the scene filter keeps rows
        return struct.pack("<HH", n_announced, q) + o.tobytes()

    good = body(20, 20)
    blob = good + body(N + 1, N + 1) + good + body(30, 30)      # scene 1 announces max_pts + 1 objects; the last body gets truncated below
    o1, o2, o3 = len(good), len(good) + 4 + 12 * (N + 1), 2 * len(good) + 4 + 12 * (N + 1)
    nbytes = len(blob) - 12 * 5                                   # scene 3: five of its thirty objects lie outside the buffer
    offs = np.array([0, o1, o2, o3, o2 + 1, nbytes - 2], np.int64)   # scene 4: odd offset; scene 5: the 4-byte head does not fit
    sb = SceneBatch(_lib.default_config(), S, N)
    b_pk = sb.buf("tlv_bytes", len(blob) + 16).upload(np.frombuffer(blob, dtype=np.uint8))
    b_of = sb.buf("tlv_off", S * 8).upload(offs)
    b_out = sb.buf("tlv_pts", S * N * 64)
    b_no = sb.buf("tlv_n", S * 4)
    sb.normalize_tlv_dev(b_pk.ptr, nbytes, b_of.ptr, ucfg, b_out.ptr, b_no.ptr)
    got = b_no.download((S,), np.int32)
    assert got[0] > 0 and got[2] == got[0], got
    assert list(got[[1, 3, 4, 5]]) == [_lib.BAD_FRAME] * 4, got
    b_dt = sb.buf("tlv_dt", S * 8).upload(np.full(S, 0.1))
    sb.step_dev(b_out.ptr, b_no.ptr, b_dt.ptr)
    with pytest.raises(_lib.MmwError) as ei:
        sb.check()
    assert ei.value.code == _lib.E_ARG
    err = sb.errors()
    assert [bool(e & 8) for e in err] == [False, True, False, True, True, True], err
    sb.close()
