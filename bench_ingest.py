"""bench.py's `ingest` leg: the HOST-FED step -- what the reference's loop does every frame (offline_main.py:40-57: rows from
the host, Utils.normalize_data, TrackBuffer.track) -- instead of frames that are resident in HBM when the clock starts.

Per step, on two streams and three device buffers:

    copy stream     H2D(frame f+1) ....................  H2D(frame f+2) ......
    tracker stream  wait H2D(f) | [normalize] step(f) |  wait H2D(f+1) | ...

from pinned host memory (a feeder writes the radar's rows there; nothing is staged through pageable memory), in four forms:

    rows_f32   normalised rows as fp32, 32 B per point  -> mmw_step_f32 (promoted in registers)        the product's host-fed path
    rows_f64   the same rows as fp64, 64 B per point    -> mmw_step          what a caller holding the reference's float64 arrays sends
    raw_f32    raw radar rows (x, y, z, doppler, peakVal) as fp32, 20 B per object -> mmw_normalize_f32 -> mmw_step
    tlv        the radar's own wire format: the detected-points TLV body of every scene's UART packet (u16 count, u16 Q, six int16
               per object: 12 B per object, ReadDataIWR1443.py:107-150) -> mmw_normalize_tlv (decode + normalize_data in one kernel)
               -> mmw_step; the host only finds the packets
    e2e_rows_f32   rows_f32 + features + MARS CNN + keypoints every frame (bench_e2e.e2e_leg with this step)

`value` of bench.py's headline stays the HBM-resident rate; these are the sustainable ones, each with the PCIe rate it moved and
which resource bounds it.  The oracle is used here only as the checker of the raw-row form (first scenes, final state)."""
import time

import numpy as np


def raw_rows_from_normalised(pts, tilt_cos, tilt_sin, s_height):
    """Raw radar rows (x, y, z, doppler, peakVal) for the synthetic scene: the inverse of point_transform_to_standard_axis
    (Utils.py:294-339) on the position, and as doppler what a radar measures of the row's velocity -- its RADIAL component
    (v . p) / |p| in the sensor's frame -- so that normalize_data (Utils.py:380-398: v = doppler * p / |p|) gives the tracker
    velocity columns that are consistent with the targets' motion, as for a real sensor.  float32 [..., 5]."""
    p64 = pts.astype(np.float64)
    y1, z1 = p64[..., 1], p64[..., 2] - s_height
    x, y, z = p64[..., 0], tilt_cos * y1 + tilt_sin * z1, -tilt_sin * y1 + tilt_cos * z1
    vy1, vz1 = p64[..., 4], p64[..., 5]
    vx, vy, vz = p64[..., 3], tilt_cos * vy1 + tilt_sin * vz1, -tilt_sin * vy1 + tilt_cos * vz1
    r = np.sqrt(x * x + y * y + z * z)
    raw = np.empty(pts.shape[:-1] + (5,), dtype=np.float32)
    raw[..., 0], raw[..., 1], raw[..., 2] = x, y, z
    raw[..., 3] = np.where(r > 0, (vx * x + vy * y + vz * z) / np.maximum(r, 1e-30), 0.0)
    raw[..., 4] = pts[..., 7]
    return raw


class HostFeed:
    """Frames from pinned host memory through `nbuf` device buffers, the copy of frame f+1 in flight while frame f is tracked."""

    def __init__(self, sb, host_frames, form, d_cnt, d_dt, outs, dev, compute_stream, nbuf=3, max_pts=None, uart_cfg=None):
        import torch
        self.torch, self.sb, self.form, self.dev, self.cs = torch, sb, form, dev, compute_stream
        self.host = torch.from_numpy(host_frames).pin_memory()          # [F, S, N, C]
        self.F = self.host.shape[0]
        self.bytes_per_frame = int(self.host[0].numel() * self.host.element_size())
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.nbuf = nbuf
        self.buf = [torch.empty(self.host.shape[1:], dtype=self.host.dtype, device=dev) for _ in range(nbuf)]
        self.ev_copied = [torch.cuda.Event() for _ in range(nbuf)]
        self.ev_used = [torch.cuda.Event() for _ in range(nbuf)]
        self.copied_upto = -1
        self.d_cnt, self.d_dt, self.outs = d_cnt, d_dt, outs
        S, N = self.host.shape[1], self.host.shape[2]
        if form in ("raw_f32", "tlv"):   # normalize_data's output (fp64, Utils.py:342-434) and the kept-row counts
            if form == "tlv":            # host_frames [F, S, stride] bytes: one TLV body per scene at a fixed stride
                N = max_pts
                self.tlv_off = torch.arange(S, dtype=torch.int64, device=dev) * int(self.host.shape[2])
                self.uart_cfg = uart_cfg
            self.norm = [torch.empty((S, N, 8), dtype=torch.float64, device=dev) for _ in range(2)]
            self.n_out = [torch.empty((S,), dtype=torch.int32, device=dev) for _ in range(2)]

    def _copy(self, f):
        torch, b = self.torch, f % self.nbuf
        with torch.cuda.stream(self.copy_stream):
            if f >= self.nbuf:
                self.copy_stream.wait_event(self.ev_used[b])    # the step that read this buffer nbuf frames ago
            self.buf[b].copy_(self.host[f], non_blocking=True)
            self.ev_copied[b].record(self.copy_stream)
        self.copied_upto = f

    def restart(self):
        self.torch.cuda.synchronize()
        self.copied_upto = -1

    def step(self, f):
        sb, b = self.sb, f % self.nbuf
        while self.copied_upto < min(f + 1, self.F - 1):          # this frame (first call) and the next one
            self._copy(self.copied_upto + 1)
        self.cs.wait_event(self.ev_copied[b])
        a, l, n = self.outs
        if self.form == "rows_f32":
            sb.step_dev_f32(self.buf[b].data_ptr(), self.d_cnt[f].data_ptr(), self.d_dt[f].data_ptr(), a.data_ptr(), l.data_ptr(), n.data_ptr())
        elif self.form == "rows_f64":
            sb.step_dev(self.buf[b].data_ptr(), self.d_cnt[f].data_ptr(), self.d_dt[f].data_ptr(), a.data_ptr(), l.data_ptr(), n.data_ptr())
        else:
            k = f & 1
            if self.form == "tlv":
                sb.normalize_tlv_dev(self.buf[b].data_ptr(), self.buf[b].numel() * self.buf[b].element_size(), self.tlv_off.data_ptr(), self.uart_cfg, self.norm[k].data_ptr(), self.n_out[k].data_ptr())
            else:
                sb.normalize_dev(self.buf[b].data_ptr(), self.d_cnt[f].data_ptr(), self.norm[k].data_ptr(), self.n_out[k].data_ptr(), f32=True)
            sb.step_dev(self.norm[k].data_ptr(), self.n_out[k].data_ptr(), self.d_dt[f].data_ptr(), a.data_ptr(), l.data_ptr(), n.data_ptr())
        self.ev_used[b].record(self.cs)

    def copies_alone_ms(self, frames):
        """the H2D copies of `frames` frames with nothing beside them: the PCIe floor of a step"""
        torch = self.torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(self.copy_stream):
            for f in range(frames):
                self.buf[f % self.nbuf].copy_(self.host[f % self.F], non_blocking=True)
        self.copy_stream.synchronize()
        return (time.perf_counter() - t0) / frames * 1e3


def _timed(sb, feed, W, F, barrier, max_over_ranks):
    import torch
    sb.reset()
    feed.restart()
    for f in range(W):
        feed.step(f)
    torch.cuda.synchronize()
    sb.check()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(W, F):
        feed.step(f)
    torch.cuda.synchronize()
    barrier()
    el = max_over_ranks(time.perf_counter() - t0)
    sb.check()
    return el


def ingest_leg(sb, pts, cnt, dts, d_cnt, d_dt, outs, W, F, S, world, barrier, max_over_ranks, dev, stream, resident_ms, e2e_resident_ms=None,
               with_e2e=True, oracle_scenes=48):
    """pts[F,S,N,8] float32 host frames of this rank (the resident legs' frames).  Returns the `ingest` object of the line."""
    import torch
    from mmwave_msc_amd import _lib

    K = F - W
    N = pts.shape[2]
    out = {"frames": K, "buffers": 3,
           "note": "frames from pinned host memory, H2D on a copy stream one frame ahead of the tracker (three device buffers); "
                   "outputs stay on the device as in the resident legs; ms_per_step / value as the headline's, over the same frame window"}
    cfg = sb.cfg

    def entry(feed, el, compute_note):
        ms = el / K * 1e3
        floor = feed.copies_alone_ms(min(K, 20))
        gbs = feed.bytes_per_frame / (ms * 1e-3) / 1e9
        return {"ms_per_step": round(ms, 4), "value": round(S * world * K / el, 1), "unit": "scene-frames/s",
                "h2d_bytes_per_step": feed.bytes_per_frame, "h2d_gb_per_s": round(gbs, 2),
                "h2d_alone_ms_per_step": round(floor, 4), "h2d_alone_gb_per_s": round(feed.bytes_per_frame / (floor * 1e-3) / 1e9, 2),
                "bound": ("pcie (the copies alone take %.0f %% of the step)" % (100 * floor / ms)) if floor >= 0.8 * ms else compute_note}

    # ---- normalised rows, fp32 and fp64 ----
    for form, host in (("rows_f32", pts), ("rows_f64", None)):
        if host is None:
            host = pts[: min(F, W + min(K, 12))].astype(np.float64)     # (a shorter window: 134 MB per frame)
        Ff = host.shape[0]
        feed = HostFeed(sb, host, form, d_cnt, d_dt, outs, dev, stream)
        el = _timed(sb, feed, W, Ff, barrier, max_over_ranks)
        Kf = Ff - W
        e = entry(feed, el * K / Kf, "the tracker's kernels (resident step %.4f ms)" % resident_ms)
        e["frames"] = Kf
        e["entry"] = "mmw_step_f32 (32 B per point, promoted to fp64 in registers)" if form == "rows_f32" else "mmw_step (64 B per point)"
        e["vs_resident"] = round(e["ms_per_step"] / resident_ms, 3)
        out[form] = e
        del feed
    # ---- raw radar rows -> mmw_normalize_f32 -> mmw_step ----
    raw = raw_rows_from_normalised(pts, float(cfg.tilt_cos), float(cfg.tilt_sin), float(cfg.s_height))
    feed = HostFeed(sb, raw, "raw_f32", d_cnt, d_dt, outs, dev, stream)
    sb.profile_reset()
    sb.stats_reset()
    sb.profile(True, kernels=(_lib.K_NORMALIZE,))
    el = _timed(sb, feed, W, F, barrier, max_over_ranks)
    sb.profile(False)
    nz_ms, nz_cnt = sb.profile_get(_lib.K_NORMALIZE)
    st = sb.stats()
    e = entry(feed, el, "normalize + the tracker's kernels")
    e["work"] = {"dbscan_calls_per_step": round(float(st[3]) / max(F, 1), 1), "mean_U": round(float(st[4]) / max(float(st[3]), 1.0), 1),
                 "tracks_per_scene": round(float(st[5]) / max(float(st[2]), 1.0), 2),
                 "note": "the tracker sees velocity columns made from the doppler (radial component only): not the workload of the rows_* forms"}
    nz_avg = nz_ms / max(nz_cnt, 1)
    nz_bytes = float(np.maximum(cnt, 0).sum()) / cnt.shape[0] * (20 + 64)     # rows in (fp32 raw) + rows out (fp64), per launch
    e["entry"] = "mmw_normalize_f32 (20 B per object in, 64 B per kept row out) + mmw_step"
    e["roofline_normalize"] = {"kernel": "k_normalize", "bound": "hbm", "achieved": round(nz_bytes / max(nz_avg, 1e-9) / 1e6, 2), "peak": 8000.0,
                               "unit": "GB/s", "frac": round(nz_bytes / max(nz_avg, 1e-9) / 1e6 / 8000.0, 6),
                               "algorithmic_bytes_per_launch": round(nz_bytes, 1), "avg_launch_ms": round(nz_avg, 5), "launches_timed": int(nz_cnt)}
    # parity of the raw form: Utils.normalize_data + TrackBuffer.track of the oracle on the first scenes, final state bit-equal
    try:
        from oracle import c_oracle as co
        ns = min(oracle_scenes, S)
        ocfg = co.default_config(tr_max_tracks=int(cfg.tr_max_tracks))
        ntr = sb.num_tracks()
        trk = sb.tracks(cap=max(int(ntr.max()), 1))
        ok = True
        for s in range(ns):
            sc = co.OracleScene(ocfg, N)
            for f in range(F):
                c = int(cnt[f, s])
                if c <= 0:
                    continue
                rows = co.normalize(ocfg, raw[f, s, :c].astype(np.float64))
                if len(rows):
                    sc.track(rows, float(dts[f, s]))
            want = sc.tracks()
            ok = ok and len(want) == int(ntr[s])
            if not ok:
                break
            for name in ("x", "P", "centroid", "spread_est", "group_disp_est", "lifetime", "point_num", "is_static", "ring_n"):
                ok = ok and bool(np.array_equal(trk[s, : ntr[s]][name], want[name]))
        e["parity"] = {"scenes_checked": ns, "frames": F, "bit_equal_vs_oracle": bool(ok),
                       "oracle": "oracle/c: normalize_data (Utils.py:342-434) + TrackBuffer.track on the same raw rows"}
    except Exception as exc:   # the checker must not cost the line
        e["parity"] = {"error": repr(exc)[:200]}
    out["raw_f32"] = e
    del feed
    # ---- the radar's wire format: TLV bodies (12 B per object) -> mmw_normalize_tlv -> mmw_step ----
    try:
        from mmwave_msc_amd import radar
        ucp = {"rangeIdxToMeters": 0.0436, "dopplerResolutionMps": 0.01, "numDopplerBins": 65536.0}   # (no index above the wrap threshold)
        QF = 9
        bodies = radar.encode_tlv_bodies(raw, np.maximum(cnt, 0), QF, ucp["dopplerResolutionMps"])   # [F, S, stride] uint8
        feed = HostFeed(sb, bodies, "tlv", d_cnt, d_dt, outs, dev, stream, max_pts=N, uart_cfg=radar.uart_cfg(ucp))
        sb.profile_reset()
        sb.profile(True, kernels=(_lib.K_NORMALIZE,))
        el = _timed(sb, feed, W, F, barrier, max_over_ranks)
        sb.profile(False)
        nz_ms, nz_cnt = sb.profile_get(_lib.K_NORMALIZE)
        e = entry(feed, el, "decode + normalize + the tracker's kernels")
        nz_avg = nz_ms / max(nz_cnt, 1)
        nz_bytes = float(np.maximum(cnt, 0).sum()) / cnt.shape[0] * (12 + 64) + 4.0 * S
        e["entry"] = "mmw_normalize_tlv (12 B per object in: the IWR1443's int16 objects, decoded on the device; 64 B per kept row out) + mmw_step"
        e["bytes_per_object"] = 12
        e["roofline_normalize"] = {"kernel": "k_normalize_tlv", "bound": "hbm", "achieved": round(nz_bytes / max(nz_avg, 1e-9) / 1e6, 2), "peak": 8000.0,
                                   "unit": "GB/s", "frac": round(nz_bytes / max(nz_avg, 1e-9) / 1e6 / 8000.0, 6),
                                   "algorithmic_bytes_per_launch": round(nz_bytes, 1), "avg_launch_ms": round(nz_avg, 5), "launches_timed": int(nz_cnt)}
        # parity: the reference's decode restated in numpy + the oracle's normalize_data + track() on the first scenes
        try:
            from oracle import c_oracle as co
            ns = min(oracle_scenes, S)
            ocfg = co.default_config(tr_max_tracks=int(cfg.tr_max_tracks))
            ntr = sb.num_tracks()
            trk = sb.tracks(cap=max(int(ntr.max()), 1))
            dec, dcnt = radar.decode_tlv_bodies_numpy(bodies[:, :ns], ucp)
            ok = True
            for s in range(ns):
                sc = co.OracleScene(ocfg, N)
                for f in range(F):
                    c = int(dcnt[f, s])
                    if c <= 0:
                        continue
                    rows = co.normalize(ocfg, dec[f, s, :c])
                    if len(rows):
                        sc.track(rows, float(dts[f, s]))
                want = sc.tracks()
                ok = ok and len(want) == int(ntr[s])
                if not ok:
                    break
                for name in ("x", "P", "centroid", "spread_est", "group_disp_est", "lifetime", "point_num", "is_static", "ring_n"):
                    ok = ok and bool(np.array_equal(trk[s, : ntr[s]][name], want[name]))
            e["parity"] = {"scenes_checked": ns, "frames": F, "bit_equal_vs_oracle": bool(ok),
                           "oracle": "ReadIWR14xx.read's decode restated in numpy (ReadDataIWR1443.py:153-171) + oracle/c normalize_data + "
                                     "TrackBuffer.track on the same bytes"}
        except Exception as exc:
            e["parity"] = {"error": repr(exc)[:200]}
        out["tlv"] = e
        del feed
    except Exception as exc:   # never lose the line over a further form
        out["tlv"] = {"error": repr(exc)[:300]}
    # ---- end to end, host-fed: rows_f32 + features + CNN + keypoints every frame ----
    if with_e2e:
        try:
            from bench_e2e import e2e_leg
            feed = HostFeed(sb, pts, "rows_f32", d_cnt, d_dt, outs, dev, stream)

            def step(f):
                if f == 0:
                    feed.restart()
                feed.step(f)
            r = e2e_leg(sb, step, W, F, S, world, barrier, max_over_ranks, dev, stream, modes=(("overlap", "f16x3"),))
            ee = {k: r[k] for k in ("value", "ms_per_step", "cnn_ms_per_step", "streams", "mode") if k in r}
            ee["h2d_gb_per_s"] = round(feed.bytes_per_frame / (r["ms_per_step"] * 1e-3) / 1e9, 2)
            if e2e_resident_ms:
                ee["resident_ms_per_step"] = e2e_resident_ms
                ee["vs_resident"] = round(r["ms_per_step"] / e2e_resident_ms, 3)
            out["e2e_rows_f32"] = ee
        except Exception as exc:
            out["e2e_rows_f32"] = {"error": repr(exc)[:300]}
    return out
