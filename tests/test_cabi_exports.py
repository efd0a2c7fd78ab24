"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/mmw.h declares (no compute call is made here)."""
import ctypes as C
import os
import re

from mmwave_msc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mmw.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mmw_[a-z0-9_]+)\s*\(", txt)))


def test_library_is_built_and_exports_every_declared_symbol():
    assert os.path.isfile(_lib.LIB_PATH), "libmmw_hip.so missing: run __graft_entry__.build()"
    L = C.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/mmw.h but not exported"
    assert sorted(declared) == sorted(_lib.EXPORTS), "python binding and header disagree"


def test_config_struct_matches_header_layout():
    L = _lib.load()
    cfg = _lib.default_config()
    # defaults = reference constants.py
    assert (cfg.fb_frames_batch, cfg.db_min_samples, cfg.tr_max_tracks, cfg.dim_x) == (2, 35, 4, 9)
    assert (cfg.db_eps, cfg.db_z_weight, cfg.db_range_weight, cfg.tr_gate) == (0.3, 0.4, 0.03, 4.5)
    assert list(cfg.kf_spread_lim) == [0.2, 0.2, 2, 1.2, 1.2, 0.2]
    assert abs(cfg.default_posture[22] - 1.5513) < 1e-6 and abs(cfg.default_posture[56] - 0.0312) < 1e-6
    import numpy as np
    assert cfg.tilt_cos == float(np.cos(np.radians(-5))) and cfg.tilt_sin == float(np.sin(np.radians(-5)))
    assert L.mmw_version().startswith(b"mmw-hip")
    assert C.sizeof(_lib.MmwConfig) == 8 * 4 + 24 * 8 + 57 * 4 + 5 * 4 + 8 * 8 + 2 * 4  # 8 ints, 24 doubles, 57 floats, 5 ints, 8 doubles, 2 ints
    assert cfg.fused_step == 0 and cfg.chain_side_stream == 0
    assert C.sizeof(_lib.MmwPostureModel) == 9 * 8   # struct mmw_posture_model: eight pointers and one int64


def test_create_fails_loudly_without_gpu_or_with_bad_args():
    import torch
    L = _lib.load()
    cfg = _lib.default_config()
    h = C.c_void_p()
    assert L.mmw_create(C.byref(cfg), 1, 4096, 0, C.byref(h)) == _lib.E_ARG  # max_pts limit
    cfg5 = _lib.default_config(fb_frames_batch=4)
    assert L.mmw_create(C.byref(cfg5), 1, 256, 0, C.byref(h)) == _lib.E_ARG  # ring = FB_FRAMES_BATCH + 1 > MMW_RING_MAX
    # (ring * max_pts itself has no limit of its own any more: clouds of more than 1920 points run on k_dbscan_huge)
    if not torch.cuda.is_available():
        rc = L.mmw_create(C.byref(cfg), 1, 256, 0, C.byref(h))
        assert rc in (_lib.E_NODEVICE, _lib.E_HIP) and not h.value
        assert L.mmw_last_error(None)


def test_track_record_dtype_matches_c_struct():
    # x9 P81 c6 mn6 mx6 sp6 gd36 n_est lifetime = 152 doubles; 8 ints; 57 floats (+pad to 8)
    assert _lib.TRACK_DTYPE.itemsize == 152 * 8 + 8 * 4 + 57 * 4 + 4
    assert _lib.SUMMARY_DTYPE.itemsize == 5 * 4 + 4 + (9 + 6 + 57) * 4 + 3 * 4


_ISA_CACHE = {}


def _device_isa(names):
    """(resource report, ISA text) of csrc/<name>.hip for gfx950 with the product's flags: ONE device-only compile per file gives
    both (the remarks on stderr, the assembly on stdout); files compile side by side and are cached for the session."""
    import os
    import shutil
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("hipcc not available")
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mmwave_msc_amd", "csrc")

    def one(name):
        out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only",
                              "-Rpass-analysis=kernel-resource-usage", "-S", "-c", os.path.join(root, name + ".hip"), "-o", "-"],
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stderr, out.stdout

    todo = [n for n in names if n not in _ISA_CACHE]
    with ThreadPoolExecutor(max_workers=4) as ex:
        for n, r in zip(todo, ex.map(one, todo)):
            _ISA_CACHE[n] = r
    return {n: _ISA_CACHE[n] for n in names}


def _kernel_report(rep):
    """[(mangled name, scratch bytes per lane, spilled VGPRs, VGPRs, waves per SIMD, spilled SGPRs)] from -Rpass-analysis=kernel-resource-usage."""
    import re
    out = []
    for blk in re.split(r"remark: Function Name: ", rep)[1:]:
        name = blk.split()[0]
        f = lambda pat: int(re.search(pat, blk).group(1))
        out.append((name, f(r"ScratchSize \[bytes/lane\]: (\d+)"), f(r"VGPRs Spill: (\d+)"), f(r" VGPRs: (\d+)"), f(r"Occupancy \[waves/SIMD\]: (\d+)"),
                    f(r"SGPRs Spill: (\d+)")))
    return out


# The one kernel that is ALLOWED to spill, with the ceiling it must stay under: k_dbscan_startup (frame 0 after a reset: every scene
# clusters its whole first frame) takes two 512-thread workgroups per CU, i.e. 128 registers per lane, for a BallTree chain that wants
# ~180 -- measured 3.0 ms per 4096 clouds that way against 4.0 ms unspilled at one workgroup per CU (k_dbscan.hip: launch_dbscan_big).
_MAY_SPILL = {"k_dbscan_startup": 52}
_SGPR_SPILL_CEILING = {"k_track": 56, "k_scene": 141, "k_predict": 12, "k_inner": 250, "k_chain": 306, "k_dbscan_big": 271, "k_dbscan_startup": 240,
                       "k_dbscan_only": 83, "k_dbscan_huge": 257, "k_dbscan_only_huge": 181, "k_post": 567}


def test_step_kernels_compile_without_scratch():
    """No kernel of a step touches scratch memory: the association kernel (all its instantiations), the one-workgroup step
    (k_scene, every points-per-thread variant), the batched Kalman kernels (k_predict, k_post's update), the DBSCAN workers
    (k_post's worker blocks, k_chain, k_dbscan_big, k_dbscan_huge, k_inner).  Read from the compiler's own resource report AND
    from the ISA: zero spilled VGPRs, and not one scratch access -- a kernel may keep an SGPR-spill stack slot the compiler
    reserves but never touches (scalars parked in the lanes of a VGPR by v_writelane).  (A spilling build once cost 9 minutes
    instead of 16 s on the GPU; round 4's review found 54 spilled VGPRs in k_post and 2 in k_predict that nothing guarded.)"""
    import re
    files = ("k_track", "k_scene", "k_kalman", "k_dbscan")
    isa = _device_isa(files)
    seen = {}
    for f in files:
        rep, asm = isa[f]
        kernels = _kernel_report(rep)
        assert kernels, rep[-2000:]
        for name, scratch, vspill, vgprs, occ, sspill in kernels:
            short = re.search(r"\d+(k_[a-z_]+?)(I|E)", name).group(1)
            seen.setdefault(short, []).append((name, vgprs, occ))
            # spilled SGPRs: every one is a v_writelane / v_readlane pair with its wait states inside a latency chain (the round-5
            # review counted 505 in k_post, 300 in k_chain that nothing guarded).  A ratchet: the ceilings are the numbers of the
            # build this test was written on (k_chain: + 3 for the tagged claim words of round 6, a correctness fix); lower them when a kernel improves, never raise them without a measurement.
            assert sspill <= _SGPR_SPILL_CEILING[short], (name, sspill, _SGPR_SPILL_CEILING[short])
            if short in _MAY_SPILL:
                assert vspill <= _MAY_SPILL[short], (name, vspill)
                continue
            assert vspill == 0, (name, vspill, vgprs, occ)
            if scratch:
                body = asm[asm.index("\n" + name + ":"):]
                body = body[: body.index("s_endpgm")]
                assert not re.search(r"\bscratch_|buffer_(load|store)\S* .*offen", body), (name, scratch)
    for k in ("k_track", "k_scene", "k_predict", "k_post", "k_chain", "k_dbscan_big", "k_dbscan_huge", "k_inner", "k_dbscan_startup"):
        assert k in seen, (k, sorted(seen))
    assert len(seen["k_track"]) == 24 and len(seen["k_scene"]) == 12 and len(seen["k_post"]) == 4 and len(seen["k_predict"]) == 2   # (k_post: two motion models x 256- and 512-thread blocks)
    # the benchmarked instantiations keep the occupancy their launch geometry is sized for
    for name, vgprs, occ in seen["k_track"]:
        if re.search(r"k_trackILi[12]ELb0ELb0ELb[01]E", name):
            assert occ >= (5 if "ILi2E" in name else 4) or vgprs <= 96, (name, vgprs, occ)


def test_gate_records_are_not_read_before_the_scalar_cache_invalidate():
    """k_track's PRED instantiations and k_scene write the gate records (C^-1, log det, Hx per track) in the SAME launch that
    reads them back through the scalar cache (constant address space, s_load).  The compiler may treat such memory as
    unchanging, so the pointer is laundered through an asm statement behind the s_dcache_inv; this reads the ISA: the
    invalidate precedes the marker, the wide scalar loads of the gate loop follow it, and none of them sits in front of
    the invalidate (k_scene loads its 64-byte scene header with one s_load_dwordx16 there: exactly one is allowed)."""
    import re
    isa = _device_isa(("k_track", "k_scene"))
    for src, pat, allowed_before in (("k_track.hip", r"^(_ZN3mmw7k_trackILi\dELb[01]ELb1ELb[01]EE\S*):", 0), ("k_scene.hip", r"^(_ZN3mmw7k_sceneILi\d+ELi\dELi\dELb[01]EE\S*):", 1)):
        asm = isa[src[:-4]][1]
        names = re.findall(pat, asm, flags=re.M)
        assert len(names) >= 6, (src, names)
        for n in names:
            body = asm[asm.index("\n" + n + ":"):]
            body = body[: body.index("s_endpgm")]
            inv, mark = body.find("s_dcache_inv"), body.find("; mmw: gate pointer opaque from here")
            assert 0 < inv < mark, (n, inv, mark)
            kern = re.search(r"s_load_dword\S* \S+ (s\[\d+:\d+\]), 0x", body).group(1)   # first scalar load: the kernel arguments
            wide = [(m.start(), m.group(1)) for m in re.finditer(r"s_load_dwordx16 \S+ (s\[\d+:\d+\]|vcc),", body)]   # (the allocator has kept the base in vcc, too)
            before = [w for w in wide if w[0] < inv and w[1] != kern]
            after = [w for w in wide if w[0] > mark]   # (whatever pair the base sits in by then: the allocator may reuse the arguments' one)
            assert len(before) <= allowed_before and len(after) >= 5, (n, len(before), len(after))


def test_dense1_tile_kernel_keeps_its_pipeline():
    """k_mars_dense1_t (csrc/k_dense.hip; three tile shapes) is software-pipelined by hand and only works as built if the compiler keeps
    three properties, read here from its report and ISA: (1) no scratch at up to 250 VGPRs and two waves per SIMD -- a spill's reload is
    an `s_waitcnt vmcnt(0)` in the middle of the counted waits for the LDS-DMA pieces; (2) the fragment waits of the steady-state loop
    are COUNTED (lgkmcnt(6) ...: fragments are requested a phase ahead and must not be waited for as a block -- what the compiler
    did to every wait while the requests were the LDS-DMA builtin); (3) between the barrier of a K-step and the first MFMA that
    follows it there is no LDS wait at all (that phase's operands are in registers)."""
    import re
    rep, asm = _device_isa(("k_dense",))["k_dense"]
    # (mangled template arguments: TBM, WN, NT, W slots) -> MFMAs and LDS-DMA pieces of a K-step per wave, pieces in flight across the barrier
    shapes = {"ILi256ELi2ELi3ELi2EE": (36, 7, 4), "ILi256ELi2ELi2ELi3EE": (24, 6, 6), "ILi128ELi4ELi1ELi3EE": (12, 4, 4)}
    rows = {k[0]: k for k in _kernel_report(rep) if "k_mars_dense1_t" in k[0]}
    assert len(rows) == 3, list(rows)
    for name, (_, scratch, vspill, vgprs, occ, sspill) in rows.items():
        n_mfma, n_dma, vm = next(v for k, v in shapes.items() if k in name)
        assert scratch == 0 and vspill == 0 and sspill == 0 and vgprs <= 256 and occ >= 2, rows[name]
        body = asm[asm.index("\n" + name + ":"):]
        body = body[: body.index("s_endpgm")]
        assert "scratch_" not in body
        # the steady-state loop: the basic block with the back edge that holds a K-step's MFMAs and one barrier
        blocks = re.split(r"\n\.LBB\d+_\d+:", body)
        loops = [b for b in blocks if b.count("v_mfma_f32_32x32x16_f16") == n_mfma and b.count("s_barrier") == 1 and b.count("global_load_lds_dwordx4") == n_dma]
        assert loops, (name, [(b.count("v_mfma"), b.count("s_barrier"), b.count("global_load_lds_dwordx4")) for b in blocks])
        for b in loops:
            waits = re.findall(r"s_waitcnt lgkmcnt\((\d+)\)", b)
            assert waits.count("0") == 1 and len(waits) >= 3 and max(int(w) for w in waits) >= 2, (name, waits)   # the one lgkmcnt(0) stands in front of the barrier
            bar = b.index("s_barrier")
            assert b.rfind("s_waitcnt lgkmcnt(0)", 0, bar) > 0
            first_mfma = b.index("v_mfma_f32_32x32x16_f16", bar)
            assert "s_waitcnt lgkmcnt" not in b[bar:first_mfma], (name, b[bar:first_mfma])
            assert re.search(r"s_waitcnt vmcnt\(%d\)" % vm, b[:bar]), name
