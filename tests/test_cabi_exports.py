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


def test_step_kernels_compile_without_scratch():
    """All three points-per-thread variants of the association kernel are kept free of register spills (a build
    that spilled, while its LDS-only barrier was still inline assembly, once produced spurious error bits on the
    GPU; the barrier is compiler builtins now, the no-spill rule stays).  This reads the compiler's own resource
    report so that a later change cannot reintroduce spills silently."""
    import os
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("hipcc not available")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mmwave_msc_amd", "csrc", "k_track.hip")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                          "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull],
                         capture_output=True, text=True, timeout=600)
    rep = out.stderr
    names = re.findall(r"Function Name: (\S*k_track\S*)", rep)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", rep)]
    vspill = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", rep)]
    assert len(names) >= 6 and len(scratch) >= 6 and len(vspill) >= 6, rep[-2000:]
    # no vector register is spilled in any instantiation
    assert all(v == 0 for v in vspill[: len(names)]), list(zip(names, vspill))
    # Any instantiation may keep an SGPR-spill stack slot the compiler reserves but never touches (scalars parked in the lanes of a
    # VGPR by v_writelane): the ISA of such a kernel must not hold a single scratch access -- the instantiations of the
    # benchmarked configurations (PPT 1 and 2, INNER = PRED = false, fp64 and fp32 rows: mangled "ILi<PPT>ELb0ELb0ELb<F32>E") included
    hot = [n for n in names if re.search(r"k_trackILi[12]ELb0ELb0ELb[01]E", n)]
    assert len(hot) == 4, names
    slotted = [n for n, v in zip(names, scratch) if v != 0]
    if slotted:
        asm = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                              "--cuda-device-only", "-S", "-c", src, "-o", "-"], capture_output=True, text=True, timeout=600).stdout
        for n in slotted:
            body = asm[asm.index("\n" + n + ":"):]
            body = body[: body.index("s_endpgm")]
            assert not re.search(r"\bscratch_|buffer_(load|store)\S* .*offen", body), n


def test_gate_records_are_not_read_before_the_scalar_cache_invalidate():
    """k_track's PRED instantiations and k_scene write the gate records (C^-1, log det, Hx per track) in the SAME launch that
    reads them back through the scalar cache (constant address space, s_load).  The compiler may treat such memory as
    unchanging, so the pointer is laundered through an asm statement behind the s_dcache_inv; this reads the ISA: the
    invalidate precedes the marker, the wide scalar loads of the gate loop follow it, and none of them sits in front of
    the invalidate (k_scene loads its 64-byte scene header with one s_load_dwordx16 there: exactly one is allowed)."""
    import os
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("hipcc not available")
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mmwave_msc_amd", "csrc")
    for src, pat, allowed_before in (("k_track.hip", r"^(_ZN3mmw7k_trackILi\dELb[01]ELb1ELb[01]EE\S*):", 0), ("k_scene.hip", r"^(_ZN3mmw7k_sceneILi\d+ELi\dELi\dELb[01]EE\S*):", 1)):
        asm = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only", "-S",
                              "-c", os.path.join(root, src), "-o", "-"], capture_output=True, text=True, timeout=900).stdout
        names = re.findall(pat, asm, flags=re.M)
        assert len(names) >= 6, (src, names)
        for n in names:
            body = asm[asm.index("\n" + n + ":"):]
            body = body[: body.index("s_endpgm")]
            inv, mark = body.find("s_dcache_inv"), body.find("; mmw: gate pointer opaque from here")
            assert 0 < inv < mark, (n, inv, mark)
            kern = re.search(r"s_load_dword\S* \S+ (s\[\d+:\d+\]), 0x", body).group(1)   # first scalar load: the kernel arguments
            wide = [(m.start(), m.group(1)) for m in re.finditer(r"s_load_dwordx16 \S+ (s\[\d+:\d+\]),", body)]
            before = [w for w in wide if w[0] < inv and w[1] != kern]
            after = [w for w in wide if w[0] > mark]   # (whatever pair the base sits in by then: the allocator may reuse the arguments' one)
            assert len(before) <= allowed_before and len(after) >= 5, (n, len(before), len(after))
