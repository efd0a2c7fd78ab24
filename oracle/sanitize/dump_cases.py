"""oracle/sanitize/dump_cases.py -- test infrastructure: the inputs of fuzz cases as flat binary files for oracle/sanitize/replay.c
(see there)."""
import os
os.makedirs("/tmp/msan", exist_ok=True)
import sys, struct, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests._fuzz import draw_case, scene_inputs, plant_nonfinite
from oracle import c_oracle as co
def dump(seed, nf, path, posture=False):
    if posture:
        from tests.test_gpu_posture_fuzz import _case
        case = _case(seed)
    else:
        case = draw_case(seed)
    kw, S, N, F = case["cfg"], case["S"], case["N"], case["F"]
    pts, cnt, dts = scene_inputs(case)
    if nf: plant_nonfinite(case, pts, cnt, rate=0.3)
    cfg = co.default_config(**kw)
    raw = bytes(C.string_at(C.addressof(cfg), C.sizeof(cfg)))
    with open(path, 'wb') as f:
        f.write(struct.pack('iiii', len(raw), S, N, F)); f.write(raw)
        f.write(np.ascontiguousarray(pts, dtype=np.float64).tobytes())
        f.write(np.ascontiguousarray(cnt, dtype=np.int32).tobytes())
        f.write(np.ascontiguousarray(dts, dtype=np.float64).tobytes())
    print(path, seed, S, N, F, len(raw), kw.get('seek_inner'))
dump(80163, True, '/tmp/msan/c80163.bin')
dump(70155, False, '/tmp/msan/c70155.bin')
dump(60447, False, '/tmp/msan/c60447.bin', posture=True)
for s in range(20): dump(5000 + s, True, f'/tmp/msan/nf{s}.bin')
for s in range(20): dump(s, False, f'/tmp/msan/r{s}.bin')
