"""usage: python scripts/fuzz_debug.py SEED [LAYOUT] -- one case of tests/test_gpu_fuzz.py with the differences printed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests._fuzz import draw_case, scene_inputs
from tests._layouts import layout_kwargs
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch
from oracle import c_oracle as co

seed = int(sys.argv[1]); layout = sys.argv[2] if len(sys.argv) > 2 else "per_scene"
case = draw_case(seed); kw = dict(case["cfg"]); S, N, F = case["S"], case["N"], case["F"]
print("case", S, N, F, kw)
pts, cnt, dts = scene_inputs(case)
cfg = co.default_config(**kw)
kw2 = dict(kw); kw2.update(layout_kwargs(layout))
sb = SceneBatch(_lib.default_config(**kw2), S, N)
print("step_kind", sb.step_kind(), "dense", sb.kalman_layout(), "t_cap", sb.track_cap, "UM", sb.UM)
scenes = [co.OracleScene(cfg, N) for _ in range(S)]
from collections import deque
rings = [deque(maxlen=kw["fb_frames_batch"] + 1) for _ in range(S)]
saved = 0
for f in range(F):
    want = [None] * S
    for s in range(S):
        c = int(cnt[f, s])
        if c != 0:
            try:
                want[s] = scenes[s].track(pts[f, s, :max(c, 0)].astype(np.float64), float(dts[f, s]))
            except RuntimeError as e:
                print("oracle error", f, s, e); want[s] = "err"
    try:
        assoc, labels, dbn = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
    except _lib.MmwError as e:
        print("gpu error", f, e, sb.errors()); break
    ntr = sb.num_tracks()
    for s in range(S):
        c = int(cnt[f, s])
        if c == 0 or want[s] == "err": continue
        oa, ol = want[s]; n = max(c, 0)
        rings[s].append(pts[f, s, :n][oa < 0].astype(np.float64))
        cloud = np.concatenate(list(rings[s])) if len(rings[s]) else np.zeros((0, 8))
        if ol is not None and len(ol) and ol.max() >= 0: rings[s].clear()
        if not np.array_equal(assoc[s, :n], oa): print("ASSOC", f, s, np.nonzero(assoc[s, :n] != oa)[0][:10])
        if (ol is None) != (dbn[s] < 0): print("DBN", f, s, dbn[s], None if ol is None else len(ol))
        elif ol is not None:
            g = labels[s, :len(ol)]
            if dbn[s] != len(ol) or not np.array_equal(g, ol):
                bad = np.nonzero(g != ol)[0]
                if saved < 4:
                    os.makedirs("gpurun_out", exist_ok=True)
                    np.savez(f"gpurun_out/fuzzcloud_{seed}_{f}_{s}.npz", cloud=cloud, want=ol, got=g, eps=kw["db_eps"], ms=kw["db_min_samples"], zw=kw["db_z_weight"], rw=kw["db_range_weight"])
                    saved += 1
                assert len(cloud) == len(ol), (len(cloud), len(ol))
                print("LABELS", f, s, "U", len(ol), "dbn", dbn[s], "bad idx", bad[:12], "...", bad[-3:], "n_bad", len(bad), "gpu", g[bad[:6]], "want", ol[bad[:6]], "ncl", ol.max() + 1, "cnt", c, "ring", scenes[s].batch_ring())
        if ntr[s] != scenes[s].n_tracks: print("NTR", f, s, ntr[s], scenes[s].n_tracks)
print("done")
