import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch
from mmwave_msc_amd.synth import make_batch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, F = 256, 6
pts, cnt, dts = make_batch(range(40, 40 + S), F, N, 3)
sb = SceneBatch(_lib.default_config(), S, N, device=0)
q = np.zeros(32, dtype=np.int32)
for f in range(F):
    t = time.perf_counter()
    a, l, d = sb.step_host(pts[f].astype(np.float64), cnt[f], dts[f])
    sb.L.mmw_diag_queue(sb.h, q.ctypes.data)
    print(f, "%.3f s" % (time.perf_counter() - t), q[:5], q[8:11], q[16:19], q[24:27], d[:4], sb.num_tracks()[:4], flush=True)
