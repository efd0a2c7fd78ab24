cd $GRAFT_REPO_ROOT
timeout 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --durations=5 2>&1 | tail -15
python - <<'PY'
import sys, numpy as np, time
sys.path.insert(0,'.')
from mmwave_msc_amd import _lib
from mmwave_msc_amd.batch import SceneBatch
from mmwave_msc_amd.synth import make_batch
S,N,F=64,256,12
sb=SceneBatch(_lib.default_config(),S,N)
p,c,d=make_batch(range(S),F,N,3)
for f in range(F):
    a,l,n=sb.step_host(p[f].astype(np.float64),c[f],d[f])
    print(f, sb.num_tracks()[:10], n[:6], (a>=0).sum())
t=sb.tracks(cap=4)
print(t[0,0]['x'], t[0,0]['lifetime'], t[0,0]['point_num'])
PY
