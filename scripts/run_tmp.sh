cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_mars_conv.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python - <<'PY'
import time, torch, numpy as np, sys
sys.path.insert(0,'.')
from mmwave_msc_amd.mars import MarsCNN, random_keras_weights
dev=torch.device("cuda:0"); B=18000
x=torch.randn(B,3,8,8,5,device=dev)
m=MarsCNN.from_keras_weights(random_keras_weights(0,3)).to(dev)
def timeit(fn,K=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(K): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/K*1e3
with torch.no_grad():
    t=timeit(lambda: m._hip_convs(x)); print("hip fused convs: %.3f ms  (%.1f TFLOP/s of 157.3)"%(t, B*6.138e6/t/1e9))
    print("full model (hip conv): %.3f ms"%timeit(lambda: m(x)))
    m.use_hip_conv=False
    print("full model (torch conv): %.3f ms"%timeit(lambda: m(x)))
PY
