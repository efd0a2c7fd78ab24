cd $GRAFT_REPO_ROOT
for S in 1024 1280 2048 2560 3840 4096 5120 8192; do
  timeout 200 python bench.py --no-cpu --no-posture --gen-workers 32 --scenes $S --steps 24 --warmup 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels'] if 'kernels' in d['roofline'] else d.get('kernels')
print($S, d['ms_per_step'], {n:round(v['avg_ms']*1e3,1) for n,v in k.items() if isinstance(v,dict)})"
done
