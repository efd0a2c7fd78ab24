cd $GRAFT_REPO_ROOT
for m in "" 1; do
  MMW_EXPERIMENT_SORT_SCENES=$m timeout 250 python bench.py --no-cpu --no-posture --gen-workers 32 --steps 40 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sorted' if '$m' else 'plain', d['value'], d['ms_per_step'], {n:round(v['avg_ms']*1e3,1) for n,v in d['kernels'].items()})"
done
