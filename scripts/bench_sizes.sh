# usage: bash scripts/bench_sizes.sh [sizes...]   -- tracker-only bench lines at the shard sizes (no CPU legs)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for S in ${@:-512 1024 4096}; do
  timeout 600 python bench.py --scenes $S --no-cpu --no-e2e --no-e2e-parity --no-cold --steps 40 --warmup 10 $MMW_BENCH_EXTRA 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print(l['config']['scenes_per_gpu'], l['value'], l['ms_per_step'], l['step_kernels'][:8], {k:v['avg_ms'] for k,v in l['kernels'].items()})"
done
