# A/B on ONE box: bench.py with two builds of the library, alternating (box-to-box and run-to-run noise is
# about +-1 %).  usage (GPU box): bash scripts/ab_bench.sh libA.so libB.so [bench args]
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for lib in $A $B; do
    MMW_LIB_NAME=$lib timeout 600 python bench.py --no-cpu --no-posture "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', d['value'], d['ms_per_step'], {k: round(v['avg_ms'] * 1e3, 1) for k, v in d['kernels'].items()})"
  done
done
