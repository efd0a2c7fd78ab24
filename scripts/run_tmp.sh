cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for pe in 4 1; do
  sed -i "s/^PROF_EVERY = .*/PROF_EVERY = $pe  # hipEvent-timed steps inside the timed region: one in PROF_EVERY/" bench.py
  timeout 250 python bench.py --no-cpu --no-posture --gen-workers 32 --steps 40 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('every', $pe, d['value'], d['ms_per_step'], d['roofline']['launches_timed'], {n:round(v['avg_ms']*1e3,1) for n,v in d['kernels'].items()})"
done
