# scratch command file for ad-hoc gpurun experiments
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 150 --warmup 10 --no-posture 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('parity'), d['work'])"
