cd $GRAFT_REPO_ROOT
timeout 600 python scripts/bench_e2e.py 2>&1 | tail -12
