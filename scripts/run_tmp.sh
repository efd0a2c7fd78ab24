cd $GRAFT_REPO_ROOT
timeout 300 python scripts/probe_timeline.py 2>&1 | tail -6
