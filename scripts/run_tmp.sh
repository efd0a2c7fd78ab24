# scratch command file for ad-hoc gpurun experiments
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/phase_stamps.py 2>&1 | tail -25
