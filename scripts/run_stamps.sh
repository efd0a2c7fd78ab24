cd $GRAFT_REPO_ROOT
timeout 300 python scripts/phase_stamps.py 2>&1 | tail -25
bash scripts/gpu_round.sh tests bench 2>&1 | grep -E "passed|failed|Error|error|assert|metric" | cut -c1-200,750-1250
