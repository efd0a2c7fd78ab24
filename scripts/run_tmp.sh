# scratch command file for ad-hoc gpurun experiments
cd $GRAFT_REPO_ROOT
python scripts/exp_big_chain.py 8 2>&1 | tail -8 | head -1
MMW_LIB_NAME=libmmw_hip_stamps.so python scripts/exp_big_chain.py 8 2>&1 | tail -14 | head -3
bash scripts/gpu_round.sh tests bench 2>&1 | grep -E "passed|failed|Error|error|assert|metric" | cut -c1-200,1290-1700
