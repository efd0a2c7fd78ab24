# scratch command file for ad-hoc gpurun experiments
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -2
bash scripts/ab_bench.sh libmmw_hip_prev.so libmmw_hip.so
