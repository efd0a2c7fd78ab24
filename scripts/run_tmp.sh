# scratch command file for ad-hoc gpurun experiments
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_dataset.py -x -q -m gpu 2>&1 | tail -15
