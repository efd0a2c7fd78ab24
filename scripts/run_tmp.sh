# scratch command file for ad-hoc gpurun experiments
cd $GRAFT_REPO_ROOT
MMW_LIB_NAME=libmmw_hip_stamps.so python scripts/exp_big_chain.py 8 512 8 2>&1 | grep -A1 "frame 0"
python scripts/exp_big_chain.py 8 512 8 2>&1 | grep "frame 0"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 900 python bench.py --no-cpu --no-posture --steps 150 --warmup 10 2>&1 | tail -1 | cut -c1-200
timeout 900 python bench.py --no-cpu --no-posture 2>&1 | tail -1 | cut -c1-200
