# usage (GPU box): bash scripts/gpu_round.sh [tests|smoke|bench|prof|pmc ...]
# Runs the requested stages; outputs that matter are written under gpurun_out/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGES="${@:-tests smoke bench}"
for st in $STAGES; do
  case $st in
    tests)
      timeout 1500 python -m pytest tests -q -m gpu --durations=5 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.txt ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.txt ;;
    bench)
      timeout 1500 python bench.py 2>&1 | tail -2 | tee gpurun_out/bench_default.json ;;
    bench256)
      timeout 900 python bench.py --scenes 256 --pts 256 --tracks 4 --c-scenes 256 --no-e2e --no-e2e-parity --no-shards --no-full --no-ingest --no-single 2>&1 | tail -2 | tee gpurun_out/bench_256.json ;;
    bench160)
      timeout 900 python bench.py --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --steps 150 --warmup 10 2>&1 | tail -1 | tee gpurun_out/bench_160frames.json ;;
    prof)
      rm -rf gpurun_out/prof
      (cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --gen-workers 1 --steps 40 --warmup 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1)
      tail -2 gpurun_out/prof_bench.log
      find gpurun_out/prof -name "*kernel_stats.csv" | head -3
      for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -12 $f; done
      # keep only the summaries (the raw trace is large)
      find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete ;;
    pmc)
      rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
      # --chain-side-stream 2: the side-stream workers are launched although counter collection serialises the kernels (the
      # concurrency probe would turn them off): k_chain is then a launch of its own in these passes -- it finds nothing to claim
      # while it runs alone and leaves; the DBSCAN bytes it would have moved are k_post's / k_dbscan_big's here
      PMC_ARGS="--no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --gen-workers 1 --steps 10 --warmup 10 --chain-side-stream 2"
      (cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $PMC_ARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1)
      (cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $PMC_ARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1)
      python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write | tee gpurun_out/pmc_summary.json
      find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +20M -delete ;;
    pmc512)   # the 8-GPU shard (512 scenes: k_scene + k_post workers)
      rm -rf gpurun_out/pmc512_fetch gpurun_out/pmc512_write
      P5="--scenes 512 --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --gen-workers 1 --steps 10 --warmup 10"
      (cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc512_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $P5 > $GRAFT_REPO_ROOT/gpurun_out/pmc512_fetch.log 2>&1)
      (cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc512_write -- python3 $GRAFT_REPO_ROOT/bench.py $P5 > $GRAFT_REPO_ROOT/gpurun_out/pmc512_write.log 2>&1)
      python scripts/pmc_summary.py gpurun_out/pmc512_fetch gpurun_out/pmc512_write | tee gpurun_out/pmc512_summary.json
      find gpurun_out/pmc512_fetch gpurun_out/pmc512_write -name "*.csv" -size +20M -delete ;;
    prof512)
      rm -rf gpurun_out/prof512
      (cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof512 -- python3 $GRAFT_REPO_ROOT/bench.py --scenes 512 --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --gen-workers 1 --steps 40 --warmup 10 > $GRAFT_REPO_ROOT/gpurun_out/prof512.log 2>&1)
      for f in $(find gpurun_out/prof512 -name "*kernel_stats.csv"); do head -6 $f | cut -c1-200; done
      find gpurun_out/prof512 -name "*kernel_trace.csv" -size +20M -delete ;;
    posture)
      rm -rf gpurun_out/prof_posture gpurun_out/pmc_posture
      (cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_posture -- python3 $GRAFT_REPO_ROOT/scripts/bench_posture.py > $GRAFT_REPO_ROOT/gpurun_out/prof_posture.log 2>&1)
      tail -1 gpurun_out/prof_posture.log
      for f in $(find gpurun_out/prof_posture -name "*kernel_stats.csv"); do head -9 $f | cut -c1-180; done
      (cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_posture -- python3 $GRAFT_REPO_ROOT/scripts/bench_posture.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_posture.log 2>&1)
      python scripts/mfma_summary.py gpurun_out/pmc_posture | tee gpurun_out/mfma_summary.json
      find gpurun_out/prof_posture gpurun_out/pmc_posture -name "*.csv" -size +20M -delete ;;
    *) echo "unknown stage $st" ;;
  esac
done
