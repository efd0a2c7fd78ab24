cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for S in 512 1024 4096; do echo "== stamps S=$S"; timeout 300 python scripts/phase_stamps.py $S 2>&1 | head -14; done > gpurun_out/r03a_stamps.txt 2>&1
for S in 512 1024 2048; do
  rm -rf gpurun_out/prof_s$S
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_s$S -- python3 $GRAFT_REPO_ROOT/bench.py --scenes $S --no-cpu --no-e2e --no-e2e-parity --no-cold --gen-workers 1 --steps 40 --warmup 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_s$S.log 2>&1)
  tail -1 gpurun_out/prof_s$S.log | cut -c1-300
  for f in $(find gpurun_out/prof_s$S -name "*kernel_stats.csv"); do head -8 $f | cut -c1-200; done
done
cat gpurun_out/r03a_stamps.txt
