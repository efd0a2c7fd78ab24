# usage: bash scripts/fused_small.sh -- small contexts: the library's step against the one-workgroup step (mmw_config.fused_step = 1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for R in 1 2; do
for CFG in "64 256 4" "128 256 4" "256 256 4" "128 512 8" "256 512 8"; do
  set -- $CFG
  for FS in 0 1; do
  python3 bench.py --scenes $1 --pts $2 --tracks $3 --c-scenes $1 --fused-step $FS --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --steps 100 --warmup 20 > gpurun_out/fs.json 2> gpurun_out/fs.err
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/fs.json").read().strip().splitlines()[-1])
    print("S=$1 N=$2 T=$3 fused_step=$FS", d["ms_per_step"], d.get("step_kernels"), {k: round(v["avg_ms"]*1e3,1) for k,v in d["kernels"].items()})
except Exception as e:
    print("S=$1 N=$2 T=$3 fused_step=$FS", "ERR", e, open("gpurun_out/fs.err").read()[-300:])
PY
  done
done
done
