# usage: bash scripts/chain_blocks.sh -- the step with 2 / 4 / 8 DBSCAN workgroups on the side stream (make DIAG=cb2 DIAGFLAGS=-DMMW_CHAIN_BLOCKS=2, ...)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for R in 1 2; do
for L in libmmw_hip.so; do
  for W in "--steps 100 --warmup 20"; do
  MMW_LIB_NAME=$L python3 bench.py --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single $W > gpurun_out/cb.json 2> gpurun_out/cb.err
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/cb.json").read().strip().splitlines()[-1])
    print("$L", "$W", d["ms_per_step"], {k: round(v["avg_ms"]*1e3,1) for k,v in d["kernels"].items()}, d.get("parity"))
except Exception as e:
    print("$L", "ERR", e, open("gpurun_out/cb.err").read()[-400:])
PY
  done
done
done
