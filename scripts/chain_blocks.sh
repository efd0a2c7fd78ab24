# usage: bash scripts/chain_blocks.sh [LIB ...] -- A/B of the step between builds of the library on ONE box, alternating, two rounds:
# the product build and diagnostic builds of the same sources (csrc/Makefile: make DIAG=cb4 DIAGFLAGS=-DMMW_CHAIN_BLOCKS=4 ->
# ../libmmw_hip_cb4.so; MMW_LIB_NAME selects one).  Round 4 used it for the number of DBSCAN workgroups on the side stream.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LIBS="${@:-libmmw_hip.so}"
for R in 1 2; do
for L in $LIBS; do
  for W in "--steps 100 --warmup 20" "--steps 150 --warmup 10"; do
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
