# usage (GPU box): bash scripts/ab_libs.sh "<lib> <lib> ..." [scenes ...] -- same-box A/B of diagnostic builds (csrc/Makefile: make DIAG=<name>
# DIAGFLAGS=...): the bench's step and kernel timings for every library, alternating, two rounds, at each context size
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LIBS="$1"; shift
SIZES="${@:-4096 512}"
for S in $SIZES; do
 for rnd in 1 2; do
  for L in $LIBS; do
    MMW_LIB_NAME=$L python3 bench.py --scenes $S --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --steps ${STEPS:-80} --warmup ${WARMUP:-20} > gpurun_out/ab_$L.json 2> gpurun_out/ab_$L.err
    python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_$L.json").read().strip().splitlines()[-1])
    print("S=$S r$rnd $L", d["ms_per_step"], {k: round(v["avg_ms"]*1e3,1) for k,v in d["kernels"].items()}, "parity", d.get("parity",{}).get("ok"))
except Exception as e:
    print("S=$S $L", "ERR", e, open("gpurun_out/ab_$L.err").read()[-600:])
PY
  done
 done
done
