# usage: bash scripts/kpost_parts.sh -- what the parts of k_post cost: the bench's kernel timings with diagnostic builds that leave one
# part out (make DIAG=nosort|noupd|nowork, see csrc/Makefile; their results are wrong on purpose)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in libmmw_hip.so libmmw_hip_nosort.so libmmw_hip_noupd.so libmmw_hip_nowork.so libmmw_hip.so; do
  MMW_LIB_NAME=$L python3 bench.py --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --steps 60 --warmup 15 > gpurun_out/kp_$L.json 2> gpurun_out/kp_$L.err
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/kp_$L.json").read().strip().splitlines()[-1])
    print("$L", d["ms_per_step"], {k: round(v["avg_ms"]*1e3,1) for k,v in d["kernels"].items()})
except Exception as e:
    print("$L", "ERR", e, open("gpurun_out/kp_$L.err").read()[-600:])
PY
done
