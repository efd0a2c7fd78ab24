# usage: bash scripts/side_threshold.sh -- the step of mid-size contexts with and without the DBSCAN workers on the side stream
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for R in 1 2; do
for S in 768 1024 1280; do
  for C in 0 1; do
  python3 bench.py --scenes $S --chain-side-stream $C --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --steps 100 --warmup 20 > gpurun_out/st.json 2> gpurun_out/st.err
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/st.json").read().strip().splitlines()[-1])
    print("S=$S css=$C", d["ms_per_step"], d.get("side_workers"), d.get("step_kernels"), {k: round(v["avg_ms"]*1e3,1) for k,v in d["kernels"].items()}, d.get("parity",{}).get("bit_equal_vs_oracle") if d.get("parity") else None)
except Exception as e:
    print("S=$S css=$C", "ERR", e, open("gpurun_out/st.err").read()[-300:])
PY
  done
done
done
