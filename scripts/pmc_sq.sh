# SQ counter passes for the tracker kernels (issue / stall / LDS counters per kernel; diagnosis only):
#   bash scripts/pmc_sq.sh [sq1 sq2 ...]   (SCENES=512 for the one-workgroup step, POP=mixed for the 1 + s mod 8 population)
# The program follows `--` directly (no env / bash -c hop); counters in their own runs, with --kernel-trace only.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  rm -rf gpurun_out/$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --gen-workers 1 --scenes ${SCENES:-4096} --population ${POP:-full} --steps 10 --warmup 10 --chain-side-stream 2 > $GRAFT_REPO_ROOT/gpurun_out/$name.log 2>&1)
}
PASSES=${@:-sq1 sq2 sq3 sq4 sq5}
want() { [[ " $PASSES " == *" $1 "* ]]; }
want sq1 && run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
want sq2 && run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_ANY
want sq3 && run sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
want sq4 && run sq4 VALUBusy SALUBusy VALUUtilization LDSBankConflict
want sq5 && run sq5 MeanOccupancyPerCU ALUStalledByLDS MemUnitBusy MemUnitStalled
python - <<'PY'
import csv, glob, collections
for name in ("sq1","sq2","sq3","sq4","sq5"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(f"gpurun_out/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            k=r["Kernel_Name"].split("(")[0].split("::")[-1]
            if not (k.startswith("k_") or "k_" in k): continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,d in acc.items():
        print(name,k,{c: round(sum(v[len(v)//2:])/max(len(v[len(v)//2:]),1)) for c,v in d.items()}, "launches", len(next(iter(d.values()))))
PY
