# usage (GPU box): bash scripts/pmc_dense_sq.sh [rows] -> issue / wait / LDS counters of the Dense-1 tile kernel (scripts/bench_dense1.py), one
# rocprofv3 --pmc pass per group; gpurun_out/pmc_dense_sq.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROWS=${1:-31744}
: > $R/gpurun_out/pmc_dense_sq.txt
(rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | grep -E "LDS|WAIT|ACTIVE_INST|INST_CYCLES|BUSY|INSTS_(LDS|VMEM|FLAT|SALU|VALU$|MFMA|VALU_MFMA)|LEVEL" | tr '\n' ' ') > $R/gpurun_out/pmc_dense_sq_available.txt
i=0
for PASS in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
            "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
            "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES"; do
  i=$((i+1)); D=$R/gpurun_out/pmc_dense_sq_$i; rm -rf $D
  timeout 300 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $D -- python3 $R/scripts/bench_dense1.py $ROWS > $D.log 2>&1
  python3 - <<PY >> $R/gpurun_out/pmc_dense_sq.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for p in glob.glob("$D/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0]
        if "dense1" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    for c in acc[k]: print(k[-30:], c, round(acc[k][c] / max(n[k][c], 1), 1), "per launch over", n[k][c])
PY
  tail -1 $D.log >> $R/gpurun_out/pmc_dense_sq.txt
  find $D -name "*.csv" -size +5M -delete
done
cat $R/gpurun_out/pmc_dense_sq.txt
