# usage (GPU box): bash scripts/pmc_conv16.sh   -> counters of k_mars_conv16 from two rocprofv3 --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_conv_a $R/gpurun_out/pmc_conv_b
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $R/gpurun_out/pmc_conv_a -- python3 $R/scripts/ubench_conv16.py > $R/gpurun_out/pmc_conv_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_conv_b -- python3 $R/scripts/ubench_conv16.py > $R/gpurun_out/pmc_conv_b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_conv_a", "pmc_conv_b"):
    acc = collections.defaultdict(float); n = collections.Counter()
    for p in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(p)):
            if "conv16" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: print(d, k, acc[k] / max(n[k], 1), "per dispatch over", n[k])
PY
tail -3 $R/gpurun_out/pmc_conv_a.log
find $R/gpurun_out/pmc_conv_a $R/gpurun_out/pmc_conv_b -name "*.csv" -size +5M -delete
