# LDS behaviour of k_mars_conv16 (diagnostic): bank-conflict cycles against all LDS-array cycles, per launch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_conv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_conv -- python3 $GRAFT_REPO_ROOT/scripts/bench_dense1.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_conv.log 2>&1)
tail -3 gpurun_out/pmc_conv.log
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob('gpurun_out/pmc_conv/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    if 'conv16' in k or 'dense1' in k:
        print(k, {c: round(sum(v)/len(v)/1e6,3) for c,v in d.items()}, 'launches', len(next(iter(d.values()))))
PY
