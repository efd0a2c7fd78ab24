# L2 behaviour of k_mars_dense1 (diagnostic): hits / misses / requests per launch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_dense
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_dense -- python3 $GRAFT_REPO_ROOT/scripts/bench_dense1.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_dense.log 2>&1)
tail -3 gpurun_out/pmc_dense.log
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob('gpurun_out/pmc_dense/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in acc.items():
    if 'dense1' in k or 'Cijk' in k:
        print(k, {c: round(sum(v)/len(v)/1e6,2) for c,v in d.items()}, 'launches', len(next(iter(d.values()))))
PY
