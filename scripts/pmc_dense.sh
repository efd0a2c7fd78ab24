# L2 behaviour of k_mars_dense1_w192 (diagnostic): hits / misses / requests, and the requests that leave the L2 for the fabric
# (Infinity Cache / HBM), per launch.  usage: bash scripts/pmc_dense.sh [rows]   (default: the e2e leg's K = T batch, 31744)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
ROWS=${1:-31744}
for PASS in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  D=gpurun_out/pmc_dense_$(echo $PASS | cut -d' ' -f1)
  rm -rf $D
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $PASS --output-format csv -d $GRAFT_REPO_ROOT/$D -- python3 $GRAFT_REPO_ROOT/scripts/bench_dense1.py $ROWS > $GRAFT_REPO_ROOT/$D.log 2>&1)
  tail -2 $D.log
done
python3 - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob('gpurun_out/pmc_dense_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out={}
for k,d in acc.items():
    if 'dense1' in k or 'Cijk' in k or 'conv16' in k:
        out[k]={c: round(sum(v)/len(v)/1e6,3) for c,v in d.items()}
        out[k]['launches']=len(next(iter(d.values())))
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/pmc_dense_summary.json','w'), indent=1)
PY
find gpurun_out/pmc_dense_* -name "*.csv" -size +20M -delete
