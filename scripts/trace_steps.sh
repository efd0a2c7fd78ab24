# usage: bash scripts/trace_steps.sh [STEPS] -- per-step wall time (k_predict start to the next k_predict start) over a long window of the 4096-scene bench under
# the kernel trace, and which launch made the slow steps slow
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
K=${1:-150}
rm -rf gpurun_out/prof_steps
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_steps -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --gen-workers 1 --steps $K --warmup 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_steps.log 2>&1)
python3 - <<PY
import csv,glob
import numpy as np
f=glob.glob('gpurun_out/prof_steps/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'mmw::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].split('(')[0].replace('void mmw::','').replace('mmw::','').split('<')[0]
pred=[i for i,r in enumerate(rows) if nm(r)=='k_predict'][-($K+1):]
steps=[]
for a,b in zip(pred[:-1],pred[1:]):
    t0=int(rows[a]['Start_Timestamp']); t1=int(rows[b]['Start_Timestamp'])
    ks={}
    for r in rows[a:b]:
        ks[nm(r)]=ks.get(nm(r),0)+(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    steps.append(((t1-t0)/1e3, ks))
d=np.array([s[0] for s in steps])
print(f"{len(d)} steps: mean {d.mean():.1f} us, median {np.median(d):.1f}, p90 {np.percentile(d,90):.1f}, max {d.max():.1f}; sum over the median {np.sum(d-np.median(d)):.0f} us = {np.sum(d-np.median(d))/len(d):.1f} us per step")
for i in np.argsort(-d)[:10]:
    print(f"  step {i}: {d[i]:.1f} us", {k: round(v,1) for k,v in steps[i][1].items()})
PY
