# usage: bash scripts/trace_kpost.sh S...  -- per-launch durations of k_post / k_track over the timed steps (kernel trace)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for S in ${@:-4096}; do
  rm -rf gpurun_out/prof_kp$S
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_kp$S -- python3 $GRAFT_REPO_ROOT/bench.py --scenes $S --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --gen-workers 1 --steps 40 --warmup 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_kp$S.log 2>&1)
  python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_kp$S/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'mmw::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].split('(')[0].replace('void mmw::','').replace('mmw::','').split('<')[0]
tr=[r for r in rows if nm(r)=='k_track'][-40:]
t0=int(tr[0]['Start_Timestamp'])
sel=[r for r in rows if int(r['Start_Timestamp'])>=t0]
out=[]
for r in sel:
    n=nm(r)
    if n in ('k_post','k_chain','k_dbscan_big'):
        out.append((n,(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
print($S,'k_post us:',[round(d) for n,s,d in out if n=='k_post'])
print($S,'k_dbscan_big us:',[round(d) for n,s,d in out if n=='k_dbscan_big'])
print($S,'k_chain us:',[round(d) for n,s,d in out if n=='k_chain'])
PY
done
