# usage (GPU box): bash scripts/trace_chain_start.sh [scenes] -- when the side stream's k_chain of a step becomes resident relative to the step's
# k_predict / k_track / k_post / k_dbscan_big (kernel trace, no counters: the streams run concurrently)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
S=${1:-4096}
rm -rf gpurun_out/prof_cs
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cs -- python3 $GRAFT_REPO_ROOT/bench.py --scenes $S --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --gen-workers 1 --steps 40 --warmup 10 --chain-side-stream 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_cs.log 2>&1)
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_cs/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'mmw::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].split('(')[0].replace('void mmw::','').replace('mmw::','').split('<')[0]
pred=[i for i,r in enumerate(rows) if nm(r)=='k_predict'][-12:]
for a,b in zip(pred[:-1],pred[1:]):
    t0=int(rows[a]['Start_Timestamp'])
    print(' | '.join(f"{nm(r)} {((int(r['Start_Timestamp'])-t0)/1e3):.1f}..{((int(r['End_Timestamp'])-t0)/1e3):.1f}" for r in rows[a:b]))
PY
