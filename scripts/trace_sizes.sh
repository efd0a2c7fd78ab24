# usage: bash scripts/trace_sizes.sh S  -- rocprofv3 kernel trace of the tracker-only bench at S scenes: per-kernel averages
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for S in ${@:-512}; do
  rm -rf gpurun_out/prof_s$S
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_s$S -- python3 $GRAFT_REPO_ROOT/bench.py --scenes $S --no-cpu --no-e2e --no-e2e-parity --no-cold --no-shards --no-full --no-ingest --no-single --gen-workers 1 --steps 40 --warmup 10 > $GRAFT_REPO_ROOT/gpurun_out/prof_s$S.log 2>&1)
  python - <<PY
import csv,glob,statistics
f=glob.glob('gpurun_out/prof_s$S/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'mmw::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
by={}
for r in rows[-160:]:
    n=r['Kernel_Name'].split('(')[0].replace('void mmw::','').replace('mmw::','')
    by.setdefault(n,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print($S, {k:(round(statistics.median(v),1), round(min(v),1), len(v)) for k,v in by.items()})
ks=[r for r in rows if 'k_track' in r['Kernel_Name']][-20:]
print('  step period us (median of k_track start deltas):', round(statistics.median([(int(b['Start_Timestamp'])-int(a['Start_Timestamp']))/1e3 for a,b in zip(ks,ks[1:])]),1))
PY
done
