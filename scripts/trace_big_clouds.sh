cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_bigc
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bigc -- python3 $GRAFT_REPO_ROOT/scripts/time_big_clouds.py > $GRAFT_REPO_ROOT/gpurun_out/prof_bigc.log 2>&1)
tail -4 gpurun_out/prof_bigc.log
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_bigc/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'mmw::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=max(i for i,r in enumerate(rows) if 'k_reset' in r['Kernel_Name'])
t0=int(rows[last]['Start_Timestamp'])
for r in rows[last:]:
    n=r['Kernel_Name'].split('(')[0].replace('void mmw::','').replace('mmw::','')
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  {n[:50]}")
PY
