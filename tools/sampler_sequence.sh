cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_samp -o s -- python tools/sampler_bench.py 10 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_samp/s_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last batch: find last k_init occurrence
idx=[i for i,r in enumerate(rows) if 'k_init' in r['Kernel_Name']]
i0=idx[-1]
t0=int(rows[i0]['Start_Timestamp'])
prev=t0
for r in rows[i0:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    print('+%7.1f gap %5.1f dur %6.1f grid %8s %s'%((s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,r.get('Grid_Size_X',r.get('Grid_Size')),n))
    prev=e
PY
rm -rf gpurun_out/prof_samp
