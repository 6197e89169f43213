#!/bin/bash
# The sampler's structural tests, its stand-alone timing (tools/sampler_bench.py) and a rocprofv3 kernel trace of it with the
# median duration of every kernel per grid size.  usage (via gpurun, from the repo root): bash tools/profile_sampler.sh
timeout 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_graph.py tests/test_gpu_golden.py -q -m gpu -x 2>&1 | tail -2
timeout 300 python tools/sampler_bench.py 2>/dev/null | tail -4
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_samp -o s -- python tools/sampler_bench.py 10 > /dev/null 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("gpurun_out/prof_samp/s_kernel_trace.csv")))
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    d[(n, r.get('Grid_Size_X', r.get('Grid_Size')))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()):
    if len(v) > 5: print(k, len(v), 'med %.1f max %.1f'%(sorted(v)[len(v)//2], max(v)))
PY
