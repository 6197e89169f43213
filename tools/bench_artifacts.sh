#!/bin/bash
# The bench half of tools/final_artifacts.sh (no test suite): default line, 200-step line, kernel trace + one-step timeline.
# usage (via gpurun, from the repo root): bash tools/bench_artifacts.sh <tag>   -> gpurun_out/<tag>_*
tag=$1; o=gpurun_out
python bench.py > $o/${tag}_bench_default.json 2> $o/${tag}_bench_default.err
cp gpurun_out/bench_pmc/summary.json $o/${tag}_pmc_summary.json 2>/dev/null
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pmc --no-epoch > $o/${tag}_bench_200steps.json 2> $o/${tag}_bench_200steps.err
bash tools/profile_bench.sh $tag > $o/${tag}_per_step_summary.txt 2>&1
python tools/step_timeline.py $o/prof_$tag > $o/${tag}_step_timeline.txt 2>&1
cp $o/prof_$tag/r_kernel_stats.csv $o/${tag}_kernel_stats_graph_bench.csv 2>/dev/null
python -c "
import json
for f in ('${tag}_bench_default','${tag}_bench_200steps'):
    d=json.loads(open('gpurun_out/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'])
"
