#!/bin/bash
# Round-end artefacts on a GPU box (via gpurun, from the repo root): the GPU suite in both GEMM routings, smoke(), the default bench
# line (with counter passes, in-step trace, CPU baseline), a 200-step line, and a kernel trace + one-step timeline of the replayed
# step.  usage: bash tools/final_artifacts.sh <tag>     -> gpurun_out/<tag>_*
tag=${1:-final}
o=gpurun_out
mkdir -p $o
{
  echo "default routing:"; timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3
  echo "KGW_STRICT=1:"; KGW_STRICT=1 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
} > $o/${tag}_gputests.txt 2>&1
python bench.py > $o/${tag}_bench_default.json 2> $o/${tag}_bench_default.err
cp gpurun_out/bench_pmc/summary.json $o/${tag}_pmc_summary.json 2>/dev/null
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pmc --no-epoch > $o/${tag}_bench_200steps.json 2> $o/${tag}_bench_200steps.err
bash tools/profile_bench.sh $tag > $o/${tag}_per_step_summary.txt 2>&1
python tools/step_timeline.py $o/prof_$tag > $o/${tag}_step_timeline.txt 2>&1
cp $o/prof_$tag/r_kernel_stats.csv $o/${tag}_kernel_stats_graph_bench.csv 2>/dev/null
tail -4 $o/${tag}_gputests.txt
