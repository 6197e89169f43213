mkdir -p gpurun_out/prof1
timeout 900 python -m pytest tests -m gpu -q --timeout 200 --tb=short 2>&1 | grep -E "^(E  |FAILED|[0-9]+ (passed|failed))" | cut -c1-300
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1 -o r1 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof1/bench.json 2> gpurun_out/prof1/err.log
cat gpurun_out/prof1/bench.json | cut -c1-1500
f=$(find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-220
