cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_hub.py -x -q -s -k "captured_training" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r5d_traj.txt
timeout 1200 python -m pytest tests/test_gpu_shard.py -x -q 2>&1 | tail -40 > gpurun_out/r5d_shard.txt
FAST="--no-cpu-baseline --no-pmc --no-in-step --no-epoch --no-kernel-timing"
timeout 600 python bench.py --steps 200 $FAST > gpurun_out/r5d_bench.json 2> gpurun_out/r5d_bench.err
KGW_PARAM_BRANCH=1 timeout 600 python bench.py --steps 200 $FAST > gpurun_out/r5d_bench_branch.json 2> gpurun_out/r5d_bench_branch.err
timeout 600 python bench.py --steps 200 $FAST > gpurun_out/r5d_bench2.json 2> gpurun_out/r5d_bench2.err
KGW_PARAM_BRANCH=1 timeout 600 python bench.py --steps 200 $FAST > gpurun_out/r5d_bench_branch2.json 2> gpurun_out/r5d_bench_branch2.err
cat gpurun_out/r5d_traj.txt gpurun_out/r5d_shard.txt
