cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_hub.py -x -q -s -k "captured_training" 2>&1 | grep "captured trajectory\|passed\|failed\|Error" | cut -c1-700 > gpurun_out/r5e_traj.txt
timeout 1200 python -m pytest tests/test_gpu_aggregate.py tests/test_gpu_fused_adam.py tests/test_gpu_dist.py tests/test_gpu_golden.py tests/test_gpu_model.py -x -q 2>&1 | tail -8 > gpurun_out/r5e_tests.txt
FAST="--no-cpu-baseline --no-pmc --no-in-step --no-epoch"
for i in 1 2; do
KGW_SRC_XCD=0 timeout 600 python bench.py --steps 200 $FAST > gpurun_out/r5e_bench_xcd0_$i.json 2> gpurun_out/r5e_bench_xcd0_$i.err
KGW_SRC_XCD=1 timeout 600 python bench.py --steps 200 $FAST > gpurun_out/r5e_bench_xcd1_$i.json 2> gpurun_out/r5e_bench_xcd1_$i.err
done
cat gpurun_out/r5e_traj.txt gpurun_out/r5e_tests.txt
