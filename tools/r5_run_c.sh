cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_hub.py -x -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r5c_hub.txt
timeout 1200 python -m pytest tests/test_gpu_aggregate.py tests/test_gpu_model.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py tests/test_gpu_shard.py -x -q 2>&1 | tail -5 > gpurun_out/r5c_more.txt
FAST="--no-cpu-baseline --no-pmc --no-in-step --no-epoch --no-kernel-timing"
timeout 600 python bench.py --steps 200 $FAST > gpurun_out/r5c_bench.json 2> gpurun_out/r5c_bench.err
KGW_LIB_PATH=/root/repo/kgwas_amd/csrc/libkgwas_hip_r4.so timeout 600 python bench.py --steps 200 $FAST > gpurun_out/r5c_bench_r4lib.json 2> gpurun_out/r5c_bench_r4lib.err
timeout 600 python bench.py --steps 200 $FAST > gpurun_out/r5c_bench2.json 2> gpurun_out/r5c_bench2.err
cat gpurun_out/r5c_hub.txt gpurun_out/r5c_more.txt
