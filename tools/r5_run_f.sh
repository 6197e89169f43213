cd /root/repo
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r5f_tests.txt
KGW_SRC_XCD=0 timeout 300 python tools/agg_layer_runner.py > gpurun_out/r5f_agg_xcd0.txt 2>/dev/null
KGW_SRC_XCD=1 timeout 300 python tools/agg_layer_runner.py > gpurun_out/r5f_agg_xcd1.txt 2>/dev/null
KGW_SRC_XCD=0 timeout 300 python tools/agg_layer_runner.py >> gpurun_out/r5f_agg_xcd0.txt 2>/dev/null
KGW_SRC_XCD=1 timeout 300 python tools/agg_layer_runner.py >> gpurun_out/r5f_agg_xcd1.txt 2>/dev/null
cat gpurun_out/r5f_tests.txt gpurun_out/r5f_agg_xcd0.txt gpurun_out/r5f_agg_xcd1.txt
