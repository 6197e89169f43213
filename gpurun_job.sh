timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py tests/test_gpu_dense.py -m gpu -q --timeout 600 --tb=short 2>&1 | grep -E "^(E  |FAILED|[0-9]+ (passed|failed))" | cut -c1-300
python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
