#!/bin/bash
# Clock / power of the GPU while k_g3_gemm runs back to back ("power-bound?"): the two 5120-wide gene products
# of the benchmark in a loop for ~4 s, rocm-smi sampled every 100 ms beside it; idle samples before and after for reference.
# usage (repo root, via gpurun):  bash tools/g3_power_trace.sh [out_file]
out=${1:-gpurun_out/g3_power_trace.txt}
mkdir -p $(dirname $out)
python - <<'PY' &
import time, torch, sys
sys.path.insert(0, '.')
from kgwas_amd import ops
M, K = 20032, 5120
X = torch.rand(M, K, device='cuda'); Xt = X.t().contiguous()
W = torch.randn(128, K, device='cuda') / K ** 0.5; dz = torch.randn(M, 128, device='cuda')
pw = ops.gemm3_pack(W, K, False); pd = ops.gemm3_pack(dz, M, True)
time.sleep(1.5)                                     # idle samples
t0 = time.time(); n = 0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
while time.time() - t0 < 4.0:
    for _ in range(50):
        ops.gemm3(X, pw); ops.gemm3(Xt, pd, transpose_out=True); n += 2
    torch.cuda.synchronize()
ev1.record(); torch.cuda.synchronize()
print('GEMM3_LOOP products %d avg_us %.1f' % (n, ev0.elapsed_time(ev1) * 1e3 / n), flush=True)
time.sleep(1.0)
PY
pid=$!
: > $out
for i in $(seq 1 70); do
  echo "t=$(date +%s.%N)" >> $out
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" >> $out
  sleep 0.1
done
wait $pid >> $out 2>&1
grep -E "GEMM3_LOOP" $out
python - "$out" <<'PY'
import re, sys
txt = open(sys.argv[1]).read().split('t=')[1:]
rows = []
for blk in txt:
    s = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', blk); p = re.search(r'Power \(W\): ([0-9.]+)', blk)
    if s or p:
        rows.append((float(blk.split()[0]), int(s.group(1)) if s else -1, float(p.group(1)) if p else -1))
if rows:
    t0 = rows[0][0]
    print('time_s sclk_MHz power_W')
    for t, s, p in rows[::3]:
        print('%.1f %d %.0f' % (t - t0, s, p))
    busy = [r for r in rows if r[2] > 0.6 * max(x[2] for x in rows)]
    print('samples %d; under load: sclk %d..%d MHz, power %.0f..%.0f W; idle power %.0f W' % (
        len(rows), min(r[1] for r in busy), max(r[1] for r in busy), min(r[2] for r in busy), max(r[2] for r in busy), min(r[2] for r in rows)))
PY
