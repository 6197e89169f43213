#!/bin/bash
# Step time of the default bench against the side sampler's launch geometry (gpurun, repo root):
#   SGS="256 1024" SHIFTS="8 11 12" bash tools/sampler_grid_sweep.sh     (SHIFTS: KGW_TS_MIN_SHIFT, log2 of the sort's rows per bucket;
#   the key-block count was swept with a temporary knob in round 6 -- 128 / 256 / 512: 1.035 / 1.034 / 1.048 ms -- and stays 256)
for sg in ${SGS:-256 512 1024}; do
 for sh in ${SHIFTS:-11}; do
  KGW_TS_MIN_SHIFT=$sh KGW_SIDE_SAMPLER_GRID=$sg python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pmc --no-epoch --no-kernel-timing > gpurun_out/r6_sg_$sg.json 2> gpurun_out/r6_sg_$sg.err
  python - $sg $sh <<'PY'
import json, sys
sg, sh = sys.argv[1:3]
try:
    d = json.loads(open('gpurun_out/r6_sg_%s.json' % sg).read().strip().splitlines()[-1])
    o = d['config'].get('sampler_overlap') or {}
    print('SG %5s shift %2s  ms/step %.4f   with %.4f  alone %.4f  sampler alone %.4f' % (sg, sh, d['ms_per_step'], o.get('step_with_side_sampler_ms', 0), o.get('step_alone_ms', 0), o.get('sampler_alone_ms', 0)))
except Exception as e:
    print('SG', sg, sh, 'failed', e)
PY
 done
done
