#!/bin/bash
# Step time of the default bench against the side sampler's launch geometry (gpurun, repo root):
#   SGS="256 1024" NBLKS="0 512" bash tools/sampler_grid_sweep.sh     (NBLKS: KGW_TS_NBLK, 0 = the library's rule)
for sg in ${SGS:-256 512 1024}; do
 for nb in ${NBLKS:-0}; do
 for sh in ${SHIFTS:-8}; do
  KGW_TS_MIN_SHIFT=$sh KGW_TS_NBLK=$nb KGW_SIDE_SAMPLER_GRID=$sg python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pmc --no-epoch --no-kernel-timing > gpurun_out/r6_sg_$sg.json 2> gpurun_out/r6_sg_$sg.err
  python - $sg $nb $sh <<'PY'
import json, sys
sg, nb, sh = sys.argv[1:4]
try:
    d = json.loads(open('gpurun_out/r6_sg_%s.json' % sg).read().strip().splitlines()[-1])
    o = d['config'].get('sampler_overlap') or {}
    print('SG %5s nblk %4s shift %2s  ms/step %.4f   with %.4f  alone %.4f  sampler alone %.4f' % (sg, nb, sh, d['ms_per_step'], o.get('step_with_side_sampler_ms', 0), o.get('step_alone_ms', 0), o.get('sampler_alone_ms', 0)))
except Exception as e:
    print('SG', sg, nb, 'failed', e)
PY
 done
 done
done
