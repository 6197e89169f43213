#!/bin/bash
# A/B of two builds on one box: the default bench alternating between kgwas_amd/csrc/libkgwas_hip_prev.so (KGW_LIB_PATH) and the
# current library.  usage (gpurun, repo root): bash tools/ab_lib.sh [rounds] [extra bench args]
n=${1:-3}; shift
for i in $(seq $n); do
  for which in prev cur; do
    if [ $which = prev ]; then export KGW_LIB_PATH=$PWD/kgwas_amd/csrc/libkgwas_hip_prev.so; else unset KGW_LIB_PATH; fi
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pmc --no-epoch --no-kernel-timing "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['config']['sampler_overlap']
print('$which  ms/step %.4f   alone %.4f  sampler alone %.4f' % (d['ms_per_step'], o['step_alone_ms'], o['sampler_alone_ms']))"
  done
done
