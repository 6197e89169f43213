#!/bin/bash
# rocprofv3 kernel trace of the default bench on a GPU box; prints the per-step kernel-time summary.
# usage (from the repo root, via gpurun):  bash tools/profile_bench.sh <tag> [bench args]
tag=${1:-run}; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-.}
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc --no-epoch --no-kernel-timing "$@" > $out/bench.json 2> $out/err.log
python - "$out" <<'PY'
import csv, sys, json
out = sys.argv[1]
rows = list(csv.DictReader(open(out + '/r_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
ncalls = sum(int(r['Calls']) for r in rows)
# forward/backward passes executed: timed + warm-up + capture warm-up steps AND the eager passes of the roofline leg
# (those run no optimizer step, so k_adam shows fewer calls)
nstep = max([int(r['Calls']) for r in rows if 'k_readout_wmse_train(' in r['Name'] or 'k_readout_wmse_bwd(' in r['Name']] + [0]) or 23   # (no eager leg: warm-ups + capture + timed)
print('kernel time total %.1f ms, %d launches; %d forward/backward passes executed (timed, warm-ups, roofline leg; the caps pre-pass adds sampler-only work)' % (tot / 1e6, ncalls, nstep))
# kernels of the step itself: launched at least once per optimizer step (leaves out the one-off library tuning runs)
step_rows = [r for r in rows if (int(r['Calls']) >= nstep or 'k_adam' in r['Name']) and 'flush_icache' not in r['Name']
             and not (r['Name'].startswith('Cijk_') and int(r['Calls']) % nstep)]
SAMP = ('k_fill_i32', 'k_init', 'k_hop_', 'k_seg_deg', 'k_scan_', 'k_fill_chunks', 'k_mark', 'k_count_pending', 'k_assign', 'k_relabel', 'k_layer_tables', 'k_t_', 'k_ts_', 'k_meta_to_host', 'k_segments_copy')
is_s = lambda r: any(t in r['Name'] for t in SAMP)
print('kernels launched every step: %.2f ms / step; of which sampler (incl. the caps pre-pass share) %.2f, %d + %d launches / step' % (sum(float(r['TotalDurationNs']) for r in step_rows) / 1e6 / nstep, sum(float(r['TotalDurationNs']) for r in step_rows if is_s(r)) / 1e6 / nstep, sum(int(r['Calls']) for r in step_rows if not is_s(r)) / nstep, sum(int(r['Calls']) for r in step_rows if is_s(r)) / nstep))
for r in step_rows[:60]:
    print('%-60s calls/step %6.1f  us/step %8.1f  avg %8.1f us %5.1f%%' % (r['Name'].replace('(anonymous namespace)::', '')[:60], int(r['Calls']) / nstep, float(r['TotalDurationNs']) / 1e3 / nstep, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
try:
    d = json.loads(open(out + '/bench.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('bench under profiler: ms/step %.3f value %.4g' % (d['ms_per_step'], d['value']) + (' roofline frac %.3f' % r['frac'] if 'frac' in r else ' (no kernel timing in this run: --no-kernel-timing)'))
except Exception as e:
    print('no bench json', e)
PY
