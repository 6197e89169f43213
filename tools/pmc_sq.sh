#!/bin/bash
# SQ issue / wait breakdown of the aggregate kernels (one --pmc pass, kernel trace only).  usage (via gpurun): bash tools/pmc_sq.sh
out=gpurun_out/pmc_sq
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-.}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $out -o sq -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-pmc --no-epoch > $out/sq.json 2> $out/sq.err
python - "$out" <<'PY'
import csv, sys, collections
out = sys.argv[1]
rows = list(csv.DictReader(open(out + '/sq_counter_collection.csv')))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:34]
    if any(t in k for t in ('k_agg', 'k_linear_wreg', 'k_tn_gemm', 'k_mlp2', 'k_param_tail', 'k_adam', 'k_transform', 'k_linear_splitk', 'k_g3')):
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    # the largest dispatch of each kernel (layer 1)
    i = max(range(len(d['SQ_WAVE_CYCLES'])), key=lambda j: d['SQ_WAVE_CYCLES'][j])
    wc = d['SQ_WAVE_CYCLES'][i]
    print('%-36s waves %8.0f  wave_cycles %.3g  wait_any %4.1f%%  wait_inst %4.1f%%  active_inst %4.1f%%  VALU/wave %7.0f  SALU/wave %7.0f  LDS/wave %6.0f' % (
        k, d['SQ_WAVES'][i], wc, 100 * d['SQ_WAIT_ANY'][i] / wc, 100 * d['SQ_WAIT_INST_ANY'][i] / wc, 100 * d['SQ_ACTIVE_INST_ANY'][i] / wc,
        d['SQ_INSTS_VALU'][i] / d['SQ_WAVES'][i], d['SQ_INSTS_SALU'][i] / d['SQ_WAVES'][i], d['SQ_INSTS_LDS'][i] / d['SQ_WAVES'][i]))
PY
