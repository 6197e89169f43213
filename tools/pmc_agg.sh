#!/bin/bash
# HBM traffic of the aggregate kernels from the TCC counters (separate --pmc passes, MI355X_MICROARCH.md HBM).
# usage (via gpurun, repo root): bash tools/pmc_agg.sh
out=gpurun_out/pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-.}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o $c -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $out/$c.json 2> $out/$c.err
done
ls $out | head -20
python - "$out" <<'PY'
import csv, sys, collections, json
out = sys.argv[1]
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = out + '/' + c + '_counter_collection.csv'
    rows = list(csv.DictReader(open(f)))
    if not rows:
        print('no rows', f); continue
    print(c, 'columns', list(rows[0].keys()))
    agg = collections.defaultdict(list)
    for r in rows:
        if r.get('Counter_Name') != c:
            continue
        agg[r['Kernel_Name'].replace('(anonymous namespace)::', '')[:40]].append(float(r['Counter_Value']))
    for k, v in agg.items():
        if any(t in k for t in ('k_agg', 'k_gather', 'k_tn_gemm')):
            v.sort()
            print('  %-42s n %3d  max %14.1f  median %14.1f' % (k, len(v), v[-1], v[len(v) // 2]))
            res.setdefault(k, {})[c] = {'max': v[-1], 'median': v[len(v) // 2], 'n': len(v)}
json.dump(res, open(out + '/pmc_summary.json', 'w'), indent=1)
PY
