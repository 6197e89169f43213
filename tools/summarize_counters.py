#!/usr/bin/env python
"""Per-kernel means of the rocprofv3 counter CSVs tools/agg_counters.sh leaves (g<i>_counter_collection.csv): the layer-1
launch (= the largest grid) of each kernel.  usage: python tools/summarize_counters.py <dir> [kernel-name prefix ...]"""
import collections
import csv
import glob
import os
import sys


def main():
    base = sys.argv[1]
    kernels = sys.argv[2:] or ['k_agg_fwd<false', 'k_agg_bwd_dst', 'k_agg_bwd_src']
    res = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(base, 'g*_counter_collection.csv')), key=lambda p: int(os.path.basename(p)[1:].split('_')[0])):
        rows = list(csv.DictReader(open(f)))
        for kern in kernels:
            agg = collections.defaultdict(list)
            for r in rows:
                n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
                if n.startswith(kern):
                    agg[r['Counter_Name']].append((float(r['Counter_Value']), int(r['Grid_Size'])))
            for c, v in agg.items():
                g = max(x[1] for x in v)
                vals = [x[0] for x in v if x[1] == g]
                res.setdefault(kern, collections.OrderedDict())[c] = (sum(vals) / len(vals), len(vals))
    if not res:
        sys.exit('no counter files under ' + base)
    names = list(res)
    ctrs = []
    for n in names:
        ctrs += [c for c in res[n] if c not in ctrs]
    print('rocprofv3 --pmc <group> --kernel-trace -- python tools/agg_layer_runner.py  (one process per group; batch 512, benchmark')
    print('graph; per kernel the launches of its largest grid = layer 1; mean over the dispatches counted in the last line)')
    print()
    print('%-34s' % 'counter' + ''.join('%22s' % n[:21] for n in names))
    for c in ctrs:
        print('%-34s' % c + ''.join(('%22.0f' % res[n][c][0]) if c in res[n] else '%22s' % '-' for n in names))
    print()
    print('dispatches averaged: ' + ', '.join('%s %d' % (n, next(iter(res[n].values()))[1]) for n in names))


if __name__ == '__main__':
    main()
