#!/usr/bin/env python
"""Mean duration of the kgw_gemm3 kernels per grid size from a rocprofv3 --kernel-trace csv (timing experiments on
kgwas_amd/csrc/kgw_gemm3.hip).  usage: python tools/g3_times.py <kernel_trace.csv> [label]"""
import collections
import csv
import sys

d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'k_g3_' in n:
        d[(n.split('(')[0][-24:], r.get('Grid_Size', r.get('Grid_Size_X')))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
for k, v in sorted(d.items()):
    v = v[3:] or v
    print(sys.argv[2] if len(sys.argv) > 2 else '', k, 'n', len(v), 'mean %.1f min %.1f' % (sum(v) / len(v), min(v)))
