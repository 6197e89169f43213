"""One steady-state training step from a rocprofv3 kernel trace: the launches between two consecutive k_adam kernels, in start
order, per hardware queue, with the idle gap before each.  usage: python tools/step_timeline.py <dir with r_kernel_trace.csv> [k]
(k: which step from the end, default 2; k = "sampled": the last step that has the side sampler's launches -- more than 10 launches
on a second queue -- beside it; bench.py ends with steps WITHOUT the sampler, its overlap check)."""
import csv, sys, glob, collections
d = sys.argv[1]
back = sys.argv[2] if len(sys.argv) > 2 else '2'
f = (glob.glob(d + '/**/*kernel_trace.csv', recursive=True) + glob.glob(d + '/*kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    r['n'] = r['n'].split('(')[0]
rows.sort(key=lambda r: r['s'])
adam = [r for r in rows if r['n'].startswith('k_accumulate_stats')] or [r for r in rows if r['n'].startswith('k_adam')]   # one per step
def window(b):
    a0_, a1_ = adam[-b - 1], adam[-b]
    return a0_, a1_, [r for r in rows if r['s'] > a0_['e'] and r['e'] <= a1_['e']]
if back == 'sampled':
    for b in range(2, len(adam) - 1):
        a0, a1, win = window(b)
        qs = collections.Counter(r.get('Queue_Id', '?') for r in win)
        if len(qs) > 1 and sorted(qs.values())[-2] > 10:
            break
else:
    a0, a1, win = window(int(back))
print('step window %.1f us, %d launches (all queues)' % ((a1['e'] - a0['e']) / 1e3, len(win)))
byq = collections.defaultdict(list)
for r in win:
    byq[r.get('Queue_Id', '?')].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r['e'] - r['s'] for r in rs)
    print('--- queue %s: %d launches, busy %.1f us' % (q, len(rs), busy / 1e3))
    prev = None
    for r in rs:
        gap = (r['s'] - prev) / 1e3 if prev else 0.0
        print('  +%8.1f  gap %6.1f  dur %7.1f  grid %-9s wg %-5s %s' % ((r['s'] - a0['e']) / 1e3, gap, (r['e'] - r['s']) / 1e3,
              r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')), r['n'][:70]))
        prev = r['e']
