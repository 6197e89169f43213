#!/usr/bin/env python
"""Predicted multi-GPU step time = measured per-rank compute + modelled xGMI time of the rank's collectives.

Inputs: a directory of bench.py lines, one JSON file each (any names; what a line is is read from the line itself):
    python bench.py [--mode full]             (P = 1, the plain single-GPU step: the baseline of its feature-width mode)
    python bench.py --as-rank 0/P ...         (ONE GPU doing rank 0's work of a P-GPU job, every collective a no-op of the right
                                              size: ms_per_step = the rank's compute time,
                                              config.communication.collectives_per_step_and_rank = what it would have sent)
Output: a markdown table on stdout (committed as profiles/r4/r4_scale_model.md, quoted in DESIGN.md section 7).

The alpha-beta model of a collective over the xGMI mesh of one MI355X node (every GPU has a direct link to each of the 7
others, ~153 GB/s per link and direction -- SURVEY.md 8e; a message rarely reaches more than ~75 % of a link):
    direct   all-reduce  = reduce-scatter + all-gather, every rank exchanging S/P with each peer over its own link:
                           2 a + 2 S / (P b)             (what RCCL's one-shot / direct algorithms do for a few MB)
             all-gather / reduce-scatter of S bytes in total:  a + S / (P b)
    ring     all-reduce  = 2 (P - 1) steps of S/P over ONE link: 2 (P - 1) a_h + 2 (P - 1) / P * S / b
             all-gather / reduce-scatter:  (P - 1) a_h + (P - 1) / P * S / b
with a = 12 us (launch + flag handshake of one phase), a_h = 3 us per ring step, b = 0.75 * 153 GB/s.  Both are printed: the
truth for 4 - 35 MB messages on 8 GPUs lies between them, and the node measurement is the driver's (SCALE_rNN.json), not this
table's.  What is exposed: seed-parallel -- the first gradient bucket is all-reduced on a side stream under the second half of
the backward (its time counts only beyond the ~0.4 ms that half takes), everything else sits between graphs on the critical
path; SNP-sharded -- every collective sits between two graph segments (+ one host round trip of ~8 us each)."""
import glob
import json
import os
import re
import sys

A_PHASE, A_HOP, BETA = 12e-6, 3e-6, 0.75 * 153e9
SECOND_HALF_S = 0.40e-3           # graph B of the split backward (feature MLPs' backward incl. the gene dW product)
CUT_S = 8e-6                      # host round trip of a collective between two graph segments


def t_collective(name, nbytes, P, algo):
    if P <= 1:
        return 0.0
    S = float(nbytes)
    if name.startswith('all_reduce'):
        return (2 * A_PHASE + 2 * S / (P * BETA)) if algo == 'direct' else (2 * (P - 1) * A_HOP + 2 * (P - 1) / P * S / BETA)
    # all_gather: bytes = the gathered total; reduce_scatter: bytes = the scattered total
    return (A_PHASE + S / (P * BETA)) if algo == 'direct' else ((P - 1) * A_HOP + (P - 1) / P * S / BETA)


def load(path):
    s = open(path).read().strip()
    return json.loads(s.splitlines()[-1]) if s else None


def main(d):
    lines = [(f, load(f)) for f in sorted(glob.glob(os.path.join(d, '*.json')))]
    lines = [(f, j) for f, j in lines if j is not None]
    mode_of = lambda j: 'full' if 'SynthKG-full' in j['config']['workload'] else 'fast'
    base = {mode_of(j): j for f, j in lines if j['config'].get('emulated_rank') is None and j['n_gpus'] == 1
            and j['config']['communication']['world_size'] == 1}
    print('| feature widths | scaling | parallelism | P | rank compute (measured, ms) | bytes handed to collectives per step and rank (MB) | '
          'predicted step, direct / ring (ms) | predicted speed-up over 1 GPU, direct / ring |')
    print('|---|---|---|---|---|---|---|---|')
    for mode, p1 in sorted(base.items()):
        t1 = p1['ms_per_step'] * 1e-3
        print(f'| {mode} | - | single GPU | 1 | {t1 * 1e3:.3f} | - | {t1 * 1e3:.3f} | 1.00 |')
        rows = []
        for f, j in lines:
            er = j['config'].get('emulated_rank')
            if er is None or er['rank'] != 0 or mode_of(j) != mode:
                continue
            P = er['world']
            scaling = j['scaling']
            par = 'shard' if j['config']['parallelism'].startswith('snp-shard') else 'seed'
            comp = j['ms_per_step'] * 1e-3
            coll = j['config']['communication']['collectives_per_step_and_rank']
            pred = {}
            for algo in ('direct', 'ring'):
                exposed = 0.0
                for name, v in coll.items():
                    calls = max(1, round(v['calls_per_step']))
                    per_call = v['bytes_per_step'] / v['calls_per_step']
                    t = calls * t_collective(name, per_call, P, algo)
                    if par == 'seed' and 'under the MLPs' in name:
                        t = max(0.0, t - SECOND_HALF_S)                  # hidden under graph B
                    elif par == 'shard':
                        t += calls * CUT_S
                    exposed += t
                pred[algo] = comp + exposed
            split = any('first gene layer' in k for k in coll)
            rows.append((scaling != 'weak', par, P, split, comp, coll, pred, scaling))
        for _, par, P, split, comp, coll, pred, scaling in sorted(rows, key=lambda r: r[:4]):
            mb = sum(v['bytes_per_step'] / v['calls_per_step'] * max(1, round(v['calls_per_step'])) for v in coll.values()) / 1e6
            # weak: P batches of 512 seeds per step (edges/s = P x per-rank rate); strong: one 512-seed batch per step
            sp = [(P * t1 / pred[a]) if scaling == 'weak' else (t1 / pred[a]) for a in ('direct', 'ring')]
            print(f'| {mode} | {scaling} | {par}{" + gene layer split" if split else ""} | {P} | {comp * 1e3:.3f} | {mb:.1f} | '
                  f'{pred["direct"] * 1e3:.3f} / {pred["ring"] * 1e3:.3f} | {sp[0]:.2f} / {sp[1]:.2f} |')
    print()
    print('Speed-up: weak = P x 512 seeds per step against 512 (whole-job edges/s, what `python bench.py --gpus P` reports by default); '
          'strong = steps per second on ONE 512-seed batch per step (what KGWAS.train does).  "single GPU" rows: the plain step of '
          'that box (`python bench.py`).')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/r4a')
