#!/usr/bin/env python
"""Time the device-side sampler ALONE (no training step beside it): consecutive 512-seed batches of the benchmark graph into a
static-capacity buffer, HIP events around every kgw_sample_batch call.  usage: python tools/sampler_bench.py [n_batches]"""
import contextlib
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgwas_amd.kgwas_data import KGWAS_Data
from kgwas_amd.sampler import BatchBuffers, NeighborLoader, sample_into

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
with contextlib.redirect_stdout(sys.stderr):
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_bench_0')
ids = np.asarray(data.train_input_nodes[1])[:512 * (n + 8)]
probe = NeighborLoader(data.data, [-1, -1], ('SNP', ids), batch_size=512, drop_last=True, device='cuda:0', prefetch=False)
dg = probe.dg.with_static_caps(probe.measure_caps(1.03))
for grid in [int(g) for g in os.environ.get('KGW_SB_GRIDS', '256,0').split(',')]:      # (0: the whole-GPU default)
    buf = BatchBuffers(dg, grid)
    seeds = torch.zeros(512, dtype=torch.int64, device='cuda:0')
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n + 4):
        seeds.copy_(probe.ids[i * 512:(i + 1) * 512])
        if i >= 4:
            ev[i - 4][0].record()
        sample_into(dg, buf, seeds, probe.seed_type, record=False)
        if i >= 4:
            ev[i - 4][1].record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
    err = int(buf.read_meta().error)
    print(f'sampler alone, eager launches, grid {grid or 2048} blocks: {t.mean():.1f} us per 512-seed batch (min {t.min():.1f}, max {t.max():.1f}), error mask {err}')
    # the same call captured once and replayed (what the training loop does on its side stream): no host launch cost in between
    g = torch.cuda.CUDAGraph()
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        sample_into(dg, buf, seeds, probe.seed_type, record=False)
    torch.cuda.current_stream().wait_stream(s_)
    with torch.cuda.graph(g):
        sample_into(dg, buf, seeds, probe.seed_type, record=False)
    for i in range(n + 4):
        seeds.copy_(probe.ids[i * 512:(i + 1) * 512])
        if i >= 4:
            ev[i - 4][0].record()
        g.replay()
        if i >= 4:
            ev[i - 4][1].record()
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ev]) * 1e3
    print(f'sampler alone, captured graph,  grid {grid or 2048} blocks: {t.mean():.1f} us per 512-seed batch (min {t.min():.1f}, max {t.max():.1f})')
