#!/usr/bin/env python
"""What does a SECOND queue cost the captured step, apart from the work it carries?  The step (benchmark workload, batch 512) timed with
(a) the real side sampler, (b) nothing beside it (stale batches), (c) a side graph of N launches that do nothing (one block, one store)
-- the command processor's share -- and (d) a side graph of N launches that stream M MB each -- memory traffic without the sampler's
atomics / LDS.  usage: python tools/side_queue_cost.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgwas_amd.graph_step import GraphTrainStep
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_synth_full_sq')
ids = np.asarray(data.train_input_nodes[1])
run = KGWAS(data, device='cuda:0', seed=1)
run.initialize_model()
gs = GraphTrainStep(run, ('SNP', ids), 512, lr=1e-4, weight_decay=5e-4)
assert gs.twin and gs.fused_adam


def timed():
    for i in range(20):
        gs.step(i % gs.n_batches)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        gs.step((20 + i) % gs.n_batches)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


class Dummy:
    def __init__(self, launches, mb):
        self.g = torch.cuda.CUDAGraph()
        self.small = torch.zeros(64, device='cuda')
        self.a = torch.zeros(max(mb, 1) * 262144, device='cuda')
        self.b = torch.zeros_like(self.a)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                self.body(launches, mb)
            torch.cuda.synchronize()
            with torch.cuda.graph(self.g, stream=s):
                self.body(launches, mb)
        torch.cuda.current_stream().wait_stream(s)

    def body(self, launches, mb):
        for _ in range(launches):
            if mb:
                self.b.copy_(self.a)
            else:
                self.small.add_(1.0)

    def replay(self):
        self.g.replay()


res = {}
res['real sampler'] = timed()
real = gs.sample_graphs
gs._skip_resample = True
res['nothing beside the step'] = timed()
gs._skip_resample = False
for name, (l, mb) in {'25 empty launches': (25, 0), '50 empty launches': (50, 0), '25 launches streaming 8 MB each (r + w)': (25, 8),
                      '25 launches streaming 64 MB each (r + w)': (25, 64)}.items():
    d = Dummy(l, mb)
    gs.sample_graphs = [d, d]
    res[name] = timed()
gs.sample_graphs = real
base = res['nothing beside the step']
for k, v in res.items():
    print('%-46s %.4f ms / step  (%+5.1f us)' % (k, v, (v - base) * 1e3))
