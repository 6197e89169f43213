#!/usr/bin/env python
"""What each HALF of the side sampler costs the captured step: the step (benchmark workload, batch 512) timed with, on the side stream,
(a) the whole sampler of the next batch, (b) nothing, (c) its hop expansion only (kgw_sample_batch_parts 0 .. 2 n_hops - 1: segments,
chunks, mark / compact / relabel), (d) its last part only (layer tables + the src-major sort of both layers, on the structures the last
full sampling left).  (c) and (d) train on stale batches -- this is a timing experiment.  usage: python tools/sampler_halves_cost.py [steps]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgwas_amd import _lib
from kgwas_amd.graph_step import GraphTrainStep
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_synth_full_sq')
ids = np.asarray(data.train_input_nodes[1])
run = KGWAS(data, device='cuda:0', seed=1)
run.initialize_model()
gs = GraphTrainStep(run, ('SNP', ids), 512, lr=1e-4, weight_decay=5e-4)


def timed():
    for i in range(20):
        gs.step(i % gs.n_batches)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        gs.step((20 + i) % gs.n_batches)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


class Parts:
    """A side graph that runs parts [a, b] of the sampler into a buffer of its OWN (the step's buffers keep their last batch)."""
    def __init__(self, a, b, buf):
        self.g = torch.cuda.CUDAGraph()
        self.a, self.b, self.buf = a, b, buf
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.body(s)
            torch.cuda.synchronize()
            with torch.cuda.graph(self.g, stream=s):
                self.body(s)
        torch.cuda.current_stream().wait_stream(s)

    def body(self, s):
        rc = _lib.lib().kgw_sample_batch_parts(C.byref(gs.dg.kg), C.byref(self.buf.c), C.c_void_p(gs.seeds.data_ptr()), gs.seeds.numel(),
                                               gs.seed_type, 0, self.a, self.b, C.c_void_p(s.cuda_stream))
        _lib.check(rc, 'kgw_sample_batch_parts')

    def replay(self):
        self.g.replay()


from kgwas_amd.sampler import BatchBuffers
from kgwas_amd.graph_step import SIDE_SAMPLER_GRID
res = {}
res['whole sampler'] = timed()
real = gs.sample_graphs
gs._skip_resample = True
res['nothing beside the step'] = timed()
gs._skip_resample = False
H = gs.dg.n_hops
spare = BatchBuffers(gs.dg, SIDE_SAMPLER_GRID)
full = Parts(0, 2 * H, spare)            # (fills the spare buffer once completely, so that the last part alone has valid inputs)
full.replay(); torch.cuda.synchronize()
for name, (a, b) in {'hop expansion only (parts 0 .. 2H-1)': (0, 2 * H - 1), 'layer tables + src-major sort only (part 2H)': (2 * H, 2 * H),
                     'whole sampler into a spare buffer': (0, 2 * H)}.items():
    d = Parts(a, b, spare)
    gs.sample_graphs = [d, d]
    res[name] = timed()
gs.sample_graphs = real
base = res['nothing beside the step']
for k, v in res.items():
    print('%-50s %.4f ms / step  (%+5.1f us)' % (k, v, (v - base) * 1e3))
