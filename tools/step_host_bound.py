"""Is the captured step's pace set by the host (graph launches, event calls) or by the GPU?  Host time spent inside step() per
call against the wall time per step (round 3: 0.10 ms against 1.21 ms -- the GPU sets the pace)."""
import os, sys, time, contextlib
sys.path.insert(0, '.')
import numpy as np, torch
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data
from kgwas_amd.graph_step import GraphTrainStep
with contextlib.redirect_stdout(sys.stderr):
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_bench_0')
ids = np.asarray(data.train_input_nodes[1])[:512 * 120]
for br in ('-',):
    run = KGWAS(data, device='cuda:0', seed=1); run.initialize_model(); run.model.train()
    gs = GraphTrainStep(run, ('SNP', ids), 512)
    for i in range(10):
        gs.step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10, 110):
        gs.step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host time in step() %.3f ms per call, wall %.3f ms per step' % ((t1 - t0) / 100 * 1e3, (t2 - t0) / 100 * 1e3))
    gs.check()
