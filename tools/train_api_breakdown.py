#!/usr/bin/env python
"""Where does the wall clock of KGWAS.train() go?  The API path a user of the reference runs (kgwas/kgwas.py:85-212) on the
benchmark workload -- full-size fast-mode synthetic graph, causal-simulation labels, batch 512 -- with a stopwatch around its
phases: loaders + resident graph, capture of the training step, every epoch's steps, every validation pass (the first one builds
its captured forward), best-model snapshots, test pass, whole-genome inference (1 061 batches), p-value post-processing + CSV.
usage: python tools/train_api_breakdown.py [epochs]   (prints a table and one JSON line)"""
import copy
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgwas_amd import graph_step as gsm
from kgwas_amd import kgwas as kmod
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T = {}
ORDER = []


def clock(name, fn):
    def wrapped(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            torch.cuda.synchronize()
            if name not in T:
                ORDER.append(name)
            T[name] = T.get(name, 0.0) + time.perf_counter() - t
    return wrapped


data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_synth_full_e2e')
run = KGWAS(data, device='cuda:0', seed=1, exp_name='e2e')
run.initialize_model()

run.make_loaders = clock('loaders + resident graph (CSR build, H2D)', run.make_loaders)
_gs_init = gsm.GraphTrainStep.__init__
gsm.GraphTrainStep.__init__ = clock('training step: capacity pass + warm-up + capture', _gs_init)
_ge_init = gsm.GraphEvalStep.__init__
gsm.GraphEvalStep.__init__ = clock('eval loaders: capacity passes + captures (val, test, inference)', _ge_init)
_ge_run = gsm.GraphEvalStep.run
gsm.GraphEvalStep.run = clock('eval forward passes (val x epochs, test, inference)', _ge_run)
kmod.deepcopy = clock('best-model snapshots (deepcopy) + lr_uni copy', copy.deepcopy)
run._postprocess = clock('p-value post-processing + CSV', run._postprocess)
_step = gsm.GraphTrainStep.step
n_steps = [0]
t_steps = [0.0]
_t_epoch0 = [None]


def step(self, i):
    if _t_epoch0[0] is None:
        torch.cuda.synchronize()
        _t_epoch0[0] = time.perf_counter()
    n_steps[0] += 1
    return _step(self, i)


gsm.GraphTrainStep.step = step
_check = gsm.GraphTrainStep.check


def check(self):
    r = _check(self)
    if _t_epoch0[0] is not None and n_steps[0] % self.n_batches == 0:
        t_steps[0] += time.perf_counter() - _t_epoch0[0]
        _t_epoch0[0] = None
    return r


gsm.GraphTrainStep.check = check

torch.cuda.synchronize()
t0 = time.perf_counter()
run.train(batch_size=512, num_workers=0, lr=1e-4, weight_decay=5e-4, epoch=epochs, save_best_model=False)
torch.cuda.synchronize()
total = time.perf_counter() - t0
T_all = dict(T)
T_all['training steps (%d x %d)' % (epochs, n_steps[0] // max(epochs, 1))] = t_steps[0]
named = sum(T_all.values())
print(f'KGWAS.train(epoch={epochs}) on the benchmark workload: {total:.2f} s wall')
for k in ['training steps (%d x %d)' % (epochs, n_steps[0] // max(epochs, 1))] + ORDER:
    print(f'  {T_all[k]:8.3f} s  {100 * T_all[k] / total:5.1f} %  {k}')
print(f'  {total - named:8.3f} s  {100 * (total - named) / total:5.1f} %  everything else (metrics, prints, Python between the phases)')
print(f'non-step share: {100 * (1 - t_steps[0] / total):.1f} %; ms per step inside train(): {1e3 * t_steps[0] / max(n_steps[0], 1):.4f}; '
      f'val {run.val_metrics}; test {run.test_metrics}')
print(json.dumps({'epochs': epochs, 'total_s': total, 'steps_s': t_steps[0], 'non_step_share': 1 - t_steps[0] / total,
                  'phases_s': {k: T_all[k] for k in T_all}}))
