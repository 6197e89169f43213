#!/usr/bin/env python
"""KGWAS.train end to end on the benchmark workload (full-size fast-mode synthetic graph, causal-simulation labels, batch 512) for a
few epochs -- the API path a user of the reference runs (kgwas/kgwas.py:85-212), wall clock per epoch and the validation Pearson r
after each.  usage: python tools/train_two_epochs.py [epochs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_synth_full_e2e')
run = KGWAS(data, device='cuda:0', seed=1, exp_name='e2e')
run.initialize_model()
torch.cuda.synchronize()
t = time.perf_counter()
run.train(batch_size=512, num_workers=0, lr=1e-4, weight_decay=5e-4, epoch=epochs, save_best_model=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f'{epochs} epochs + test + inference + post-processing: {dt:.2f} s wall; val {run.val_metrics}; test {run.test_metrics}')
