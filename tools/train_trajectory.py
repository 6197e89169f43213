#!/usr/bin/env python
"""Train N captured steps on the benchmark graph (BASELINE.json configs[1], seed 1) and dump the parameters + the per-step loss:
the arithmetic routes are compared by running it under different switches (KGW_GEMM3=0 KGW_MLP2_SPLIT=0: fp32 matrix pipe
throughout; KGW_SHORT_ROWS=0: another summation order in one aggregate kernel) -- profiles/r2/r2_l_trajectory_compare.txt.
usage (via gpurun, repo root):  [switches] python tools/train_trajectory.py <out.pt> [steps]"""
import sys, os, numpy as np, torch, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data
from kgwas_amd.graph_step import GraphTrainStep
out = sys.argv[1]
with contextlib.redirect_stdout(sys.stderr):
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_traj')
run = KGWAS(data, device='cuda:0', seed=1)
run.initialize_model()
ids = np.asarray(data.train_input_nodes[1])
gs = GraphTrainStep(run, ('SNP', ids[:512 * 320]), 512, lr=1e-4, weight_decay=5e-4)
losses = []
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    l = gs.step(i)
    losses.append(l.detach().clone())
torch.cuda.synchronize()
sd = {k: v.detach().float().cpu() for k, v in run.model.named_parameters() if not isinstance(v, torch.nn.parameter.UninitializedParameter)}
sd['__losses__'] = torch.stack(losses).double().cpu()
torch.save(sd, out)
print('saved', len(sd))
