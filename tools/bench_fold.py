#!/usr/bin/env python
"""kgw_fold_fwd / kgw_fold_bwd at the benchmark's layer-1 shape (23 relations of 29, 3 MLPs): run under
`rocprofv3 --kernel-trace --stats` and read the average durations (KGW_FOLD_PARTS masks the backward's block classes)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgwas_amd import ops
from kgwas_amd.model import RelationPack
NR, C = 29, 128
edge_types = [(f's{r}', f'r{r}', f'd{r}') for r in range(NR)]
rel_ids = list(range(17, 23)) + list(range(0, 17))
pack = RelationPack(edge_types, rel_ids, C).cuda()
sm = np.array([1] * 6 + [0] * 6 + [1] * 5 + [2] * 6, dtype=np.int32)
dm = np.array([0] * 6 + [1] * 17, dtype=np.int32)
tab = (np.asarray(rel_ids, dtype=np.int32), sm, dm)
fc = []
for m in range(3):
    fc += [torch.randn(C, C).cuda().requires_grad_(True), torch.randn(C).cuda().requires_grad_(True)]
U = torch.randn(NR, C).cuda().requires_grad_(True); V = torch.randn(NR, C).cuda().requires_grad_(True)
outs = ops.fold_fc_output_hip(pack, U, V, fc, tab)
gs = [torch.randn_like(o) for o in outs]
def fb():
    o = ops.fold_fc_output_hip(pack, U, V, fc, tab)
    torch.autograd.backward(o, gs)
for _ in range(50): fb()
torch.cuda.synchronize()
print('done; read the kernel durations from rocprofv3 --kernel-trace --stats')
