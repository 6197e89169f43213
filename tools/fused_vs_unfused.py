#!/usr/bin/env python
"""Full-size check of the launch merges of rounds 4-5: N captured training steps on the benchmark workload with every merge on (fused
optimiser launch + operand image, merged transform backward, grouped short products, the parameter-only forward work as rider blocks of the
gene product, the parameter-only end of the backward pass inside the grouped products' launch) against the same steps with all of them off --
losses and every parameter must be IDENTICAL bit for bit.  usage: python tools/fused_vs_unfused.py [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgwas_amd import ops
from kgwas_amd.graph_step import GraphTrainStep
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_synth_full_fvu')
ids = np.asarray(data.train_input_nodes[1])
out = []
for on in (True, False):
    ops._FUSED_ADAM = ops._MERGED_TRANSFORM_BWD = ops._DEFER_PRODUCTS = ops._DUV_PIECES = ops._G3_RIDERS = ops._PARAM_TAIL = ops._DEFER_REDUCE = ops._DEFER_READOUT_FOLD = ops._PACK_FUSED = on
    run = KGWAS(data, device='cuda:0', seed=1)
    run.initialize_model()
    init = lambda v: not isinstance(v, torch.nn.parameter.UninitializedParameter)      # (lazy PyG-style placeholders: never used)
    if out:
        run.model.load_state_dict(out[0][2], strict=False)
    sd0 = {k: v.clone() for k, v in run.model.state_dict().items() if init(v)}
    gs = GraphTrainStep(run, ('SNP', ids), 512, lr=1e-4, weight_decay=5e-4)
    assert gs.fused_adam == on
    losses = [gs.step(i % gs.n_batches) for i in range(n)]
    gs.check()
    losses = [float(x) for x in losses[-5:]] + [float(losses[0])]
    out.append((losses, {k: v.clone() for k, v in run.model.state_dict().items() if init(v)}, sd0, gs.describe()))
    del gs
same_loss = out[0][0] == out[1][0]
diff = [k for k in out[0][1] if not torch.equal(out[0][1][k], out[1][1][k])]
moved = sum(float((out[0][1][k].double() - out[0][2][k].double()).abs().sum()) for k in out[0][1] if out[0][1][k].dtype.is_floating_point)
print(f'{n} steps, batch 512, full-size fast-mode graph: losses identical {same_loss} (last {out[0][0][-2]:.9g}), '
      f'{len(out[0][1]) - len(diff)} of {len(out[0][1])} state tensors bit-identical, sum |parameter change| {moved:.4g}')
print('merged :', out[0][3])
print('legacy :', out[1][3])
sys.exit(0 if same_loss and not diff else 1)
