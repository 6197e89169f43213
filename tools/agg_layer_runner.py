#!/usr/bin/env python
"""Twelve eager training batches of the benchmark graph (forward_loss + backward, batch 512) with HIP events around the three
attention-aggregate kernels of layer 1 -- the process tools/agg_counters.sh profiles (one rocprofv3 --pmc group per run).
(Until round 4 this was tools/xcd_experiment.py: it also dealt the dst-major kernels' chunks to the XCDs by source-row range --
FETCH_SIZE -29 %, the launch slower, DESIGN.md section 5 -- through a KgwLayerArgs.chunk_perm hook that round 5 removed.)
usage: python tools/agg_layer_runner.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from kgwas_amd import ops
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.kgwas_data import KGWAS_Data
    from kgwas_amd.sampler import NeighborLoader
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, gwas_kind='causal', data_path='/tmp/kgwas_bench_0')
    run = KGWAS(data, device='cuda:0', seed=1)
    run.initialize_model()
    run.model.train()
    ld_w = run._ld_weight_vector()
    ids = np.asarray(data.train_input_nodes[1])[:512 * 12]
    ops.TIMER.enabled = False
    for i, batch in enumerate(NeighborLoader(data.data, [-1, -1], ('SNP', ids), batch_size=512, drop_last=True, device='cuda:0', prefetch=False)):
        ops.TIMER.enabled = i >= 2
        for q in run.model.parameters():
            q.grad = None
        loss, _ = run.model.forward_loss(batch.x_dict, batch.edge_index_dict, 512, batch.n_id('SNP'), batch.dg.y['SNP'], ld_w)
        loss.backward()
    torch.cuda.synchronize()
    for (tag, layer), d in sorted(ops.TIMER.summary().items()):
        if layer == 1:
            print(f'{tag:8s} layer {layer}: {d["ms"] / d["n"] * 1e3:7.1f} us  ({d["n"]} launches, loss {float(loss):.6f})')


if __name__ == '__main__':
    main()
