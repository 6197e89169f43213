#!/usr/bin/env python
"""Experiment: does dealing the chunks of k_agg_fwd / k_agg_bwd_dst to the XCDs BY SOURCE ROW RANGE cut the L2-miss
re-fetch?  Builds the permutation on the host (numpy) for a few benchmark batches and times layer-1 forward + backward
with and without it (HIP events around the kernels); run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` to see bytes.
usage: python tools/xcd_experiment.py [mode]   mode: none | type | range"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_perm(batch, layer, mode):
    m, dg = batch.meta, batch.dg
    sc = dg.schema
    nc = int(m.n_chunks[layer - 1])
    ch = batch.buf.chunks[:nc * 8].view(nc, 8).cpu().numpy()
    e0, e1, rel = ch[:, 0], ch[:, 1], ch[:, 3]
    cost = (e1 - e0) + 8
    src_t = np.asarray(sc.src_type)[rel]
    if mode == 'type':
        key = src_t.astype(np.int64)
    else:       # 'range': source type, then the chunk's FIRST source row (sources ascend inside a segment)
        col = batch.buf.col_local.cpu().numpy()
        key = src_t.astype(np.int64) * (1 << 32) + col[e0]
    order = np.argsort(key, kind='stable')
    cum = np.cumsum(cost[order])
    cuts = np.searchsorted(cum, cum[-1] * np.arange(1, 8) / 8.0)
    pieces = np.split(order, cuts)
    nmax = max(len(p) for p in pieces)
    plen = 32 * ((nmax + 3) // 4)
    perm = np.full(plen, -1, dtype=np.int32)
    for x, p in enumerate(pieces):
        k = np.arange(len(p))
        perm[(k // 4) * 32 + x * 4 + k % 4] = p
    dev = batch.buf.chunks.device
    return torch.from_numpy(perm).to(dev), torch.tensor([plen], dtype=torch.int32, device=dev), [len(p) for p in pieces]


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'none'
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
        if mode != 'none':
            p, n, sizes = build_perm(batch, 1, mode)
            batch.chunk_perm = {1: (p, n)}
            if i == 0:
                print('pieces', sizes, 'perm len', int(n), 'chunks', int(batch.meta.n_chunks[0]), file=sys.stderr)
        ops.TIMER.enabled = i >= 2
        for q in run.model.parameters():
            q.grad = None
        loss, _ = run.model.forward_loss(batch.x_dict, batch.edge_index_dict, 512, batch.n_id('SNP'), batch.dg.y['SNP'], ld_w)
        loss.backward()
    torch.cuda.synchronize()
    for (tag, layer), d in sorted(ops.TIMER.summary().items()):
        if layer == 1:
            print(f'{mode:6s} {tag:8s} layer {layer}: {d["ms"] / d["n"] * 1e3:7.1f} us  ({d["n"]} launches, loss {float(loss):.6f})')


if __name__ == '__main__':
    main()
