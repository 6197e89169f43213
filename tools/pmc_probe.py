#!/usr/bin/env python
"""Workload of bench.py's counter passes: run it UNDER `rocprofv3 --pmc <counters> --kernel-trace` (bench.py does, one
pass per counter group -- MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).

Dispatch sequence, so that the per-dispatch counter rows can be attributed by kernel name + order:
  1. CALIBRATION: k_gather_rows copying `--cal-rows` rows of 128 floats with identity ids -- the aggregate kernels' own
     access pattern (one float4 per lane, 512-byte rows), a known byte count (rows * 516 read, rows * 512 written) that
     is far larger than the 256 MiB Infinity Cache, so its FETCH_SIZE / WRITE_SIZE are HBM bytes; the ratio
     known / counted is the correction applied to every other kernel of the same pass (the guide's gfx950 x2 fetch
     correction, re-measured in the run instead of assumed);
  2. `--steps` eager training steps (forward + backward, no optimizer) at batch 512 on the benchmark graph;
  3. `--steps` more at `--big-batch` seeds per batch: the sampled working set of one launch then exceeds the
     Infinity Cache, so FETCH_SIZE of THOSE launches is HBM traffic rather than an upper bound on it.
Writes <out>.json: the order of the k_agg_fwd / k_agg_bwd_* dispatches with their edge / row counts."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--batch-size', type=int, default=512)
    ap.add_argument('--big-batch', type=int, default=4096)
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--snp-scale', type=float, default=1.0)
    ap.add_argument('--mode', default='fast')
    ap.add_argument('--cal-rows', type=int, default=2 * 1024 * 1024)
    ap.add_argument('--skip-steps', type=int, default=5, help='first batch = batch index skip-steps of the training order')
    args = ap.parse_args()

    from kgwas_amd import ops
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.kgwas_data import KGWAS_Data
    from kgwas_amd.sampler import NeighborLoader, gather_rows
    dev = 'cuda:0'
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        data = KGWAS_Data.from_synthetic(scale=args.scale, seed=1, mode=args.mode, gwas_kind='causal',
                                         data_path='/tmp/kgwas_bench_0', snp_scale=args.snp_scale)
    run = KGWAS(data, device=dev, seed=1)
    run.initialize_model()
    run.model.train()
    ld_w = run._ld_weight_vector()
    ids = np.asarray(data.train_input_nodes[1])

    # warm everything that allocates / tunes BEFORE the calibration kernel, so the dispatch order below is clean
    def steps(bs, n, first):
        seeds = ids[first * bs:(first + n) * bs]
        if len(seeds) < n * bs:
            seeds = np.resize(ids, n * bs)
        recs = []
        for batch in NeighborLoader(data.data, [-1, -1], ('SNP', seeds), batch_size=bs, drop_last=True, device=dev):
            for p in run.model.parameters():
                p.grad = None
            loss, _ = run.model.forward_loss(batch.x_dict, batch.edge_index_dict, bs, batch.n_id('SNP'), batch.dg.y['SNP'], ld_w)
            loss.backward()
            m, NT = batch.meta, batch.dg.schema.NT
            recs.append({'batch_size': bs,
                         'layers': [{'layer': l + 1, 'edges': int(m.n_edges[l]), 'z_rows': int(m.z_base[l][NT]),
                                     'n_src': int(m.src_base[l][NT]), 'chunks': int(m.n_chunks[l])} for l in range(2)]})
        torch.cuda.synchronize()
        return recs

    steps(args.batch_size, 1, 0)                       # warm-up (dispatches before the calibration marker are ignored)
    n = args.cal_rows
    src = torch.rand(n, 128, device=dev)
    idt = torch.arange(n, dtype=torch.int32, device=dev)
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    gather_rows(src, idt, dst)                          # the LARGEST k_gather_rows dispatch of the run = the calibration
    torch.cuda.synchronize()
    del src, dst, idt
    out = {'calibration': {'kernel': 'k_gather_rows', 'rows': n, 'read_bytes': n * 516, 'write_bytes': n * 512},
           'steps': steps(args.batch_size, args.steps, args.skip_steps)}
    if args.big_batch:
        out['steps'] += steps(args.big_batch, args.steps, 1)
    json.dump(out, open(args.out, 'w'))


if __name__ == '__main__':
    main()
