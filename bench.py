#!/usr/bin/env python
"""bench.py -- KGWAS hot path on MI355X: full fast-mode KG minibatch training.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one training step of kgwas/kgwas.py:129-151 on one 512-seed batch per GPU: device-side
2-hop full-neighbourhood sampling, feature slicing, 3 feature MLPs, 2 fused attention-aggregate
layers, read-out, LD-weighted MSE, backward, Adam -- on the workload BASELINE.json's metric is quoted
on (configs[1]): SynthKG-fast (784 256 SNPs / 20 032 genes / ~20.6 M directed edges, features
20 / 5120 / 128), causal-simulation-like labels, seed 1.  Inputs (graph, features, labels) are resident
in HBM before the timed region starts.

Metric: edges aggregated per second = sum over timed steps, layers and live relations of the edges the
aggregate kernels actually gather, over wall time (max over ranks); N > 1 is weak scaling (every rank
trains on its own 512-seed batches, one flat gradient all-reduce per step over RCCL).
Extra keys: ``roofline`` (layer-1 forward aggregate kernel: algorithmic bytes / HIP-event time vs the
8 TB/s HBM peak), ``cpu_baseline`` (the CPU oracle = op-for-op PyG restatement, timed on this box's host
cores on a bounded sample of the same batches), ``breakdown`` (per-kernel event times).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable by a float4 copy)


def algorithmic_bytes_fwd(n_edges, z_rows):
    """B_f = E*(4C+8) + N_d*(4C+8), C = 128  ->  520 B per edge + 520 B per (row, relation)."""
    return 520 * n_edges + 520 * z_rows


def algorithmic_bytes_bwd(n_edges, z_rows, n_src):
    """B_b = E*(8C+24) + (N_d + N_s)*(4C+8)."""
    return 1048 * n_edges + 520 * (z_rows + n_src)


def cpu_baseline(data, batch_size, budget_s=25.0, max_steps=2):
    """Reference PyG CPU path, restated (oracle/): numpy full-neighbour sampler + x[n_id] slicing + unpruned
    2-layer HeteroGNN forward/backward + Adam, all host cores.  Bounded sample: as many steps as fit in
    ~budget_s (at least 1 after 1 warm-up)."""
    from oracle.gat_oracle import HeteroGNNOracle, weighted_mse
    from oracle.sampler_np import FullNeighborSamplerNP
    # more threads than ~16 only add OpenMP fork/join cost on these small scatter ops (256 threads: 355 s/step
    # vs 13 s/step with 8); use what actually helps and report it
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    g = data.data
    t0 = time.time()
    smp = FullNeighborSamplerNP(g.edge_index_dict, g.num_nodes_dict, 2)
    torch.manual_seed(1)
    model = HeteroGNNOracle(g.edge_types, 128, 1, 2, 'GAT', 'sum', data.snp_init_dim_size, data.gene_init_dim_size,
                            data.go_init_dim_size, 1)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=5e-4)
    setup_s = time.time() - t0
    ids = np.asarray(data.train_input_nodes[1])
    y_all = g['SNP'].y
    w_all = torch.zeros(g['SNP'].x.shape[0], dtype=torch.float64)
    w_all[torch.from_numpy(np.asarray(data.all_ids))] = torch.from_numpy(np.asarray(data.ldsc_weight))
    times, edges = [], []
    for step in range(max_steps + 1):
        seeds = ids[step * batch_size:(step + 1) * batch_size]
        t = time.time()
        n_id, ei = smp.sample('SNP', seeds)
        x = {k: g[k].x[v] for k, v in n_id.items()}
        opt.zero_grad()
        out = model(x, ei, batch_size)
        loss = weighted_mse(out, y_all[n_id['SNP'][:batch_size]], w_all[n_id['SNP'][:batch_size]])
        loss.backward()
        opt.step()
        dt = time.time() - t
        if step > 0:
            times.append(dt)
            edges.append(2 * sum(int(v.shape[1]) for v in ei.values()))      # both layers touch every sampled edge
        if step > 0 and sum(times) + dt > budget_s:
            break
    tot_t = sum(times)
    return {'value': sum(edges) / tot_t, 'unit': 'edges/s', 'cores': cores, 'kind': 'port',
            's_per_step': tot_t / len(times),
            'sample': f'{len(times)} training steps (after 1 warm-up) of batch {batch_size} on the same SynthKG-fast '
                      f'batches; unpruned (both layers over every sampled edge) like PyG; torch threads={cores}'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--scale', type=float, default=1.0, help='SynthKG scale (1.0 = reference size)')
    ap.add_argument('--batch-size', type=int, default=512)
    ap.add_argument('--snp-scale', type=float, default=1.0,
                    help='multiply the SNP count and the SNP->Gene edges only (12.75 = the ~10 M-SNP full-cohort case of BASELINE.json configs[3]); not the headline workload')
    ap.add_argument('--mode', default='fast', choices=['fast', 'full'],
                    help="feature widths: 'fast' 20/5120/128 (BASELINE.json configs[1], the default) or 'full' 70/57742/128 (configs[4])")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--eager', action='store_true', help='issue every launch from the host instead of replaying one HIP graph per step')
    args = ap.parse_args()

    # stdout carries EXACTLY one JSON line: everything else that writes to file descriptor 1 (native libraries included --
    # a collective backend announcing its peers, a BLAS tuner) goes to stderr until the line is printed
    sys.stdout.flush()
    _stdout_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}')
    assert torch.cuda.is_available(), 'bench.py needs a ROCm GPU'
    local_rank %= torch.cuda.device_count()      # (single-GPU boxes: KGW_DIST_BACKEND=gloo lets 2 ranks share cuda:0 for a dry run)
    torch.cuda.set_device(local_rank)
    dev = f'cuda:{local_rank}'
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('KGW_DIST_BACKEND', 'nccl')          # "nccl" == RCCL over xGMI on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)

    from kgwas_amd import dist as kdist
    from kgwas_amd import ops
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.kgwas_data import KGWAS_Data
    from kgwas_amd.sampler import NeighborLoader

    t0 = time.time()
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):        # stdout carries exactly one JSON line
        data = KGWAS_Data.from_synthetic(scale=args.scale, seed=1, mode=args.mode, gwas_kind='causal',
                                         data_path=f'/tmp/kgwas_bench_{rank}', snp_scale=args.snp_scale)
    run = KGWAS(data, device=dev, seed=1)
    run.initialize_model()
    if world > 1:
        kdist.broadcast_params(run.model)
    ld_w = run._ld_weight_vector()
    bs = args.batch_size
    need = (args.steps + args.warmup) * bs
    ids = np.asarray(data.train_input_nodes[1])
    # weak scaling: rank r trains on batches r, r+world, ... of the reference's fixed batch order
    nb = len(ids) // bs
    mine = ids[:nb * bs].reshape(nb, bs)[rank::world].reshape(-1)
    if len(mine) < need:
        mine = np.resize(mine, need)
    mine = mine[:need]
    run.model.train()

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.eager:
        opt = torch.optim.Adam(run.model.parameters(), lr=1e-4, weight_decay=5e-4)
        it = iter(NeighborLoader(data.data, [-1, -1], ('SNP', mine), batch_size=bs, drop_last=True, device=dev))
        stats_e = [0, 0]

        def do_step(i):
            batch = next(it)
            run.train_step(batch, opt, ld_w, world)
            stats_e[0] += sum(batch.n_edges_per_layer)
            stats_e[1] += 2 * batch.n_edges_sampled
        mode = 'eager launches'
    else:
        from kgwas_amd.graph_step import GraphTrainStep
        gs = GraphTrainStep(run, ('SNP', mine), bs, lr=1e-4, weight_decay=5e-4)

        def do_step(i):
            gs.step(i)
        mode = ('HIP graphs: step graph (fwd + bwd' + (' + Adam)' if gs.capture_optimizer else '), RCCL all-reduce + Adam eager') +
                (' with the next batch sampled by a second graph on a side stream' if gs.twin else ', sampling inside it'))
    setup_s = time.time() - t0

    for i in range(args.warmup):
        do_step(i)
    if not args.eager:
        torch.cuda.synchronize()
        gs.stats.zero_()
    else:
        stats_e[0] = stats_e[1] = 0
    sync()
    t_start = time.perf_counter()
    for i in range(args.steps):
        do_step(args.warmup + i)
    sync()
    elapsed = time.perf_counter() - t_start
    seeds = args.steps * bs
    if args.eager:
        edges_kernel, edges_ref = stats_e
    else:
        st = gs.check()                       # raises if a batch overflowed the static layout
        edges_kernel, edges_ref = sum(st[:2]), 2 * st[2]

    # kernel-level timing for the roofline: HIP events around the aggregate launches on their stream, in an
    # eager pass over the same batches right after the timed region (events cannot bracket nodes of a replayed graph)
    if not args.no_kernel_timing:
        opt_r = torch.optim.Adam(run.model.parameters(), lr=1e-4, weight_decay=5e-4)
        n_r = min(args.steps, 20)
        it_r = iter(NeighborLoader(data.data, [-1, -1], ('SNP', mine[args.warmup * bs:(args.warmup + n_r) * bs]),
                                   batch_size=bs, drop_last=True, device=dev))
        ops.TIMER.enabled = True
        for _ in range(n_r):
            run.train_step(next(it_r), opt_r, ld_w, world)
        torch.cuda.synchronize()
        ops.TIMER.enabled = False

    stats = torch.tensor([elapsed, float(edges_kernel), float(edges_ref), float(seeds)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats[:1].clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        tot = stats[1:].clone()
        torch.distributed.all_reduce(tot, op=torch.distributed.ReduceOp.SUM)
        elapsed = float(tmax[0]); edges_kernel, edges_ref, seeds = (float(x) for x in tot)

    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0:
        return
    summ = ops.TIMER.summary()
    breakdown, roof = {}, None
    for (tag, layer), d in sorted(summ.items()):
        byts = algorithmic_bytes_fwd(d['edges'], d['z_rows']) if tag == 'fwd' else None
        entry = {'launches': d['n'], 'avg_ms': d['ms'] / d['n'], 'edges_per_launch': d['edges'] / d['n']}
        if tag == 'fwd':
            entry['algorithmic_GBs'] = byts / (d['ms'] * 1e-3) / 1e9
        elif tag == 'bwd_dst':
            entry['algorithmic_GBs'] = (528 * d['edges'] + 1028 * d['z_rows']) / (d['ms'] * 1e-3) / 1e9
        elif tag == 'bwd_src':
            entry['algorithmic_GBs'] = (528 * d['edges'] + 520 * d['n_src']) / (d['ms'] * 1e-3) / 1e9
        breakdown[f'agg_{tag}_l{layer}'] = entry
    if ('fwd', 1) in summ:
        d = summ[('fwd', 1)]
        ach = algorithmic_bytes_fwd(d['edges'], d['z_rows']) / (d['ms'] * 1e-3) / 1e9
        roof = {'kernel': 'k_agg_fwd (layer-1 attention aggregate, forward)',
                'timing': 'hipEvent pair recorded by the C ABI immediately around the k_agg_fwd launch on its stream (KgwLayerArgs.ev_before/ev_after), eager pass over the timed batches',
                'bound': 'hbm', 'achieved': ach,
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                'bytes_per_launch': algorithmic_bytes_fwd(d['edges'], d['z_rows']) / d['n'],
                'avg_launch_ms': d['ms'] / d['n'], 'traffic': None}
        # rocprofv3 --stats averages ALL launches of the kernel name (layer 1: ~1 M edges, layer 2: ~2 k edges) together:
        # the figure to compare with profiles/*kernel_stats*.csv
        allf = [v for (tag, _), v in summ.items() if tag == 'fwd']
        roof['avg_ms_all_launches_of_this_kernel_name'] = sum(v['ms'] for v in allf) / max(1, sum(v['n'] for v in allf))
        pmc = os.path.join(ROOT, 'profiles', 'pmc_agg_fwd.json')
        if os.path.exists(pmc):
            try:
                roof['traffic'] = json.load(open(pmc)).get('hbm_bytes_per_launch')
            except Exception:
                pass
        if roof['traffic']:
            # `achieved` follows the contract (ALGORITHMIC bytes / launch time) and can exceed the peak: a source row gathered
            # by several relations / destination rows is counted every time but fetched from HBM once.  The HBM-side rate
            # of the same launch, from the PMC traffic:
            roof['hbm_side_GBs'] = roof['traffic'] / (roof['avg_launch_ms'] * 1e-3) / 1e9
            roof['hbm_side_frac'] = roof['hbm_side_GBs'] / HBM_PEAK_GBS
            roof['note'] = ('achieved = algorithmic bytes (SURVEY 8d: 520 B per edge and per segment) / HIP-event time; the PMC '
                            'counters see less HBM traffic than that (traffic, hbm_side_*): re-used source rows are served by L2 / '
                            'Infinity Cache')
    cpu = None
    if not args.no_cpu_baseline and world == 1:          # the CPU leg is timed at N = 1 only
        cpu = cpu_baseline(data, bs)
    ms = elapsed / args.steps * 1e3
    out = {
        'metric': 'edges aggregated/sec (full fast-mode KG minibatch training; epoch time in config)',
        'value': edges_kernel / elapsed, 'unit': 'edges/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': ('SynthKG-fast full KG (784256 SNP / 20032 Gene / ~20.6M directed edges; features '
                                '20/5120/128) + causal-simulation GWAS seed=1, batch 512 seeds per GPU, 2-layer GAT-128, '
                                'Adam(1e-4, wd 5e-4) -- BASELINE.json configs[1]') if (args.mode == 'fast' and args.snp_scale == 1.0 and args.scale == 1.0) else
                               (f'SynthKG-fast with scale={args.scale}, snp_scale={args.snp_scale} (see config.graph) -- NOT the headline configuration; '
                                'snp_scale 12.75 is the ~10 M-SNP full-cohort case of BASELINE.json configs[3] on ONE GPU') if args.mode == 'fast' else
                               ('SynthKG-full: same graph, full-mode feature widths 70/57742/128 -- BASELINE.json configs[4] '
                                'on one GPU, not the headline configuration'),
                   'scale': args.scale, 'snp_scale': args.snp_scale,
                   'graph': {'nodes': {t: int(data.data[t].x.shape[0]) for t in data.data.node_types},
                             'directed_edges': int(sum(data.data[et].edge_index.shape[1] for et in data.data.edge_types))},
                   'batch_size_per_gpu': bs, 'parallelism': f'seed-dp{world}', 'execution': mode,
                   'edges_per_step_kernel': edges_kernel / args.steps / world,
                   'edges_per_step_reference_equivalent': edges_ref / args.steps / world,
                   'reference_equivalent_edges_per_s': edges_ref / elapsed,
                   'seeds_per_s': seeds / elapsed,
                   'epoch_time_s_956_steps': 956 * ms / 1e3 / world,
                   'setup_s': setup_s},
        'roofline': roof, 'cpu_baseline': cpu, 'breakdown': breakdown,
    }
    sys.stdout.flush()
    os.dup2(_stdout_fd, 1)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
