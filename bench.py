#!/usr/bin/env python
"""bench.py -- KGWAS hot path on MI355X: full fast-mode KG minibatch training.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ... --
     or plainly: without WORLD_SIZE in the environment bench.py starts its N ranks itself, one per GPU over RCCL; on a box with
     fewer than N GPUs the ranks share devices over gloo -- a dry run of the code path, flagged in the line, not a measurement)
    python bench.py --as-rank R/P [--scaling ... --parallelism ...]
     ONE GPU doing the work of rank R of a P-GPU job with every collective stubbed (torch.distributed's "fake" backend): the
     per-rank compute time a P-GPU run would have, to which tools/scale_model.py adds the xGMI cost of the collectives the line lists

One "step" = one training step of kgwas/kgwas.py:129-151 on one 512-seed batch per GPU: device-side
2-hop full-neighbourhood sampling, feature slicing, 3 feature MLPs, 2 fused attention-aggregate
layers, read-out, LD-weighted MSE, backward, Adam -- on the workload BASELINE.json's metric is quoted
on (configs[1]): SynthKG-fast (784 256 SNPs / 20 032 genes / ~20.6 M directed edges, features
20 / 5120 / 128), causal-simulation-like labels, seed 1.  Inputs (graph, features, labels) are resident
in HBM before the timed region starts.

Metric: edges aggregated per second = sum over timed steps, layers and live relations of the edges the
aggregate kernels actually gather, over wall time (max over ranks).  N > 1: --scaling weak (default: every
rank trains on its own 512-seed batches) or strong (one 512-seed batch per step split over the ranks, what
KGWAS.train does); --parallelism seed (default: graph replicated, gradient buckets all-reduced over RCCL,
the first one under the second half of the backward) or shard (SNP rows sharded, partial-softmax exchange).

Extra keys, all measured by this run:
  roofline     layer-1 forward aggregate kernel: HIP-event time of the launch; L2-side traffic of the same
               launch shape from three `rocprofv3 --pmc` passes this script starts over tools/pmc_probe.py
               (FETCH_SIZE / WRITE_SIZE / TCC hit-miss + MFMA counters; calibrated in the pass on a 1 GiB
               gather of known byte count); frac = traffic / time / 8 TB/s with the time of the launch INSIDE the
               replayed step (round 6; `isolated_launch`: the same launch with the GPU to itself); compulsory and
               algorithmic bytes; the same at a batch whose working set exceeds the Infinity Cache
               (`beyond_infinity_cache`: the figure whose bytes are HBM bytes); MFMA pipe occupancy of the dense kernels
               roofline.in_step: the aggregate kernels INSIDE the replayed step (beside the next batch's sampler), from one more
               process under `rocprofv3 --kernel-trace` -- median layer-1 dispatch of the last 20 steps, frac by the same traffic
  config.sampler_overlap   measured after the timed region: steps with the side sampler, without it, sampler alone; overlap =
               share of the sampler's time that did not show up in the step (`overlapped` false = the two graphs serialised)
  cpu_baseline the CPU oracle (op-for-op PyG restatement) on this box's host cores, 20 steps after 3 warm-ups
  breakdown    per-kernel event times of the three aggregate kernels
  config.epoch_measured   one whole epoch (956 steps + validation pass), wall clock, every batch sampled live; `cached_batches`: a second,
               labelled figure -- epoch 2 of KGWAS.train, the batches of epoch 1 put back from HBM (graph_step.BatchCache)
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


def cpu_baseline(data, batch_size, budget_s=55.0, warmup=3, max_steps=20):
    """Reference PyG CPU path, restated (oracle/): numpy full-neighbour sampler + x[n_id] slicing + unpruned
    2-layer HeteroGNN forward/backward + Adam, on this box's host cores.  SURVEY.md 8d asks for >= 20 steps after 3
    warm-ups; at ~2.5 s per step that is a minute of CPU work, so the timed part stops at ``budget_s`` seconds and the
    line says how many steps fitted."""
    from oracle.gat_oracle import HeteroGNNOracle, weighted_mse
    from oracle.sampler_np import FullNeighborSamplerNP
    # more threads than ~16 only add OpenMP fork/join cost on these small scatter ops (256 threads: 355 s/step
    # vs 13 s/step with 8); use what actually helps and report it
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    g = data.data
    smp = FullNeighborSamplerNP(g.edge_index_dict, g.num_nodes_dict, 2)
    torch.manual_seed(1)
    model = HeteroGNNOracle(g.edge_types, 128, 1, 2, 'GAT', 'sum', data.snp_init_dim_size, data.gene_init_dim_size,
                            data.go_init_dim_size, 1)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=5e-4)
    ids = np.asarray(data.train_input_nodes[1])
    y_all = g['SNP'].y
    w_all = torch.zeros(g['SNP'].x.shape[0], dtype=torch.float64)
    w_all[torch.from_numpy(np.asarray(data.all_ids))] = torch.from_numpy(np.asarray(data.ldsc_weight))
    times, edges = [], []
    for step in range(warmup + max_steps):
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
        if step >= warmup:
            times.append(dt)
            edges.append(2 * sum(int(v.shape[1]) for v in ei.values()))      # both layers touch every sampled edge
            if sum(times) + dt > budget_s:
                break
    tot_t = sum(times)
    return {'value': sum(edges) / tot_t, 'unit': 'edges/s', 'cores': cores, 'kind': 'port',
            's_per_step': tot_t / len(times), 'steps': len(times), 'warmup': warmup,
            'sample': f'{len(times)} training steps after {warmup} warm-ups (SURVEY 8d asks for 20; the timed part is cut at '
                      f'{budget_s:.0f} s of CPU work) of batch {batch_size} on the same SynthKG-fast batches; unpruned (both layers '
                      f'over every sampled edge) like PyG; torch threads={cores} of {os.cpu_count()} host cores'}


# --------------------------------------------------------------------------------------------------------------
# HBM-side traffic of the aggregate kernels, measured IN THIS RUN: tools/pmc_probe.py under rocprofv3, one pass per
# counter group (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots: FETCH_SIZE and WRITE_SIZE cannot share a pass)
# --------------------------------------------------------------------------------------------------------------
PMC_GROUPS = (('fetch', ['FETCH_SIZE']), ('write', ['WRITE_SIZE']),
              ('cache', ['TCC_HIT_sum', 'TCC_MISS_sum', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', 'GRBM_GUI_ACTIVE']))


def _pmc_rows(csv_path):
    import csv
    rows = list(csv.DictReader(open(csv_path)))
    out = {}
    for i, r in enumerate(rows):
        name = r.get('Kernel_Name', '').replace('(anonymous namespace)::', '').replace('void ', '')
        did = int(r.get('Dispatch_Id', i) or i)
        out.setdefault((did, name), {})[r['Counter_Name']] = float(r['Counter_Value'])
    return [(did, name, c) for (did, name), c in sorted(out.items())]


def run_pmc_passes(args, outdir, timeout_s=300):
    """Returns (summary dict, error string or None).  Every pass is a fresh process: `rocprofv3 --pmc <group>
    --kernel-trace -- python tools/pmc_probe.py`; counters are attributed per dispatch by kernel name and order."""
    import shutil
    import subprocess
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not on PATH'
    os.makedirs(outdir, exist_ok=True)
    probe = os.path.join(ROOT, 'tools', 'pmc_probe.py')
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    per = {}
    notes = []
    for tag, counters in PMC_GROUPS:
        pj = os.path.join(outdir, f'{tag}_probe.json')
        cmd = ['rocprofv3', '--pmc', *counters, '--kernel-trace', '--output-format', 'csv', '-d', outdir, '-o', tag, '--',
               sys.executable, probe, '--out', pj, '--batch-size', str(args.batch_size), '--scale', str(args.scale),
               '--snp-scale', str(args.snp_scale), '--mode', args.mode, '--big-batch', str(args.big_batch)]
        try:
            with open(os.path.join(outdir, f'{tag}.log'), 'w') as lf:
                rc = subprocess.run(cmd, cwd='/tmp', env=env, stdout=lf, stderr=subprocess.STDOUT, timeout=timeout_s).returncode
        except Exception as e:                                   # timeout, missing binary, ...
            notes.append(f'{tag}: {type(e).__name__}')
            continue
        cp = os.path.join(outdir, f'{tag}_counter_collection.csv')
        if rc != 0 or not os.path.exists(cp) or not os.path.exists(pj):
            notes.append(f'{tag}: rocprofv3 exit {rc}')
            continue
        rows = _pmc_rows(cp)
        probe_info = json.load(open(pj))
        # calibration = the k_gather_rows dispatch with the largest value of the pass's first counter
        cal = [(c.get(counters[0], 0.0), did) for did, name, c in rows if name.startswith('k_gather_rows')]
        if not cal:
            notes.append(f'{tag}: no calibration dispatch')
            continue
        cal_val, cal_id = max(cal)
        per[tag] = {'rows': [(did, name, c) for did, name, c in rows if did > cal_id], 'cal': dict(next(c for d, n, c in rows if d == cal_id)),
                    'probe': probe_info}
    if 'fetch' not in per or 'write' not in per:
        return None, '; '.join(notes) or 'counter passes incomplete'
    info = per['fetch']['probe']
    n_small = sum(1 for s_ in info['steps'] if s_['batch_size'] == args.batch_size)
    corr_f = info['calibration']['read_bytes'] / max(per['fetch']['cal'].get('FETCH_SIZE', 0.0) * 1024.0, 1.0)
    corr_w = info['calibration']['write_bytes'] / max(per['write']['cal'].get('WRITE_SIZE', 0.0) * 1024.0, 1.0)

    def layer1(tag, kernel, counter):
        """per step, the layer-1 dispatch of `kernel` = the larger of the step's two (layer 1, layer 2) dispatches"""
        vals = [c.get(counter, 0.0) for _, name, c in per[tag]['rows'] if name.startswith(kernel)]
        return [max(vals[i:i + 2]) for i in range(0, len(vals) - 1, 2)]
    summ = {'fetch_correction': corr_f, 'write_correction': corr_w,
            'calibration': {'kernel': 'k_gather_rows, identity ids, %d rows x 512 B (1 GiB >> the 256 MiB Infinity Cache)' % info['calibration']['rows'],
                            'FETCH_SIZE_KB_raw': per['fetch']['cal'].get('FETCH_SIZE'), 'WRITE_SIZE_KB_raw': per['write']['cal'].get('WRITE_SIZE'),
                            'known_read_bytes': info['calibration']['read_bytes'], 'known_write_bytes': info['calibration']['write_bytes']},
            'kernels': {}, 'notes': notes}
    for kernel in ('k_agg_fwd<false', 'k_agg_bwd_dst', 'k_agg_bwd_src'):
        f = layer1('fetch', kernel, 'FETCH_SIZE'); w = layer1('write', kernel, 'WRITE_SIZE')
        ent = {}
        for which, sl in (('batch', slice(0, n_small)), ('big_batch', slice(n_small, None))):
            ff, ww = f[sl], w[sl]
            if not ff or not ww:
                continue
            fb = float(np.median(ff)) * 1024.0 * corr_f
            wb = float(np.median(ww)) * 1024.0 * corr_w
            ent[which] = {'fetch_bytes': fb, 'write_bytes': wb, 'bytes': fb + wb, 'FETCH_SIZE_KB_raw': float(np.median(ff)),
                          'WRITE_SIZE_KB_raw': float(np.median(ww)), 'dispatches': len(ff)}
            if 'cache' in per:
                h = layer1('cache', kernel, 'TCC_HIT_sum')[sl]; m_ = layer1('cache', kernel, 'TCC_MISS_sum')[sl]
                if h and m_ and (np.median(h) + np.median(m_)) > 0:
                    ent[which]['l2_hit_rate'] = float(np.median(h) / (np.median(h) + np.median(m_)))
        summ['kernels'][kernel] = ent
    # layer shapes of the probe's batches (for the compulsory / algorithmic byte counts of the same launches)
    summ['probe_layers'] = {'batch': [s_['layers'][0] for s_ in info['steps'] if s_['batch_size'] == args.batch_size],
                            'big_batch': [s_['layers'][0] for s_ in info['steps'] if s_['batch_size'] != args.batch_size]}
    if 'cache' in per:                                            # MFMA pipe occupancy of the dense kernels
        mf = {}
        for _, name, c in per['cache']['rows']:
            if name.startswith(('k_linear', 'k_tn_gemm', 'k_fold', 'k_g3_gemm', 'k_mlp2', 'Cijk_')) and c.get('GRBM_GUI_ACTIVE', 0) > 0:
                key = name.split('(')[0][:48]
                d = mf.setdefault(key, {'mfma_busy_cycles': 0.0, 'gui_active_cycles': 0.0, 'dispatches': 0})
                d['mfma_busy_cycles'] += c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0); d['gui_active_cycles'] += c['GRBM_GUI_ACTIVE']; d['dispatches'] += 1
        summ['mfma'] = mf
    json.dump(summ, open(os.path.join(outdir, 'summary.json'), 'w'), indent=1)
    return summ, None


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, exactly as the documented
    command does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    <same arguments>).  The children inherit stdout: rank 0's JSON line is the only thing written to it."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: what RCCL needs on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    print('bench.py: starting %d ranks: %s' % (args.gpus, ' '.join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def run_in_step_trace(args, outdir, timeout_s=420):
    """Durations of the aggregate kernels INSIDE the benchmarked step: the HIP-event figures of the
    roofline come from an eager pass in which each kernel has the GPU to itself, but in the captured step the next batch's
    sampler runs beside them.  One more process: `rocprofv3 --kernel-trace -- python bench.py` (20 replayed steps, no eager
    legs), then per kernel name the layer-1 dispatch of each of the last steps (the longer of a step's two dispatches).
    Returns ({kernel: {'median_us', 'mean_us', 'n'}}, error or None)."""
    import csv
    import glob
    import shutil
    import subprocess
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not on PATH'
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR='/tmp', KGW_BENCH_OVERLAP_CHECK='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    n_steps = 20
    cmd = ['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', outdir, '-o', 'instep', '--', sys.executable,
           os.path.abspath(__file__), '--steps', str(n_steps), '--warmup', '3', '--no-cpu-baseline', '--no-pmc', '--no-epoch',
           '--no-kernel-timing', '--no-in-step', '--batch-size', str(args.batch_size), '--scale', str(args.scale),
           '--snp-scale', str(args.snp_scale), '--mode', args.mode]
    try:
        with open(os.path.join(outdir, 'instep.log'), 'w') as lf:
            rc = subprocess.run(cmd, cwd='/tmp', env=env, stdout=lf, stderr=subprocess.STDOUT, timeout=timeout_s).returncode
    except Exception as e:
        return None, type(e).__name__
    files = glob.glob(os.path.join(outdir, '**', 'instep_kernel_trace.csv'), recursive=True)
    if rc != 0 or not files:
        return None, f'rocprofv3 exit {rc}'
    by = {}
    SAMPLER = ('k_fill_i32', 'k_init', 'k_hop_', 'k_seg_deg', 'k_scan_', 'k_fill_chunks', 'k_mark', 'k_count_pending', 'k_assign', 'k_relabel',
               'k_layer_tables', 'k_ts_', 'k_t_end', 'k_meta_to_host', '__amd_rocclr')
    adam_t, starts = [], []
    rows_all = list(csv.DictReader(open(files[0])))
    for r in rows_all:
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        if name.startswith('k_adam'):
            adam_t.append(int(r['Start_Timestamp']))
        elif not any(name.startswith(k) for k in SAMPLER):
            starts.append(int(r['Start_Timestamp']))
    for r in rows_all:
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        for key in ('k_agg_fwd<', 'k_agg_bwd_dst', 'k_agg_bwd_src', 'k_g3_gemm', 'k_mlp2_fwd3', 'k_mlp2_bwd_first3'):
            if name.startswith(key):
                by.setdefault(key.rstrip('<'), []).append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    out = {}
    for key, v in by.items():
        v.sort()
        d = [x[1] for x in v][-2 * n_steps:]                 # two dispatches per step (layer 1 / 2, forward / dW, SNP / gene)
        big = [max(d[i:i + 2]) for i in range(0, len(d) - 1, 2)]
        if big:
            out[key] = {'median_us': float(np.median(big)), 'mean_us': float(np.mean(big)), 'n': len(big)}
    # launches of one replayed step on its own queue = the step's kernels between two consecutive optimiser launches (the last
    # kernel of a step), the optimiser launch included; the side sampler's launches are not counted
    adam_t.sort()
    if len(adam_t) >= 3:
        a, b = adam_t[-2], adam_t[-1]
        out['launches_per_step'] = sum(1 for t in starts if a < t < b) + 1
    os.remove(files[0])                                       # (tens of MB: the summary is what is kept)
    return out, None


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
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help="N > 1: 'weak' = every rank trains on its own --batch-size seeds per step (default); 'strong' = one "
                         "--batch-size batch per step split over the ranks, what KGWAS.train does (kgwas_amd/kgwas.py)")
    ap.add_argument('--parallelism', default='seed', choices=['seed', 'shard'],
                    help="'seed' = seed-data-parallel, graph replicated (SURVEY 8e-i); 'shard' = SNP rows sharded by id range, "
                         "Gene / GO replicated, partial-softmax exchange (SURVEY 8e-ii, north_star); implies strong scaling")
    ap.add_argument('--big-batch', type=int, default=4096, help='seeds per batch of the second roofline measurement (working set > Infinity Cache)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 counter passes (roofline.traffic = null)')
    ap.add_argument('--no-in-step', action='store_true', help='skip the kernel-trace pass that times the aggregate kernels inside the replayed step')
    ap.add_argument('--no-epoch', action='store_true', help='skip the measured epoch (956 training steps + validation pass)')
    ap.add_argument('--eager', action='store_true', help='issue every launch from the host instead of replaying one HIP graph per step')
    ap.add_argument('--as-rank', default=None, metavar='R/P',
                    help='emulate rank R of a P-GPU job on ONE GPU: its share of the seeds / of the SNP range, every collective a '
                         'no-op of the right size ("fake" process group); prints that rank\'s compute time per step')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and args.as_rank is None:
        raise SystemExit(self_launch(args))

    # stdout carries EXACTLY one JSON line: everything else that writes to file descriptor 1 (native libraries included --
    # a collective backend announcing its peers, a BLAS tuner) goes to stderr until the line is printed
    sys.stdout.flush()
    _stdout_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus and args.as_rank is None:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks (or unset WORLD_SIZE: bench.py starts them itself)')
    assert torch.cuda.is_available(), 'bench.py needs a ROCm GPU'
    n_dev = torch.cuda.device_count()
    shared_devices = world > n_dev               # fewer GPUs than ranks: the ranks share devices (RCCL refuses that: gloo)
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    dev = f'cuda:{local_rank}'
    emulated = None
    if args.as_rank is not None:
        # rank R of P on one GPU: partitioning, buffers and launches of that rank, collectives that move nothing
        import torch.distributed as dist
        from torch.testing._internal.distributed.fake_pg import FakeStore
        r_, p_ = (int(v) for v in args.as_rank.split('/'))
        assert 0 <= r_ < p_ and world == 1, '--as-rank R/P runs in ONE process'
        dist.init_process_group(backend='fake', rank=r_, world_size=p_, store=FakeStore())
        rank, world, emulated = r_, p_, {'rank': r_, 'world': p_}
    elif world > 1 or os.environ.get('KGW_FORCE_MULTIRANK_PATH') == '1':       # (the latter: one rank through the RCCL path)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29555'); os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        backend = os.environ.get('KGW_DIST_BACKEND', 'gloo' if shared_devices else 'nccl')   # "nccl" == RCCL over xGMI on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)
    shard = args.parallelism == 'shard'
    strong = shard or args.scaling == 'strong'

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
    nb = len(ids) // bs
    if strong:
        # strong scaling: the reference's own batch order; every rank works on ITS part of each 512-seed batch
        mine = ids[:nb * bs]
        if len(mine) < need:
            mine = np.resize(mine, need)
        mine = mine[:need]
        if not shard:
            mine = kdist.shard_batches(mine, bs, rank, world)
        bs_rank = bs // world if not shard else bs
    else:
        # weak scaling: rank r trains on batches r, r+world, ... of the reference's fixed batch order
        mine = ids[:nb * bs].reshape(nb, bs)[rank::world].reshape(-1)
        if len(mine) < need:
            mine = np.resize(mine, need)
        mine = mine[:need]
        bs_rank = bs
    run.model.train()

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    gs = None
    if shard:
        from kgwas_amd.shard import ShardedTrainer
        st_ = ShardedTrainer(run, ('SNP', mine), bs, lr=1e-4, weight_decay=5e-4, use_graph=not args.eager)
        stats_e = [0, 0]

        def do_step(i):
            e = st_.step(i)
            if e is not None:                 # (uncaptured form counts as it goes; the captured one after the timed region)
                stats_e[0] += e[0]; stats_e[1] += e[1]
        mode = st_.describe()
    elif args.eager:
        opt = torch.optim.Adam(run.model.parameters(), lr=1e-4, weight_decay=5e-4)
        it = iter(NeighborLoader(data.data, [-1, -1], ('SNP', mine), batch_size=bs_rank, drop_last=True, device=dev))
        stats_e = [0, 0]

        def do_step(i):
            batch = next(it)
            run.train_step(batch, opt, ld_w, world)
            stats_e[0] += sum(batch.n_edges_per_layer)
            stats_e[1] += 2 * batch.n_edges_sampled
        mode = 'eager launches'
    else:
        from kgwas_amd.graph_step import GraphTrainStep
        # (every rank would repeat the batch-independent first gene Linear, weak scaling or strong: split by gene rows over the
        #  ranks where that pays -- GraphTrainStep's default, ops.gene_layer_split_pays)
        gs = GraphTrainStep(run, ('SNP', mine), bs_rank, lr=1e-4, weight_decay=5e-4)

        def do_step(i):
            gs.step(i)
        mode = gs.describe()
    setup_s = time.time() - t0

    def collective_counters():
        """{name: [calls, bytes]} of everything the trainers handed to a collective since the counters were cleared"""
        src = {k: list(v) for k, v in kdist.COLLECTIVES.items()}
        if shard:
            src.update({k: list(v) for k, v in st_.collectives().items()})
        elif gs is not None and gs.gene_shard is not None:
            src.update({k: list(v) for k, v in gs.gene_shard.bytes.items() if v[0]})
        return src
    # count what the warm-up + timed steps move, not the trainers' set-up passes (capacity dry runs, capture warm-ups)
    kdist.COLLECTIVES.clear()
    if shard:
        st_.xchg.collectives = {}
        if st_.gene_shard is not None:
            for v in st_.gene_shard.bytes.values():
                v[0] = v[1] = 0
    elif gs is not None and gs.gene_shard is not None:
        for v in gs.gene_shard.bytes.values():
            v[0] = v[1] = 0

    for i in range(args.warmup):
        do_step(i)
    if gs is not None:
        torch.cuda.synchronize()
        gs.stats.zero_()
    else:
        stats_e[0] = stats_e[1] = 0
    sync()
    # (N > 1: every collective of the timed steps also timed by HIP events on its stream -- two event records per call, no sync)
    ctimer = kdist.CollectiveTimer() if (world > 1 and emulated is None) or os.environ.get('KGW_FORCE_MULTIRANK_PATH') == '1' else None
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record()
    if ctimer is not None:
        with ctimer:
            for i in range(args.steps):
                do_step(args.warmup + i)
    else:
        for i in range(args.steps):
            do_step(args.warmup + i)
    ev1.record()
    sync()
    elapsed = time.perf_counter() - t_start
    rank_ms = ev0.elapsed_time(ev1) / args.steps          # this rank's own device time per step (no barrier inside)
    coll_src = collective_counters()          # (before the untimed passes below -- edge counting, overlap check -- add their own)
    # products handed to the framework's GEMM library so far (set-up, warm-up, capture, timed steps): must be 0 on the headline
    lib_calls, lib_sites = ops.LIBRARY_GEMM.calls, {f'{k[0]} {k[1]}': v for k, v in ops.LIBRARY_GEMM.by_site.items()}
    seeds = args.steps * (bs if shard else bs_rank)
    if gs is not None:
        st = gs.check()                       # raises if a batch overflowed the static layout
        edges_kernel, edges_ref = sum(st[:2]), 2 * st[2]
    else:
        if shard and st_.use_graph:
            st_.check()                       # raises if a batch overflowed the static layout
            for i in range(args.steps):       # edges of the timed batches, each counted once across the ranks (untimed pass)
                e1, e2 = st_.count_edges(args.warmup + i)
                stats_e[0] += e1; stats_e[1] += e2
        edges_kernel, edges_ref = stats_e
    if shard and rank != 0:
        seeds = 0                             # (every rank works on the same batches: count the seeds once)
    # did the next batch's sampler really run beside the step (two streams on one hardware queue would serialise silently)?
    # measured, after the timed region: n steps with it, n without, n sampler replays alone (every rank takes part)
    overlap = None
    if os.environ.get('KGW_BENCH_OVERLAP_CHECK', '1') == '1':
        if gs is not None:
            overlap = gs.measure_overlap(min(args.steps, 20))
        elif shard and st_.use_graph:
            overlap = st_.measure_overlap(min(args.steps, 20))

    single = world == 1 and not shard
    # kernel-level timing for the roofline: HIP events around the aggregate launches on their stream, in an
    # eager pass over the same batches right after the timed region (events cannot bracket nodes of a replayed graph)
    big_summ = {}
    if not args.no_kernel_timing and single:
        opt_r = torch.optim.Adam(run.model.parameters(), lr=1e-4, weight_decay=5e-4)
        n_r = min(args.steps, 20)
        it_r = iter(NeighborLoader(data.data, [-1, -1], ('SNP', mine[args.warmup * bs:(args.warmup + n_r) * bs]),
                                   batch_size=bs, drop_last=True, device=dev))
        ops.TIMER.enabled = True
        for _ in range(n_r):
            run.train_step(next(it_r), opt_r, ld_w, world)
        torch.cuda.synchronize()
        ops.TIMER.enabled = False
        summ = ops.TIMER.summary()
        # the same kernels on batches of --big-batch seeds: one launch's working set (~0.3 GB of gathered rows) no longer
        # fits the 256 MiB Infinity Cache, so the counter traffic of THOSE launches is HBM traffic
        if args.big_batch and args.big_batch * 3 <= len(ids):
            ops.TIMER.records.clear()
            it_b = iter(NeighborLoader(data.data, [-1, -1], ('SNP', ids[args.big_batch:4 * args.big_batch]), batch_size=args.big_batch,
                                       drop_last=True, device=dev, prefetch=False))
            ops.TIMER.enabled = True
            for _ in range(3):
                bb = next(it_b)
                for p_ in run.model.parameters():
                    p_.grad = None
                loss_b, _ = run.model.forward_loss(bb.x_dict, bb.edge_index_dict, args.big_batch, bb.n_id('SNP'), bb.dg.y['SNP'], ld_w)
                loss_b.backward()
            torch.cuda.synchronize()
            ops.TIMER.enabled = False
            big_summ = ops.TIMER.summary()
            del it_b, bb
    else:
        summ = {}

    # one measured epoch at reference scale: the 956 training steps of the fixed batch order + the validation pass
    epoch = None
    if not args.no_epoch and single and not args.eager and args.scale == 1.0:
        from kgwas_amd.graph_step import GraphTrainStep
        from kgwas_amd.utils import evaluate_minibatch_clean
        run.make_loaders(bs)
        ge = GraphTrainStep(run, ('SNP', ids), bs, lr=1e-4, weight_decay=5e-4)
        for i in range(3):
            ge.step(i)
        torch.cuda.synchronize()
        te = time.perf_counter()
        for i in range(ge.n_batches):
            ge.step(i)
        ge.check()
        t_train = time.perf_counter() - te
        tv = time.perf_counter()
        val = evaluate_minibatch_clean(run.val_loader, run.model, dev)
        torch.cuda.synchronize()
        t_val = time.perf_counter() - tv
        tv = time.perf_counter()
        evaluate_minibatch_clean(run.val_loader, run.model, dev)          # every later epoch: the captured forward graph is reused
        torch.cuda.synchronize()
        t_val2 = time.perf_counter() - tv
        epoch = {'train_steps': ge.n_batches, 'train_s': t_train, 'val_batches': len(run.val_loader), 'val_s': t_val,
                 'val_s_later_epochs': t_val2, 'epoch_s': t_train + t_val, 'epoch_s_later_epochs': t_train + t_val2,
                 'note': 'one pass over the 489 839 training SNPs in the reference batch order (kgwas/kgwas.py:129) + '
                 'the validation pass of kgwas.py:157 (val_s: first use, includes measuring its static capacities and capturing its '
                 'forward graph; val_s_later_epochs: the same pass again, as every later epoch runs it)'}
        del ge
        # a SECOND, separately labelled figure (the headline and the epoch above sample every batch live): what KGWAS.train's epochs
        # >= 2 cost -- the loader's batch order is fixed (kgwas/kgwas.py:93-101), so the batches sampled in epoch 1 are kept in HBM and
        # put back by one copy launch each instead of being sampled again (graph_step.BatchCache)
        try:
            gc_ = GraphTrainStep(run, ('SNP', ids), bs, lr=1e-4, weight_decay=5e-4, cache_batches=True)
            if gc_.cache is not None:
                torch.cuda.synchronize()
                te = time.perf_counter()
                for i in range(gc_.n_batches):
                    gc_.step(i)
                gc_.check()
                t_fill = time.perf_counter() - te
                te = time.perf_counter()
                for i in range(gc_.n_batches):
                    gc_.step(i)
                gc_.check()
                t_cached = time.perf_counter() - te
                epoch['cached_batches'] = {
                    'train_s_epoch_1_sampling_and_keeping': t_fill, 'train_s_epoch_2_from_the_cache': t_cached,
                    'ms_per_step_from_the_cache': t_cached / gc_.n_batches * 1e3, 'cache_gib': gc_.cache.slots.numel() / 2 ** 30,
                    'mb_per_batch': gc_.cache.slot_bytes / 1e6,
                    'note': 'NOT the headline: what epochs >= 2 of KGWAS.train(epoch > 1) cost (batches of epoch 1 kept in HBM, '
                            'bit-identical training: tests/test_gpu_graph.py); value / ms_per_step / epoch_s above sample every batch live'}
            del gc_
        except Exception as e:                                  # (never let the second figure take the line down)
            epoch['cached_batches'] = {'error': repr(e)}

    # what moved between the ranks: every collective of the timed region, by name, per step and rank (world 1: empty)
    coll = {}
    if world > 1 or os.environ.get('KGW_FORCE_MULTIRANK_PATH') == '1':
        n_all = args.steps + args.warmup
        coll = {k: {'calls_per_step': v[0] / n_all, 'bytes_per_step': v[1] / n_all} for k, v in coll_src.items()}
    comm = {'world_size': world, 'backend': (torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
            'rccl': bool(torch.distributed.is_initialized() and torch.distributed.get_backend() == 'nccl'),
            'collectives_per_step_and_rank': coll}
    if shared_devices:
        comm['note'] = ('%d ranks on %d GPU(s): the ranks SHARE devices and talk over %s -- a dry run of the multi-rank code path, '
                        'not a scaling measurement' % (world, n_dev, comm['backend']))
    if emulated is not None:
        comm['note'] = ('--as-rank %d/%d: ONE GPU doing the work of that rank, every collective a no-op of the listed size ("fake" '
                        'process group); ms_per_step is the rank\'s compute time, value the rank\'s own throughput; '
                        'tools/scale_model.py adds the xGMI time of the collectives' % (rank, world))
    stats = torch.tensor([elapsed, float(edges_kernel), float(edges_ref), float(seeds)], dtype=torch.float64, device=dev)
    if ctimer is not None:
        # measured device time of each collective, next to what the alpha-beta model of tools/scale_model.py gives for the same
        # message on P GPUs over xGMI (direct / ring): ONE driver SCALE record calibrates its two constants
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        from scale_model import A_HOP, A_PHASE, BETA, t_collective
        meas = ctimer.summary(args.steps)
        for k, v in meas.items():
            nbytes = int(k[k.index('(') + 1:k.index(' B)')])
            v['model_ms_per_call_direct'] = t_collective(k, nbytes, world, 'direct') * 1e3
            v['model_ms_per_call_ring'] = t_collective(k, nbytes, world, 'ring') * 1e3
        comm['collectives_measured'] = meas
        comm['collectives_measured_note'] = ('HIP events on the calling stream around every torch.distributed collective of the timed steps '
                                             '(this rank); model_*: tools/scale_model.py, alpha %.0f us per phase / %.0f us per ring hop, beta %.0f GB/s '
                                             'per link' % (A_PHASE * 1e6, A_HOP * 1e6, BETA / 1e9))
    comm['per_rank_ms'] = {'this_rank': rank_ms}
    if world > 1 and emulated is None:
        tmax = stats[:1].clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        tot = stats[1:].clone()
        torch.distributed.all_reduce(tot, op=torch.distributed.ReduceOp.SUM)
        elapsed = float(tmax[0]); edges_kernel, edges_ref, seeds = (float(x) for x in tot)
        # every rank's own event-timed ms per step, and the rank count as the collective backend itself sees it (a sum of ones)
        mine_ms = torch.tensor([rank_ms], dtype=torch.float64, device=dev)
        all_ms = [torch.zeros_like(mine_ms) for _ in range(world)]
        torch.distributed.all_gather(all_ms, mine_ms)
        all_ms = [float(t[0]) for t in all_ms]
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        torch.distributed.all_reduce(ones, op=torch.distributed.ReduceOp.SUM)
        comm['per_rank_ms'] = {'min': min(all_ms), 'max': max(all_ms), 'by_rank': all_ms,
                               'note': 'HIP-event time of the timed steps / steps on each rank\'s own stream; ms_per_step is the barrier-to-barrier wall clock, max over ranks'}
        comm['ranks_seen_by_the_backend'] = int(round(float(ones[0])))

    if torch.distributed.is_initialized():
        if emulated is None:
            torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank != 0 and emulated is None:
        return

    pmc, pmc_err = (None, 'skipped')
    if not args.no_pmc and single and not args.no_kernel_timing:
        pmc, pmc_err = run_pmc_passes(args, os.path.join(ROOT, 'gpurun_out', 'bench_pmc'))

    in_step, in_step_err = (None, 'skipped')
    if not args.no_in_step and single and not args.no_kernel_timing and not args.eager:
        in_step, in_step_err = run_in_step_trace(args, os.path.join(ROOT, 'gpurun_out', 'bench_instep'))

    def agg_entry(tag, d):
        entry = {'launches': d['n'], 'avg_ms': d['ms'] / d['n'], 'edges_per_launch': d['edges'] / d['n']}
        alg = {'fwd': algorithmic_bytes_fwd(d['edges'], d['z_rows']), 'bwd_dst': 528 * d['edges'] + 1028 * d['z_rows'],
               'bwd_src': 528 * d['edges'] + 520 * d['n_src']}[tag]
        entry['algorithmic_GBs'] = alg / (d['ms'] * 1e-3) / 1e9
        return entry
    breakdown = {f'agg_{tag}_l{layer}': agg_entry(tag, d) for (tag, layer), d in sorted(summ.items())}
    for (tag, layer), d in sorted(big_summ.items()):
        breakdown[f'agg_{tag}_l{layer}_batch{args.big_batch}'] = agg_entry(tag, d)

    def roofline_of(d, counters, what):
        """d: TIMER summary of the layer-1 forward launches; counters: run_pmc_passes entry of the same launch shape."""
        n = d['n']
        alg = algorithmic_bytes_fwd(d['edges'], d['z_rows']) / n
        # bytes a launch cannot avoid: every row of the layer input once (all are gathered at least once), the column
        # index and the logit of every edge once, the Z row + softmax statistics of every segment, the chunk records
        comp = (512 * d['n_src'] + 8 * d['edges'] + 520 * d['z_rows']) / n
        ms = d['ms'] / n
        r = {'kernel': 'k_agg_fwd (layer-1 attention aggregate, forward)', 'workload': what, 'bound': 'hbm',
             'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'avg_launch_ms': ms, 'edges_per_launch': d['edges'] / n,
             'algorithmic_bytes': alg, 'algorithmic_GBs': alg / ms / 1e6, 'algorithmic_frac': alg / ms / 1e6 / HBM_PEAK_GBS,
             'compulsory_bytes': comp, 'compulsory_GBs': comp / ms / 1e6,
             'timing': 'hipEvent pair recorded by the C ABI immediately around the k_agg_fwd launch on its stream (KgwLayerArgs.ev_before/ev_after), eager pass'}
        if counters:
            r['traffic'] = counters['bytes']
            r['traffic_fetch_bytes'], r['traffic_write_bytes'] = counters['fetch_bytes'], counters['write_bytes']
            r['achieved'] = counters['bytes'] / ms / 1e6
            r['traffic_over_compulsory'] = counters['bytes'] / comp
            if 'l2_hit_rate' in counters:
                r['l2_hit_rate'] = counters['l2_hit_rate']
            r['achieved_is'] = ('L2-side (fabric) bytes of the launch, FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes made by '
                                'this run, corrected by the known byte count of a 1 GiB row gather in the same pass, / HIP-event time')
        else:
            r['traffic'] = None
            r['achieved'] = comp / ms / 1e6
            r['achieved_is'] = 'no counters in this run: compulsory bytes / HIP-event time (a LOWER bound on the traffic rate)'
        r['frac'] = r['achieved'] / HBM_PEAK_GBS
        return r

    roof = None
    if ('fwd', 1) in summ:
        ck = (pmc or {}).get('kernels', {}).get('k_agg_fwd<false', {})
        roof = roofline_of(summ[('fwd', 1)], ck.get('batch'), f'batch {bs} (the benchmark workload)')
        # rocprofv3 --stats averages ALL launches of the kernel name (layer 1: ~1 M edges, layer 2: ~2 k edges) together:
        # the figure to compare with profiles/*kernel_stats*.csv
        allf = [v for (tag, _), v in summ.items() if tag == 'fwd']
        roof['avg_ms_all_launches_of_this_kernel_name'] = sum(v['ms'] for v in allf) / max(1, sum(v['n'] for v in allf))
        if ('fwd', 1) in big_summ:
            roof['beyond_infinity_cache'] = roofline_of(big_summ[('fwd', 1)], ck.get('big_batch'),
                                                        f'batch {args.big_batch}: the gathered rows of one launch exceed the 256 MiB Infinity Cache, FETCH_SIZE is HBM traffic')
        if pmc:
            roof['counter_files'] = 'gpurun_out/bench_pmc/{fetch,write,cache}_counter_collection.csv + summary.json (written by this run)'
            roof['fetch_correction'], roof['write_correction'] = pmc['fetch_correction'], pmc['write_correction']
            roof['other_kernels'] = {k: v for k, v in pmc['kernels'].items() if k != 'k_agg_fwd<false'}
            if pmc.get('mfma'):
                # MFMA pipe occupancy of the dense kernels: SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs,
                # GRBM_GUI_ACTIVE over the 8 XCDs (checked on k_linear_wreg: 1.0 M fp32 32x32x2 MFMAs x 64 cycles = 64 M busy
                # cycles counted as 61.4 M; 49 us x 2.57 GHz x 8 = 1.01 M GUI cycles) => SIMD-cycles = GUI / 8 x 1024
                roof['mfma_util'] = {k: {'mfma_busy_over_simd_cycles': v['mfma_busy_cycles'] / max(v['gui_active_cycles'] * 128.0, 1.0),
                                         'dispatches': v['dispatches']} for k, v in pmc['mfma'].items()}
        else:
            roof['counter_note'] = f'counter passes unavailable: {pmc_err}'
        if pmc and not ck.get('batch'):
            roof['counter_note'] = 'counter passes ran but no k_agg_fwd dispatch was attributed (kernel renamed?): traffic unknown'
        # the same kernels INSIDE the replayed step (beside the next batch's sampler): rocprofv3 kernel trace of 20 replays
        if in_step:
            ins = {'how': 'rocprofv3 --kernel-trace over a second run of this script (20 replayed steps): per step the layer-1 dispatch '
                          '(the longer of the two) of each kernel, median over the steps; `frac` = this run\'s counter traffic of the '
                          'same launch shape / that time / 8 TB/s (the isolated eager launch above: `frac`)'}
            for key, tag in (('k_agg_fwd', 'k_agg_fwd<false'), ('k_agg_bwd_dst', 'k_agg_bwd_dst'), ('k_agg_bwd_src', 'k_agg_bwd_src')):
                if key in in_step:
                    e = dict(in_step[key])
                    tb = ((pmc or {}).get('kernels', {}).get(tag, {}).get('batch') or {}).get('bytes')
                    if tb:
                        e['traffic'] = tb
                        e['achieved'] = tb / (e['median_us'] * 1e-6) / 1e9
                        e['frac'] = e['achieved'] / HBM_PEAK_GBS
                    ins[key] = e
            for key in ('k_g3_gemm', 'k_mlp2_fwd3', 'k_mlp2_bwd_first3', 'launches_per_step'):
                if key in in_step:
                    ins[key] = in_step[key]
            roof['in_step'] = ins
            # the line's `frac` / `achieved` are what the STEP pays -- the launch inside the replayed step, beside
            # the next batch's sampler; the isolated eager launch moves to a sub-key; beyond_infinity_cache stays the HBM-true figure
            e = ins.get('k_agg_fwd') or {}
            if 'frac' in e:
                roof['isolated_launch'] = {'avg_launch_ms': roof['avg_launch_ms'], 'achieved': roof['achieved'], 'frac': roof['frac'],
                                           'timing': roof['timing']}
                roof['achieved'], roof['frac'] = e['achieved'], e['frac']
                roof['avg_launch_ms_in_step'] = e['median_us'] * 1e-3
                roof['frac_is'] = ('IN-STEP: this run\'s counter traffic of the layer-1 k_agg_fwd launch / its median duration inside the replayed '
                                   'step (rocprofv3 kernel trace of 20 replays, sampler beside it) / 8 TB/s; `isolated_launch`: the same launch '
                                   'with the GPU to itself (HIP events); `beyond_infinity_cache`: the launch shape whose bytes are HBM bytes')
        else:
            roof['in_step'] = {'note': f'kernel-trace pass unavailable: {in_step_err}'}
            roof['frac_is'] = 'the ISOLATED eager launch (HIP events); no in-step kernel trace in this run'

        hit = roof.get('l2_hit_rate')
        roof['bounded_by'] = ('at batch %d the launch touches %.0f MB of distinct rows (< 256 MiB Infinity Cache): the fabric-side bytes above are '
                              'served by L3 + HBM together, so `frac` is an UPPER bound on HBM utilisation; the binding resource is the per-XCD '
                              'L2 miss path (each of the 8 non-coherent L2s fetches its own copy of a hot row%s); see beyond_infinity_cache for the '
                              'launch shape where the same counters are HBM bytes' %
                              (bs, roof['compulsory_bytes'] / 1e6, '' if hit is None else ', L2 hit rate %.2f' % hit))
    cpu = None
    if not args.no_cpu_baseline and world == 1:          # the CPU leg is timed at N = 1 only
        cpu = cpu_baseline(data, bs)
    ms = elapsed / args.steps * 1e3
    headline = args.mode == 'fast' and args.snp_scale == 1.0 and args.scale == 1.0
    par = (f'snp-shard{world} (SNP rows by id range, Gene/GO replicated, partial-softmax exchange)' if shard else f'seed-dp{world}')
    out = {
        'metric': 'edges aggregated/sec (full fast-mode KG minibatch training; epoch time in config)',
        'value': edges_kernel / elapsed, 'unit': 'edges/s', 'n_gpus': 1 if emulated is not None else world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak',
        'scaling_note': ('weak: every GPU trains on its own 512-seed batches of the reference\'s order against a replicated graph -- a GLOBAL batch of N x 512 seeds per step, i.e. a different optimisation trajectory from the reference\'s 512 at the same lr --, one '
                         'all-reduce of the parameter gradients per step (SURVEY 8e-i) -- the default of --gpus N and the mode that scales: '
                         'a 784 k-SNP graph fits one MI355X many times over.  The partitioning north_star prescribes (--parallelism shard: '
                         'SNP rows by id range, Gene / GO replicated, partial-softmax exchange) is built and tested but is predicted BELOW '
                         '1x at this graph size (profiles/r4/r4_j_scale_model.md: 0.65 - 0.95x at 8 GPUs; strong seed-parallel 1.2 - 1.6x): '
                         'every rank repeats the gene / GO side, which is 70 % of a batch\'s edges'),
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': ('SynthKG-fast full KG (784256 SNP / 20032 Gene / ~20.6M directed edges; features '
                                '20/5120/128) + causal-simulation GWAS seed=1, batch 512 seeds %s, 2-layer GAT-128, '
                                'Adam(1e-4, wd 5e-4) -- BASELINE.json configs[1]' % ('per step, split over the GPUs' if strong else 'per GPU')) if headline else
                               (f'SynthKG-fast with scale={args.scale}, snp_scale={args.snp_scale} (see config.graph) -- NOT the headline configuration; '
                                'snp_scale 12.75 is the ~10 M-SNP full-cohort case of BASELINE.json configs[3]') if args.mode == 'fast' else
                               ('SynthKG-full: same graph, full-mode feature widths 70/57742/128 -- BASELINE.json configs[4], '
                                'not the headline configuration'),
                   'scale': args.scale, 'snp_scale': args.snp_scale,
                   'graph': {'nodes': {t: int(data.data[t].x.shape[0]) for t in data.data.node_types},
                             'directed_edges': int(sum(data.data[et].edge_index.shape[1] for et in data.data.edge_types))},
                   'batch_size_per_gpu': bs_rank, 'global_batch': bs if strong else bs * world, 'parallelism': par, 'execution': mode,
                   'edges_per_step_kernel': edges_kernel / args.steps / (1 if strong else world),
                   'edges_per_step_reference_equivalent': edges_ref / args.steps / (1 if strong else world),
                   'reference_equivalent_edges_per_s': edges_ref / elapsed,
                   'seeds_per_s': seeds / elapsed,
                   'epoch_time_s_956_steps': 956 * ms / 1e3 / (1 if strong else world),
                   'epoch_measured': epoch,
                   'rccl_world_size': world if comm['rccl'] else 0,
                   'emulated_rank': emulated,
                   'sampler_overlap': overlap,
                   'communication': comm,
                   'library_gemm_calls': lib_calls,
                   'library_gemm_note': ('products of set-up + warm-up + capture + the timed steps that went to hipBLASLt / rocBLAS instead of this '
                                         'package\'s HIP kernels (kgwas_amd.ops.LIBRARY_GEMM; KGW_STRICT=1 makes any such route an error)' +
                                         (': ' + json.dumps(lib_sites) if lib_sites else '')),
                   'arithmetic': ('fp32 operands, fp32 accumulation, fp32 results.  The two 5120-wide gene products and the 128 x 128 products '
                                  'of the SNP feature MLP (forward, first-layer backward) run on the bf16 matrix pipe from three EXACT bf16 '
                                  'pieces per fp32 operand (a = a1 + a2 + a3, six piece products kept, the dropped ones <= 3*2^-25|ab| per term: '
                                  'below one fp32 multiply-add rounding); measured against float64 at or below the fp32 product\'s own error '
                                  '(tests/test_gpu_gemm3.py, tests/test_split3_bound.py, DESIGN.md section 1); '
                                  'KGW_GEMM3=0 KGW_MLP2_SPLIT=0 run them on the fp32 pipe instead')
                                 if (os.environ.get('KGW_GEMM3', '1') != '0' or os.environ.get('KGW_MLP2_SPLIT', '1') != '0') else
                                 'fp32 operands, fp32 accumulation, fp32 matrix pipe throughout (KGW_GEMM3=0 KGW_MLP2_SPLIT=0)',
                   'setup_s': setup_s},
        'roofline': roof, 'cpu_baseline': cpu, 'breakdown': breakdown,
    }
    sys.stdout.flush()
    try:                                      # C stdio too: RCCL prints its version banner with printf, block-buffered on a pipe
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.dup2(_stdout_fd, 1)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
