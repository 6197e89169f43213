"""CPU, world_size 2, gloo: the multi-GPU path (kgwas_amd/dist.py) -- seeds of every batch are split over
the ranks, each rank computes its slice, parameter gradients are averaged in one flat all-reduce.  The
result must equal the single-process gradient of the full batch (same SGD step as the reference,
SURVEY.md 8e-i).  The model here is the CPU oracle (the HIP model needs a GPU); what is under test is the
sharding + collective logic, which is device independent."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from kgwas_amd import dist as kdist
        from kgwas_amd.kgwas_data import KGWAS_Data
        from oracle.gat_oracle import HeteroGNNOracle, weighted_mse
        from oracle.sampler_np import FullNeighborSamplerNP
        torch.set_num_threads(2)
        data = KGWAS_Data.from_synthetic(scale=0.002, seed=3, feat_dims={'Gene': 40}, data_path=f'/tmp/kgwas_dist_{rank}')
        g = data.data
        torch.manual_seed(100 + rank)                       # deliberately different init per rank
        model = HeteroGNNOracle(g.edge_types, 128, 1, 2, 'GAT', 'sum', 20, 40, 128, 1, dtype=torch.float64)
        kdist.broadcast_params(model)                       # rank 0's weights everywhere
        assert kdist.rank_world() == (rank, world)
        bs = 32
        ids = np.asarray(data.train_input_nodes[1][:2 * bs])
        mine = kdist.shard_batches(ids, bs, rank, world)
        smp = FullNeighborSamplerNP(g.edge_index_dict, g.num_nodes_dict, 2)
        y_all = g['SNP'].y.double()
        per = bs // world
        seeds = mine[:per]                                  # this rank's slice of batch 0
        n_id, ei = smp.sample('SNP', seeds)
        x = {t: g[t].x[n_id[t]].double() for t in g.node_types}
        loss = weighted_mse(model(x, ei, per), y_all[n_id['SNP'][:per]], torch.ones(per, dtype=torch.float64))
        loss.backward()
        kdist.allreduce_grads(model, world)
        torch.save({'grads': {n: p.grad for n, p in model.named_parameters() if p.grad is not None},
                    'params': {n: p.detach() for n, p in model.named_parameters()}, 'seeds': seeds},
                   os.path.join(out_dir, f'rank{rank}.pt'))
        if rank == 0:                                       # single-process reference on the full batch
            for p in model.parameters():
                p.grad = None
            n_id, ei = smp.sample('SNP', ids[:bs])
            x = {t: g[t].x[n_id[t]].double() for t in g.node_types}
            loss = weighted_mse(model(x, ei, bs), y_all[n_id['SNP'][:bs]], torch.ones(bs, dtype=torch.float64))
            loss.backward()
            torch.save({n: p.grad for n, p in model.named_parameters() if p.grad is not None},
                       os.path.join(out_dir, 'full.pt'))
    finally:
        dist.destroy_process_group()


def test_seed_sharded_gradients_equal_full_batch(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt', weights_only=False)
    r1 = torch.load(tmp_path / 'rank1.pt', weights_only=False)
    full = torch.load(tmp_path / 'full.pt', weights_only=False)
    assert not np.array_equal(r0['seeds'], r1['seeds'])
    for n in r0['params']:                                  # broadcast worked
        assert torch.equal(r0['params'][n], r1['params'][n]), n
    assert set(r0['grads']) == set(r1['grads']) == set(full)
    for n, gfull in full.items():
        assert torch.equal(r0['grads'][n], r1['grads'][n]), n               # all-reduce: identical on both ranks
        assert torch.allclose(r0['grads'][n], gfull, rtol=1e-9, atol=1e-12), n


def _worker_split(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from kgwas_amd import dist as kdist
        from kgwas_amd import ops
        gs = ops.GeneLayerShard(rank, world, None, inline=False)
        ok = gs.selftest(torch.device('cpu'))                  # its two collectives on a few KB, verdict agreed over the ranks
        assert all(v == [0, 0] for v in gs.bytes.values())    # (the self-test is not counted as traffic of the step)
        # gradient liveness is RE-agreed when a rank sees a gradient on a parameter the first agreement left out
        m = torch.nn.Linear(3, 2, bias=True)
        m.weight.grad = torch.full_like(m.weight, float(rank + 1))
        kdist.allreduce_grads(m, world)                       # step 1: the bias is dead on every rank
        first = (m.weight.grad.clone(), m.bias.grad)
        m.weight.grad = torch.full_like(m.weight, 1.0)
        if rank == 1:
            m.bias.grad = torch.full_like(m.bias, 4.0)        # step 2: alive on ONE rank only
        kdist.allreduce_grads(m, world)
        second_b = m.bias.grad                               # (round 5: the flag rides in the bucket and is looked at one call later)
        m.weight.grad = torch.full_like(m.weight, 1.0)
        m.bias.grad = None                                    # step 3: NO rank has the bias gradient (a relation that shows up on
        kdist.allreduce_grads(m, world)                       # non-consecutive batches): re-agreed all the same -- ADVICE r5
        third_b = m.bias.grad
        m.weight.grad = torch.full_like(m.weight, 1.0)
        m.bias.grad = None
        if rank == 1:
            m.bias.grad = torch.full_like(m.bias, 4.0)
        kdist.allreduce_grads(m, world)                       # step 4: its next appearance is reduced on both ranks
        torch.save({'ok': ok, 'first_w': first[0], 'first_b': first[1], 'second_b': second_b, 'third_b': third_b,
                    'fourth_b': m.bias.grad}, os.path.join(out_dir, f's{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_gene_layer_split_selftest_and_liveness_reagreement(tmp_path):
    world = 2
    mp.spawn(_worker_split, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        d = torch.load(tmp_path / f's{r}.pt', weights_only=False)
        assert d['ok'] is True
        assert torch.equal(d['first_w'], torch.full((2, 3), 1.5)) and d['first_b'] is None
        # the parameter that came alive on rank 1 is never stepped locally on one rank: the step that sees it drops the local
        # gradient on every rank (the flag travels in the gradient bucket, no extra collective / host sync per step) and the next
        # one makes it live on BOTH ranks -- although no rank has a gradient for it at THAT step (the rank that raised the flag
        # remembers which parameter it was): zeros there, and the mean of 0 and 4 at its next appearance
        assert d['second_b'] is None
        assert d['third_b'] is not None and torch.equal(d['third_b'], torch.zeros(2))
        assert d['fourth_b'] is not None and torch.equal(d['fourth_b'], torch.full((2,), 2.0))


def test_gene_layer_split_default_follows_what_it_saves():
    from kgwas_amd import ops
    assert not ops.gene_layer_split_pays(1, 57742) and not ops.gene_layer_split_pays(2, 5120)
    assert ops.gene_layer_split_pays(4, 5120) and ops.gene_layer_split_pays(8, 5120)
    assert ops.gene_layer_split_pays(2, 57742)                # the 57 742-wide features: from two ranks on
    assert not ops.gene_layer_split_pays(8, 96)               # narrow features never take the resident route anyway
    # the decision table as documented -- width 5 120: from FOUR ranks (three is a wash), width unknown: the same rule
    assert not ops.gene_layer_split_pays(3, 5120)
    assert [ops.gene_layer_split_pays(w, 0) for w in (1, 2, 3, 4, 8)] == [False, False, False, True, True]
