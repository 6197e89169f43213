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
