"""CPU (world_size 2, gloo): the algorithm of the SNP-sharded mode (kgwas_amd/shard.py) restated on the
float64 oracle -- rank-local graphs, the hop-1 frontier union, the partial online-softmax states (m, s, sum exp(e-m) h)
merged in rank order, and the backward rule (complete dZ for the sharded edges, partial everywhere else, SUM of the
parameter gradients) -- against the unsharded oracle.  The HIP kernels of the same mode are tested on the GPU in
tests/test_gpu_shard.py; the host-side graph partition is tested here without any process group."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    """One Gene<-SNP relation of the committed tiny case: H (float64), u, v, edges, an upstream gradient."""
    from tests.golden import gat_case as gc
    e = gc.original_edges()[('SNP', 'ABC', 'Gene')]
    n_s, n_d, C = gc.NODES['SNP'], gc.NODES['Gene'], 16
    Hs = torch.from_numpy(gc.hash01(n_s * C, 11).reshape(n_s, C)) - 0.5
    Hd = torch.from_numpy(gc.hash01(n_d * C, 12).reshape(n_d, C)) - 0.5
    u = torch.from_numpy(gc.hash01(C, 13)) - 0.5
    v = torch.from_numpy(gc.hash01(C, 14)) - 0.5
    dZ = torch.from_numpy(gc.hash01(n_d * C, 15).reshape(n_d, C)) - 0.5
    return torch.from_numpy(e), Hs, Hd, u, v, dZ, n_s, n_d


def _full(e, Hs, Hd, u, v, n_d):
    from oracle.pyg_semantics import segment_softmax
    src, dst = e[0], e[1]
    logit = torch.nn.functional.leaky_relu((Hs @ u)[src] + (Hd @ v)[dst], 0.2)
    alpha = segment_softmax(logit, dst, n_d)
    return torch.zeros(n_d, Hs.shape[1], dtype=Hs.dtype).index_add(0, dst, alpha.unsqueeze(1) * Hs[src])


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from kgwas_amd.shard import shard_range
        e, Hs, Hd, u, v, dZ_full, n_s, n_d = _case()
        lo, hi = shard_range(n_s, rank, world)
        keep = (e[0] >= lo) & (e[0] < hi)
        src, dst = e[0][keep], e[1][keep]
        Hs = Hs.clone().requires_grad_(True)
        u = u.clone().requires_grad_(True)
        logit = torch.nn.functional.leaky_relu((Hs @ u)[src] + (Hd @ v)[dst], 0.2)
        # partial online-softmax state of this rank (what k_agg_fwd leaves under KgwLayerArgs.partial_rels)
        m = torch.full((n_d,), -1e30, dtype=Hs.dtype).scatter_reduce(0, dst, logit.detach(), reduce='amax', include_self=True)
        w = (logit - m[dst]).exp()
        s = torch.zeros(n_d, dtype=Hs.dtype).index_add(0, dst, w)
        acc = torch.zeros(n_d, Hs.shape[1], dtype=Hs.dtype).index_add(0, dst, w.unsqueeze(1) * Hs[src])
        state = torch.cat([m[:, None], s[:, None], acc.detach()], 1)
        states = [torch.empty_like(state) for _ in range(world)]
        dist.all_gather(states, state)
        # merge in rank order (k_softmax_merge)
        M = torch.full((n_d,), -1e30, dtype=Hs.dtype)
        for st in states:
            M = torch.where(st[:, 1] > 0, torch.maximum(M, st[:, 0]), M)
        S = torch.zeros(n_d, dtype=Hs.dtype)
        V = torch.zeros(n_d, Hs.shape[1], dtype=Hs.dtype)
        for st in states:
            f = torch.where(st[:, 1] > 0, (st[:, 0] - M).exp(), torch.zeros_like(M))
            S += st[:, 1] * f
            V += st[:, 2:] * f[:, None]
        Z = V / (S + 1e-16)[:, None]
        # backward: this rank's upstream gradient is PARTIAL (here: a 1/world share); its own edges need the sum
        dZ = dZ_full / world
        dist.all_reduce(dZ)                                         # == dZ_full on every rank
        # the merged result as a function of THIS rank's edges, the other ranks' contributions being constants: exactly
        # what the backward kernels differentiate (d alpha_ij = alpha_ij (<dZ_i, h_j> - <dZ_i, Z_i>) with the MERGED
        # statistics and the merged Z_i; the max is a constant like PyG's detached one)
        f = torch.where(s > 0, (m - M).exp(), torch.zeros_like(M))
        V_other = V - acc.detach() * f[:, None]
        S_other = S - s.detach() * f
        Z_mine = (acc * f[:, None] + V_other) / (s * f + S_other + 1e-16)[:, None]
        assert torch.allclose(Z_mine.detach(), Z, rtol=0, atol=1e-14)
        (Z_mine * dZ).sum().backward()
        torch.save({'Z': Z.detach(), 'dHs': Hs.grad.clone(), 'du': u.grad.clone(), 'lo': lo, 'hi': hi},
                   os.path.join(out_dir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_partial_softmax_merge_and_backward_rule_two_ranks(tmp_path):
    world = 2
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    recs = [torch.load(os.path.join(tmp_path, f'rank{r}.pt')) for r in range(world)]
    e, Hs, Hd, u, v, dZ, n_s, n_d = _case()
    Hs = Hs.clone().requires_grad_(True)
    u = u.clone().requires_grad_(True)
    Zf = _full(e, Hs, Hd, u, v, n_d)
    (Zf * dZ).sum().backward()
    for r in recs:
        assert torch.allclose(r['Z'], Zf.detach(), rtol=0, atol=1e-13)            # hub row (300 in-edges) split over both ranks
    assert torch.equal(recs[0]['Z'], recs[1]['Z'])                                # fixed merge order: identical bits
    # source-row gradients: each rank holds the rows it owns; attention-vector gradient: the SUM over the ranks
    dHs = torch.zeros_like(Hs)
    for r in recs:
        dHs[r['lo']:r['hi']] = r['dHs'][r['lo']:r['hi']]
        outside = r['dHs'].clone(); outside[r['lo']:r['hi']] = 0
        assert float(outside.abs().max()) == 0.0
    assert torch.allclose(dHs, Hs.grad, rtol=1e-10, atol=1e-13)
    assert torch.allclose(recs[0]['du'] + recs[1]['du'], u.grad, rtol=1e-10, atol=1e-13)


def test_shard_graph_partitions_the_edges():
    """Host side: the rank-local graphs keep every edge exactly once on the sharded side and whole on the replicated side."""
    from kgwas_amd.kgwas_data import KGWAS_Data
    from kgwas_amd.shard import shard_graph, shard_range
    data = KGWAS_Data.from_synthetic(scale=0.005, seed=2, feat_dims={'Gene': 16}, data_path='/tmp/kgwas_shard_host')
    g = data.data
    world = 3
    n = int(g['SNP'].x.shape[0])
    tot = {et: 0 for et in g.edge_types}
    for r in range(world):
        loc, lo, hi = shard_graph(g, r, world)
        assert (lo, hi) == shard_range(n, r, world) and loc['SNP'].x.shape[0] == hi - lo
        assert torch.equal(loc['SNP'].x, g['SNP'].x[lo:hi]) and torch.equal(loc['SNP'].y, g['SNP'].y[lo:hi])
        for et in g.edge_types:
            e = loc[et].edge_index
            if 'SNP' in (et[0], et[2]):
                tot[et] += e.shape[1]
                col = 0 if et[0] == 'SNP' else 1
                assert e.shape[1] == 0 or (int(e[col].min()) >= 0 and int(e[col].max()) < hi - lo)
            else:
                assert torch.equal(e, g[et].edge_index)
    for et in g.edge_types:
        if 'SNP' in (et[0], et[2]):
            assert tot[et] == g[et].edge_index.shape[1], et
