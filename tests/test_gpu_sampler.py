"""-m gpu: kgw_sample_batch against the PyG NeighborLoader([-1]*L) restatement (oracle/pyg_semantics.py).
Integer work => exact comparisons."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import pyg_semantics as P
from tests.helpers import global_edge_set

pytestmark = pytest.mark.gpu


def _loader(data, ids, bs, L=2, **kw):
    from kgwas_amd.sampler import NeighborLoader
    return NeighborLoader(data, num_neighbors=[-1] * L, input_nodes=('SNP', ids), batch_size=bs, device='cuda:0', **kw)


def _oracle_sets(data, seeds, L=2):
    smp = P.FullNeighborSampler(data.edge_index_dict, data.num_nodes_dict, L)
    n_id, ei, hops = smp.sample('SNP', seeds)
    edges = {}
    for et, e in ei.items():
        s, _, d = et
        pairs = np.stack([n_id[s].numpy()[e[0].numpy()], n_id[d].numpy()[e[1].numpy()]], axis=1) if e.shape[1] \
            else np.zeros((0, 2), np.int64)
        order = np.lexsort((pairs[:, 0], pairs[:, 1]))
        edges[et] = pairs[order]
    return n_id, edges, hops


@pytest.mark.parametrize('which', ['small', 'edge'])
@pytest.mark.parametrize('L', [1, 2, 3])
def test_sampler_matches_pyg_semantics(small_kg, edge_case_graph, which, L):
    data = small_kg.data if which == 'small' else edge_case_graph[0]
    n_snp = data['SNP'].x.shape[0]
    rng = np.random.default_rng(5)
    ids = rng.choice(n_snp, size=96, replace=False)
    ld = _loader(data, ids, 32, L=L)
    assert len(ld) == 3
    for b, batch in enumerate(ld):
        seeds = ids[b * 32:(b + 1) * 32]
        n_id_o, edges_o, hops_o = _oracle_sets(data, seeds, L)
        assert batch['SNP'].batch_size == 32
        # seeds first, in seed order (the only ordering contract the reference consumes, model.py:86)
        assert np.array_equal(batch.n_id('SNP')[:32].cpu().numpy(), seeds)
        for t in data.node_types:
            mine = batch.n_id(t).cpu().numpy()
            assert len(np.unique(mine)) == len(mine), f'duplicate nodes in {t}'
            assert np.array_equal(np.sort(mine), np.sort(n_id_o[t].numpy())), f'node set of {t} differs'
            # hop-major, ascending global id inside a hop (beyond the seeds)
            m = batch.meta
            ti = batch.dg.schema.type_id[t]
            for k in range(1, L + 1):
                a, bnd = int(m.node_off[ti][k]), int(m.node_off[ti][k + 1])
                seg = mine[a:bnd]
                assert np.all(np.diff(seg) > 0), f'{t} hop {k} not sorted'
                ref_hop = np.sort(n_id_o[t].numpy()[hops_o[t].numpy() == k])
                assert np.array_equal(seg, ref_hop), f'{t} hop {k} membership differs'
        for et in data.edge_types:
            assert np.array_equal(global_edge_set(batch, et), edges_o[et]), f'edge multiset of {et} differs'
        assert batch.n_edges_sampled == sum(len(v) for v in edges_o.values())


@pytest.mark.parametrize('which', ['small', 'edge'])
def test_block_structures_are_consistent(small_kg, edge_case_graph, which):
    """Chunk list covers every local edge exactly once; multi-chunk segments are listed; the src-major
    structure of every layer is a permutation of the layer's live edges grouped by (source row, slot)."""
    from kgwas_amd._lib import KGW_CHUNK
    if which == 'small':
        data = small_kg.data
        ids = np.random.default_rng(0).choice(data['SNP'].x.shape[0], size=512, replace=False)
    else:   # seeds around the 1500-edge hub gene 0
        data = edge_case_graph[0]
        ids = np.random.default_rng(0).choice(1500, size=64, replace=False)
    batch = next(iter(_loader(data, ids, len(ids), L=2)))
    dg, m, buf = batch.dg, batch.meta, batch.buf
    sc = dg.schema
    L = dg.num_layers
    n_chunks_all = int(m.chunk_end[L - 1])
    n_edges_all = int(m.edge_end[L - 1])
    ch = buf.chunks[:n_chunks_all * 8].view(-1, 8).cpu().numpy()
    col = buf.col_local[:n_edges_all].cpu().numpy()
    assert n_chunks_all > 0 and n_edges_all > 0
    # chunks tile [0, n_edges) in order
    assert ch[0, 0] == 0 and ch[-1, 1] == n_edges_all
    assert np.array_equal(ch[1:, 0], ch[:-1, 1])
    assert np.all(ch[:, 1] - ch[:, 0] <= KGW_CHUNK) and np.all(ch[:, 1] > ch[:, 0])
    # hub rows exist in this graph -> multi-chunk segments recorded
    n_multi = sum(int(m.multi_cnt[h]) for h in range(L))
    assert n_multi == len(np.unique(ch[ch[:, 5] > 1][:, 4]))
    assert n_multi > 0, 'test graph should contain rows above KGW_CHUNK edges'
    for l in range(1, L + 1):
        nc, ne = int(m.n_chunks[l - 1]), int(m.n_edges[l - 1])
        live = np.array([dg.kg.rel_live[l - 1][r] for r in range(sc.NR)], dtype=bool)
        chl = ch[:nc]
        chl = chl[live[chl[:, 3]]]
        n_live_edges = int((chl[:, 1] - chl[:, 0]).sum())
        assert int(m.t_entries[l - 1]) == n_live_edges
        t_rows = int(m.t_base[l - 1][sc.NT])
        tptr = buf.t_ptr[l - 1][:t_rows + 1].cpu().numpy()
        tedge = buf.t_edge[l - 1][:n_live_edges].cpu().numpy()
        tz = buf.t_zrow[l - 1][:n_live_edges].cpu().numpy()
        trel = buf.t_rel[l - 1][:n_live_edges].cpu().numpy()
        assert tptr[0] == 0 and tptr[-1] == n_live_edges and np.all(np.diff(tptr) >= 0)
        # permutation of the live edge ids
        expect = np.concatenate([np.arange(a, b) for a, b in chl[:, :2]]) if len(chl) else np.zeros(0, np.int64)
        assert np.array_equal(np.sort(tedge), np.sort(expect))
        # every entry sits in the row of its (source, slot) and carries its destination Z row
        e2chunk = np.searchsorted(ch[:, 1], tedge, side='right')
        rel = ch[e2chunk, 3]
        row = ch[e2chunk, 2]
        src_t = sc.src_type[rel]
        dst_t = sc.dst_type[rel]
        tb = np.array([m.t_base[l - 1][t] for t in range(sc.NT + 1)])
        zb = np.array([m.z_base[l - 1][t] for t in range(sc.NT + 1)])
        trow = tb[src_t] + col[tedge] * sc.R_src[src_t] + sc.slot_src[rel]
        pos = np.arange(n_live_edges)
        assert np.all(tptr[trow] <= pos) and np.all(pos < tptr[trow + 1])
        assert np.array_equal(tz, zb[dst_t] + row * sc.R_dst[dst_t] + sc.slot_dst[rel])
        assert np.array_equal(buf.t_rel[l - 1][:n_live_edges].cpu().numpy(), rel)      # relation id per entry
        # octet flags (the backward's 8-rows-per-wavefront path): set exactly for the groups of 8 real source rows of one
        # short-row type that hold no destination row and no row above 8 entries
        n_src_rows = int(m.src_base[l - 1][sc.NT])
        flags = buf.t_cnt[l - 1][:(n_src_rows + 7) // 8].cpu().numpy()
        sb = np.array([m.src_base[l - 1][t] for t in range(sc.NT + 1)])
        for o in range(len(flags)):
            u0 = 8 * o
            ty = int(np.searchsorted(sb[1:], u0, side='right'))
            j0 = u0 - sb[ty]
            ok = bool((dg.short_type_mask >> ty) & 1) and j0 + 8 <= int(m.n_src[l - 1][ty]) and \
                not (sc.R_dst[ty] > 0 and j0 < int(m.n_rows[l - 1][ty]))
            if ok:
                Rs = int(sc.R_src[ty])
                t0 = tb[ty] + j0 * Rs
                ok = all(tptr[t0 + (q + 1) * Rs] - tptr[t0 + q * Rs] <= 8 for q in range(8))
            assert bool(flags[o]) == ok, (l, o)
        assert ne <= n_edges_all
        # deterministic order: inside every src-major row the entries ascend by edge id (the structure is the edge list
        # STABLY sorted by row: k_ts_scatter / k_ts_rows rank equal keys by lane order, nothing depends on arrival order)
        row_of = np.repeat(np.arange(t_rows), np.diff(tptr))
        same = row_of[1:] == row_of[:-1]
        assert np.all(tedge[1:][same] > tedge[:-1][same])


def test_full_graph_block(edge_case_graph):
    """full_graph mode: identity node maps, every edge of every relation exactly once."""
    from kgwas_amd.sampler import sample_full_graph
    data = edge_case_graph[0]
    batch = sample_full_graph(data, 2, 'cuda:0')
    for t in data.node_types:
        n = data[t].x.shape[0]
        assert np.array_equal(batch.n_id(t).cpu().numpy(), np.arange(n))
    tot = 0
    for et in data.edge_types:
        ei = data[et].edge_index.numpy()
        got = global_edge_set(batch, et)
        order = np.lexsort((ei[0], ei[1]))
        assert np.array_equal(got, ei.T[order])
        tot += ei.shape[1]
    assert batch.n_edges_sampled == tot


def test_loader_contract(small_kg):
    """len / drop_last / prefetch on-off give identical batches (kgwas.py:93-94: fixed order, no shuffle)."""
    data = small_kg.data
    ids = np.asarray(small_kg.train_input_nodes[1][:200])
    a = _loader(data, ids, 64, drop_last=True)
    b = _loader(data, ids, 64, drop_last=False, prefetch=False)
    assert len(a) == 3 and len(b) == 4
    sizes = []
    for x, y in zip(a, b):
        assert np.array_equal(x.n_id('SNP').cpu().numpy(), y.n_id('SNP').cpu().numpy())
        assert x.n_edges_sampled == y.n_edges_sampled
        assert torch.equal(x['SNP'].y[:64], data['SNP'].y[x['SNP']['n_id'][:64].cpu()].cuda())
        sizes.append(x.n_edges_sampled)
    assert [bb['SNP'].batch_size for bb in b] == [64, 64, 64, 8]
    xd = next(iter(a)).x_dict
    assert set(xd.keys()) == set(data.node_types)


def test_loader_on_cached_graph_equals_direct(tiny_kg, tmp_path):
    """kgwas_amd/ingest.py: a DeviceGraph fed from the on-disk CSR cache samples exactly what one built from COO does."""
    from kgwas_amd import ingest
    from kgwas_amd.kgwas_data import KGWAS_Data
    cache = ingest.convert(tiny_kg, str(tmp_path / 'cache'))
    other = ingest.load(KGWAS_Data(str(tmp_path / 'dp')), cache)
    ids = np.random.default_rng(5).choice(tiny_kg.data['SNP'].x.shape[0], size=32, replace=False)
    a = next(iter(_loader(tiny_kg.data, ids, 32, L=2)))
    b = next(iter(_loader(other.data, ids, 32, L=2)))
    for t in tiny_kg.data.node_types:
        assert torch.equal(a.n_id(t), b.n_id(t))
        assert torch.equal(a.x_dict[t], b.x_dict[t])
    ea, eb = a.edge_index_dict, b.edge_index_dict
    for et in ea:
        assert torch.equal(ea[et], eb[et])


@pytest.mark.parametrize('shift', [8, 12, 14])
def test_coarse_bucket_sort_plans_build_the_same_structures(shift):
    """the src-major sort's coarse plans (2^12 rows per bucket, four wavefronts; 2^14, one wavefront) launch
    k_ts_rows with 64 KB of dynamic LDS + its static words -- above the 64 KB default limit -- and are only chosen above
    8 M / 65 M src-major rows, which no test graph has.  KGW_TS_MIN_SHIFT forces them (read once per process, hence the
    subprocess); the structure checks above must hold unchanged.  (8: the fine buckets that were the default until round 6 --
    the default is 2^11 rows now.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KGW_TS_MIN_SHIFT=str(shift))
    p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', os.path.join(root, 'tests', 'test_gpu_sampler.py'),
                        '-k', 'block_structures or matches_pyg_semantics'], cwd=root, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    assert ' passed' in p.stdout
