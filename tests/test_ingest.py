"""Row f-3 (graph ingest): the reference's pickle layout (kgwas/kgwas_data.py:112-273) -> on-disk cache -> identical
graph.  CPU only."""
import os
import pickle

import numpy as np
import torch

from kgwas_amd.graph import build_csr
from kgwas_amd.kgwas_data import KGWAS_Data


def _write_reference_pickles(dp, rng):
    """A tiny KG in the reference's file layout: node_idx2id / node_id2idx / edge_index pickles + the embedding
    dicts load_kg reads for the default options (enformer SNP, esm gene, random GO)."""
    n = {'SNP': 60, 'Gene': 9, 'CellularComponent': 4, 'BiologicalProcess': 5, 'MolecularFunction': 3}
    pre = {'SNP': 'rs', 'Gene': 'ENSG', 'CellularComponent': 'CC:', 'BiologicalProcess': 'BP:', 'MolecularFunction': 'MF:'}
    idx2id = {t: {i: f'{pre[t]}{i}' for i in range(k)} for t, k in n.items()}
    id2idx = {t: {v: i for i, v in m.items()} for t, m in idx2id.items()}
    e = {
        ('SNP', 'ABC', 'Gene'): np.stack([rng.integers(0, 60, 80), rng.integers(0, 9, 80)]),
        ('SNP', 'TSS', 'Gene'): np.stack([np.arange(60), np.arange(60) % 9]),
        ('Gene', 'Gene-Literature-Gene', 'Gene'): np.stack([rng.integers(0, 9, 20), rng.integers(0, 9, 20)]),
        ('Gene', 'Gene-Associates-BiologicalProcess', 'BiologicalProcess'): np.stack([rng.integers(0, 9, 12), rng.integers(0, 5, 12)]),
        ('Gene', 'Gene-Colocalizes-CellularComponent', 'CellularComponent'): np.stack([rng.integers(0, 9, 7), rng.integers(0, 4, 7)]),
        ('Gene', 'Gene-Contributes-MolecularFunction', 'MolecularFunction'): np.stack([rng.integers(0, 9, 5), rng.integers(0, 3, 5)]),
    }
    net = os.path.join(dp, 'cell_kg', 'network')
    os.makedirs(net, exist_ok=True)
    for name, obj in (('node_idx2id.pkl', idx2id), ('node_id2idx.pkl', id2idx),
                      ('edge_index.pkl', {k: v.tolist() for k, v in e.items()})):
        with open(os.path.join(net, name), 'wb') as f:
            pickle.dump(obj, f)
    return n, idx2id


def _emb_files(dp, idx2id, rng):
    """Embedding dicts for most (not all) nodes, at the paths / widths KGWAS_Data.load_kg expects."""
    from kgwas_amd import kgwas_data as kd
    for table, t, name in ((kd._SNP_EMB, 'SNP', 'enformer'), (kd._GENE_EMB, 'Gene', 'esm')):
        path, dim = table[name]
        full = os.path.join(dp, path)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        feat = {v: rng.random(dim).astype(np.float32) for i, v in idx2id[t].items() if i % 7 != 3}
        with open(full, 'wb') as f:
            pickle.dump(feat, f)


def test_cache_round_trip_equals_direct_load(tmp_path):
    rng = np.random.default_rng(0)
    dp = str(tmp_path)
    n, idx2id = _write_reference_pickles(dp, rng)
    _emb_files(dp, idx2id, rng)

    a = KGWAS_Data(dp)
    a.load_kg(cache=False)                         # the direct path (pickles -> tensors), no cache written
    assert not os.path.exists(os.path.join(dp, 'cell_kg', 'kgwas_amd_cache'))
    b = KGWAS_Data(dp)
    b.load_kg()                                    # converts and writes the cache
    c = KGWAS_Data(dp)
    c.load_kg()                                    # served from the cache
    assert '_extra' in c.data.__dict__ and 'csr' in c.data._extra

    for other in (b, c):
        assert list(other.data.node_types) == list(a.data.node_types)
        assert [tuple(e) for e in other.data.edge_types] == [tuple(e) for e in a.data.edge_types]
        assert (other.snp_init_dim_size, other.gene_init_dim_size, other.go_init_dim_size) == \
            (a.snp_init_dim_size, a.gene_init_dim_size, a.go_init_dim_size)
        for t in a.data.node_types:
            assert torch.equal(other.data[t].x, a.data[t].x)
            assert all(other.idx2id[t][i] == a.idx2id[t][i] for i in range(n[t]))
            assert all(other.id2idx[t][a.idx2id[t][i]] == i for i in range(n[t]))
        for et in a.data.edge_types:
            ns, nd = n[et[0]], n[et[2]]
            rpa, cola = build_csr(a.data[et].edge_index, ns, nd)
            rpo, colo = build_csr(other.data[et].edge_index, ns, nd)
            assert np.array_equal(rpa, rpo) and np.array_equal(cola, colo)          # same edge multiset per relation
    # the cached CSR is exactly what DeviceGraph would build
    for et, (rp, col) in c.data._extra['csr'].items():
        rpa, cola = build_csr(a.data[et].edge_index, n[et[0]], n[et[2]])
        assert np.array_equal(np.asarray(rp), rpa) and np.array_equal(np.asarray(col), cola)
    # different options -> different cache entry, not a stale hit
    d = KGWAS_Data(dp)
    d.load_kg(sample_edges=True, sample_ratio=0.5)
    assert sum(int(d.data[et].edge_index.shape[1]) for et in d.data.edge_types) < \
        sum(int(a.data[et].edge_index.shape[1]) for et in a.data.edge_types)
