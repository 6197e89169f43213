"""Graph ingest: the reference's pickled knowledge graph -> an mmap-able on-disk cache in the layout the sampler
consumes.

The reference keeps the KG as Python pickles -- ``cell_kg/network/{node_idx2id,node_id2idx,edge_index}.pkl``
(dict-of-dicts / dict of [2,E] lists) and per-embedding ``{id: vector}`` dicts -- and rebuilds tensors from them with
per-node Python loops on every start (kgwas/kgwas_data.py:112-273, e.g. ``torch.vstack([... for i in
range(len(node_map))])`` at :144-145,181-182,234-235), then applies ``ToUndirected`` + ``AddSelfLoops`` (:271-272).
``convert`` does that ONCE (with this package's vectorised ``KGWAS_Data.load_kg``) and writes

    <cache>/meta.json                       node / edge types, counts, feature widths, the load_kg options
    <cache>/x_<type>.npy                    contiguous float32 feature matrix per node type
    <cache>/rowptr_<k>.npy, col_<k>.npy     per relation k (after ToUndirected + AddSelfLoops): the dst-major CSR
                                            of kgwas_amd.graph.build_csr (int64 row pointers, int32 sources sorted
                                            inside a row = PyG to_csc order) -- what DeviceGraph uploads
    <cache>/ids_<type>.npy                  node id strings in index order (idx2id / id2idx are rebuilt from them)

``load`` maps the arrays (``np.load(mmap_mode=...)``) into a ``KGWAS_Data``: no sorting, no Python loops; the COO
``edge_index`` tensors of the reference's HeteroData duck-type are rebuilt from the CSR with two vectorised calls,
and ``DeviceGraph`` uploads the cached CSR as it is.
The unseeded ``torch.rand`` fill-ins of the reference are seeded (``seed``) so that a cache and a direct load agree."""
from __future__ import annotations

import json
import os
from collections import OrderedDict

import numpy as np
import torch

from .graph import HeteroGraph, build_csr

CACHE_VERSION = 2


def source_fingerprint(paths) -> dict:
    """{path: [size, mtime_ns]} of the files a cache was converted from: a cache whose sources changed is stale."""
    out = {}
    for p in paths:
        if p and os.path.exists(p):
            st = os.stat(p)
            out[p] = [int(st.st_size), int(st.st_mtime_ns)]
    return out


def _fname(kind: str, key: str) -> str:
    return f'{kind}_{key}.npy'


def convert(kg_data, cache_dir: str, options: dict = None, sources: dict = None) -> str:
    """Write the cache for a loaded ``KGWAS_Data`` (after ``load_kg`` / ``from_synthetic``).  Returns ``cache_dir``.
    ``sources``: ``source_fingerprint`` of the files the graph was read from (kept in meta.json; ``load_kg`` rejects
    the cache when it no longer matches)."""
    import pickle
    data: HeteroGraph = kg_data.data
    os.makedirs(cache_dir, exist_ok=True)
    node_types = list(data.node_types)
    edge_types = [tuple(e) for e in data.edge_types]
    nn = {t: int(data[t].num_nodes) for t in node_types}
    for t in node_types:
        np.save(os.path.join(cache_dir, _fname('x', t)), np.ascontiguousarray(data[t].x.numpy(), dtype=np.float32))
        idmap = kg_data.idx2id[t]
        ids = np.asarray([str(idmap[i]) for i in range(nn[t])])
        np.save(os.path.join(cache_dir, _fname('ids', t)), ids)
    n_edges = []
    for k, et in enumerate(edge_types):
        rp, col = build_csr(data[et].edge_index, nn[et[0]], nn[et[2]])
        np.save(os.path.join(cache_dir, _fname('rowptr', str(k))), rp)
        np.save(os.path.join(cache_dir, _fname('col', str(k))), col)
        n_edges.append(int(rp[-1]))
    meta = {'version': CACHE_VERSION, 'node_types': node_types, 'num_nodes': nn,
            'edge_types': [list(e) for e in edge_types], 'num_edges': n_edges,
            'feat_dims': {'snp': int(kg_data.snp_init_dim_size), 'gene': int(kg_data.gene_init_dim_size),
                          'go': int(kg_data.go_init_dim_size)},
            'options': dict(options or {}), 'sources': dict(sources or {})}
    # the id maps exactly as the reference's pickles hold them (key types, alias keys of node_id2idx.pkl that idx2id
    # lacks): a cached load must resolve GWAS ids like the first, uncached one
    maps = {}
    for name in ('idx2id', 'id2idx'):
        m = getattr(kg_data, name, None)
        if isinstance(m, dict) and all(isinstance(v, dict) for v in m.values()):
            maps[name] = m
    if len(maps) == 2:
        with open(os.path.join(cache_dir, 'idmaps.pkl'), 'wb') as f:
            pickle.dump(maps, f, protocol=pickle.HIGHEST_PROTOCOL)
    with open(os.path.join(cache_dir, 'meta.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    return cache_dir


def read_meta(cache_dir: str):
    p = os.path.join(cache_dir, 'meta.json')
    if not os.path.exists(p):
        return None
    with open(p) as f:
        m = json.load(f)
    return m if m.get('version') == CACHE_VERSION else None


class _IdMap:
    """idx -> id view over the cached string array (``idx2id[type]``)."""

    def __init__(self, ids):
        self._ids = ids

    def __len__(self):
        return len(self._ids)

    def __getitem__(self, i):
        return str(self._ids[int(i)])

    def __contains__(self, i):
        return 0 <= int(i) < len(self._ids)

    def values(self):
        return [str(v) for v in self._ids]

    def keys(self):
        return range(len(self._ids))

    def items(self):
        return ((i, str(v)) for i, v in enumerate(self._ids))


def load(kg_data, cache_dir: str, mmap: bool = True):
    """Fill ``kg_data`` (a fresh ``KGWAS_Data``) from a cache: ``.data`` (HeteroGraph with resident CSR), ``.idx2id``,
    ``.id2idx``, the three ``*_init_dim_size``.  Returns ``kg_data``."""
    meta = read_meta(cache_dir)
    if meta is None:
        raise FileNotFoundError(f'no KG cache (version {CACHE_VERSION}) in {cache_dir}')
    mode = 'r' if mmap else None
    data = HeteroGraph()
    idx2id, id2idx = {}, {}
    for t in meta['node_types']:
        x = np.load(os.path.join(cache_dir, _fname('x', t)), mmap_mode='c' if mmap else None)   # copy-on-write mapping
        data[t].x = torch.from_numpy(x)
        ids = np.load(os.path.join(cache_dir, _fname('ids', t)), mmap_mode=None)
        idx2id[t] = _IdMap(ids)
        id2idx[t] = {str(v): i for i, v in enumerate(ids)}
    csr = OrderedDict()
    for k, et in enumerate(meta['edge_types']):
        et = tuple(et)
        rp = np.load(os.path.join(cache_dir, _fname('rowptr', str(k))), mmap_mode=mode)
        col = np.load(os.path.join(cache_dir, _fname('col', str(k))), mmap_mode=mode)
        csr[et] = (rp, col)
        # the reference's HeteroData duck-type exposes COO per relation: rebuilt from the CSR with two vectorised calls
        dst = np.repeat(np.arange(len(rp) - 1, dtype=np.int64), np.diff(np.asarray(rp)))
        data[et].edge_index = torch.from_numpy(np.stack([np.asarray(col, dtype=np.int64), dst]))
    data._extra['csr'] = csr
    kg_data.data = data
    mp = os.path.join(cache_dir, 'idmaps.pkl')
    if os.path.exists(mp):                       # the original maps, key types and aliases intact
        import pickle
        with open(mp, 'rb') as f:
            maps = pickle.load(f)
        idx2id, id2idx = maps['idx2id'], maps['id2idx']
    kg_data.idx2id, kg_data.id2idx = idx2id, id2idx
    kg_data.snp_init_dim_size = meta['feat_dims']['snp']
    kg_data.gene_init_dim_size = meta['feat_dims']['gene']
    kg_data.go_init_dim_size = meta['feat_dims']['go']
    return kg_data
