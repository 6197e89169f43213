"""Heterogeneous knowledge-graph container and one-off graph build.

``HeteroGraph`` is the duck-type of the ``torch_geometric.data.HeteroData`` surface the reference
touches (kgwas/kgwas_data.py:129,259-272,532-539; kgwas/kgwas.py:63,97-113; kgwas/model.py:26):
``graph[node_type].x / .y / .n_id``, ``graph[edge_type].edge_index``, ``.node_types``,
``.edge_types``, ``.x_dict``, ``.edge_index_dict``, ``.metadata()``, ``.to(device)``.

The graph-build semantics consumed by the hot path are implemented here with
vectorised numpy: ``to_undirected`` == T.ToUndirected() (kgwas_data.py:271), ``add_self_loops`` ==
T.AddSelfLoops() (kgwas_data.py:272), and ``build_csr`` == the (dst, src)-sorted CSC that PyG's
NeighborLoader builds once per edge type (kgwas/kgwas.py:99-113).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch

EdgeType = Tuple[str, str, str]


class _Store(dict):
    """Attribute bag (``store.x`` == ``store['x']``) like a PyG storage object."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @property
    def num_nodes(self):
        if 'x' in self:
            return int(self['x'].shape[0])
        if 'num_nodes_' in self:
            return int(self['num_nodes_'])
        raise AttributeError('num_nodes')


class HeteroGraph:
    def __init__(self):
        self._nodes: "OrderedDict[str, _Store]" = OrderedDict()
        self._edges: "OrderedDict[EdgeType, _Store]" = OrderedDict()
        self._extra = {}

    # -- HeteroData-like access ---------------------------------------------------------------
    def __getitem__(self, key):
        if isinstance(key, tuple):
            key = tuple(key)
            if key not in self._edges:
                self._edges[key] = _Store()
            return self._edges[key]
        if key not in self._nodes:
            self._nodes[key] = _Store()
        return self._nodes[key]

    def __setattr__(self, k, v):
        if k.startswith('_'):
            object.__setattr__(self, k, v)
        else:
            self._extra[k] = v   # e.g. data.train_mask = ...  (kgwas_data.py:541-544)

    def __getattr__(self, k):
        extra = object.__getattribute__(self, '_extra')
        if k in extra:
            return extra[k]
        raise AttributeError(k)

    @property
    def node_types(self) -> List[str]:
        return list(self._nodes.keys())

    @property
    def edge_types(self) -> List[EdgeType]:
        return [k for k, v in self._edges.items() if 'edge_index' in v]

    def metadata(self):
        return self.node_types, self.edge_types

    @property
    def x_dict(self):
        return {k: v['x'] for k, v in self._nodes.items() if 'x' in v}

    @property
    def edge_index_dict(self):
        return OrderedDict((k, v['edge_index']) for k, v in self._edges.items() if 'edge_index' in v)

    @property
    def num_nodes_dict(self) -> Dict[str, int]:
        return {k: v.num_nodes for k, v in self._nodes.items()}

    def to(self, device, *args, **kwargs):
        for st in list(self._nodes.values()) + list(self._edges.values()):
            for k, v in list(st.items()):
                if torch.is_tensor(v):
                    st[k] = v.to(device)
        return self


# --------------------------------------------------------------------------------------------
# graph transforms (numpy, int64 in / int64 out)
# --------------------------------------------------------------------------------------------
def _as_np(ei) -> np.ndarray:
    if torch.is_tensor(ei):
        ei = ei.cpu().numpy()
    return np.asarray(ei, dtype=np.int64).reshape(2, -1)


def to_undirected(edge_index_dict, num_nodes: Dict[str, int]):
    """T.ToUndirected() on a hetero graph: bipartite relations gain ``rev_<rel>`` mirrors appended
    after all original types; same-type relations are symmetrised + coalesced (sorted by (row,col),
    duplicates dropped)."""
    out: "OrderedDict[EdgeType, np.ndarray]" = OrderedDict()
    rev: "OrderedDict[EdgeType, np.ndarray]" = OrderedDict()
    for (s, rel, d), ei in edge_index_dict.items():
        ei = _as_np(ei)
        if s != d:
            out[(s, rel, d)] = ei
            rev[(d, 'rev_' + rel, s)] = ei[::-1].copy()
        else:
            n = int(num_nodes[s])
            key = np.concatenate([ei[0] * n + ei[1], ei[1] * n + ei[0]])
            key = np.unique(key)
            out[(s, rel, d)] = np.stack([key // n, key % n])
    out.update(rev)
    return out


def add_self_loops(edge_index_dict, num_nodes: Dict[str, int]):
    """T.AddSelfLoops(): append N (i,i) loops to every same-type relation (existing loops kept)."""
    out: "OrderedDict[EdgeType, np.ndarray]" = OrderedDict()
    for (s, rel, d), ei in edge_index_dict.items():
        ei = _as_np(ei)
        if s == d:
            loop = np.arange(int(num_nodes[s]), dtype=np.int64)
            ei = np.concatenate([ei, np.stack([loop, loop])], axis=1)
        out[(s, rel, d)] = ei
    return out


def build_csr(edge_index, n_src: int, n_dst: int):
    """dst-major CSR of one relation: rows = destination nodes, entries = source ids sorted
    ascending inside a row (PyG ``to_csc`` order).  Returns (rowptr int64[n_dst+1], col int32[E])."""
    ei = _as_np(edge_index)
    src, dst = ei[0], ei[1]
    if src.size and (src.min() < 0 or src.max() >= n_src or dst.min() < 0 or dst.max() >= n_dst):
        raise ValueError('edge_index out of range')
    order = np.argsort(dst * np.int64(n_src) + src, kind='stable')
    col = src[order].astype(np.int32)
    rowptr = np.zeros(n_dst + 1, dtype=np.int64)
    np.cumsum(np.bincount(dst, minlength=n_dst), out=rowptr[1:])
    return rowptr, col


class GraphSchema:
    """Static type-level description shared by sampler, model and kernels: node-type order,
    relation order, and for every relation its slot among the relations that share its
    destination type (column block of the per-destination Z buffer) and its source type."""

    def __init__(self, node_types: Iterable[str], edge_types: Iterable[EdgeType]):
        self.node_types: List[str] = list(node_types)
        self.edge_types: List[EdgeType] = [tuple(e) for e in edge_types]
        self.type_id = {t: i for i, t in enumerate(self.node_types)}
        self.NT = len(self.node_types)
        self.NR = len(self.edge_types)
        self.src_type = np.array([self.type_id[s] for s, _, _ in self.edge_types], dtype=np.int32)
        self.dst_type = np.array([self.type_id[d] for _, _, d in self.edge_types], dtype=np.int32)
        self.R_dst = np.zeros(self.NT, dtype=np.int32)   # relations per destination type
        self.R_src = np.zeros(self.NT, dtype=np.int32)   # relations per source type
        self.slot_dst = np.zeros(self.NR, dtype=np.int32)
        self.slot_src = np.zeros(self.NR, dtype=np.int32)
        for r in range(self.NR):
            self.slot_dst[r] = self.R_dst[self.dst_type[r]]
            self.R_dst[self.dst_type[r]] += 1
            self.slot_src[r] = self.R_src[self.src_type[r]]
            self.R_src[self.src_type[r]] += 1
        self.rels_by_dst = [[r for r in range(self.NR) if self.dst_type[r] == t] for t in range(self.NT)]
        self.rels_by_src = [[r for r in range(self.NR) if self.src_type[r] == t] for t in range(self.NT)]

    def live_relations(self, num_layers: int, out_type: str = 'SNP'):
        """Relations / node types structurally connected to the read-out: layer l
        relation r is live iff its dst type is consumed at layer l+1 (or is the read-out type at the
        last layer).  Everything else receives ``grad=None`` in the reference and is skipped by Adam."""
        live_types = [None] * (num_layers + 1)     # node types whose layer-l output is consumed
        live_rel = [None] * (num_layers + 1)
        live_types[num_layers] = {self.type_id[out_type]}
        for l in range(num_layers, 0, -1):
            rels = [r for r in range(self.NR) if int(self.dst_type[r]) in live_types[l]]
            live_rel[l] = rels
            need = set()
            for r in rels:
                need.add(int(self.src_type[r]))
                need.add(int(self.dst_type[r]))   # x_dst feeds lin_dst / (same-type) lin_src
            live_types[l - 1] = need
        return live_rel, live_types
