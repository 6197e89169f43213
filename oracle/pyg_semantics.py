"""Restatement of the torch_geometric behaviours the reference relies on (TEST INFRASTRUCTURE).

Third-party dependency: torch_geometric (unpinned, requirements.txt:6; era 2.1-2.3 judging by
kgwas/conv.py:7,177).  Not vendored in /root/reference, not installable here -> restated from its
published semantics; every function names the reference call site that needs it.

All functions are plain torch on CPU tensors (int64 indices like the reference).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

import torch

EdgeType = Tuple[str, str, str]


# ----------------------------------------------------------------------------------------------
# Graph transforms -- kgwas/kgwas_data.py:271-272  (T.ToUndirected(), T.AddSelfLoops())
# ----------------------------------------------------------------------------------------------
def coalesce(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """PyG ``coalesce``: sort by (row, col) and drop duplicate edges."""
    row, col = edge_index[0], edge_index[1]
    key = row * num_nodes + col
    key = torch.unique(key, sorted=True)
    return torch.stack([key // num_nodes, key % num_nodes], dim=0)


def to_undirected(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """PyG ``to_undirected`` (same-type relation): cat with the flipped list, then coalesce."""
    row, col = edge_index[0], edge_index[1]
    both = torch.stack([torch.cat([row, col]), torch.cat([col, row])], dim=0)
    return coalesce(both, num_nodes)


def to_undirected_hetero(edge_index_dict: "OrderedDict[EdgeType, torch.Tensor]",
                         num_nodes: Dict[str, int]) -> "OrderedDict[EdgeType, torch.Tensor]":
    """``T.ToUndirected()`` on HeteroData (kgwas_data.py:271).

    Bipartite edge types (src type != dst type) keep their list untouched and gain a mirror type
    ``(dst, 'rev_'+rel, src)`` appended after all existing types; same-type relations are
    symmetrised and coalesced in place.
    """
    out: "OrderedDict[EdgeType, torch.Tensor]" = OrderedDict()
    rev: "OrderedDict[EdgeType, torch.Tensor]" = OrderedDict()
    for (src, rel, dst), ei in edge_index_dict.items():
        ei = torch.as_tensor(ei, dtype=torch.long)
        if src != dst:
            out[(src, rel, dst)] = ei
            rev[(dst, 'rev_' + rel, src)] = torch.stack([ei[1], ei[0]], dim=0)
        else:
            out[(src, rel, dst)] = to_undirected(ei, num_nodes[src])
    for k, v in rev.items():
        out[k] = v
    return out


def add_self_loops_hetero(edge_index_dict: "OrderedDict[EdgeType, torch.Tensor]",
                          num_nodes: Dict[str, int]) -> "OrderedDict[EdgeType, torch.Tensor]":
    """``T.AddSelfLoops()`` (kgwas_data.py:272): same-type relations get N appended (i,i) loops,
    existing loops are NOT removed; bipartite relations are untouched."""
    out: "OrderedDict[EdgeType, torch.Tensor]" = OrderedDict()
    for (src, rel, dst), ei in edge_index_dict.items():
        if src == dst:
            loop = torch.arange(num_nodes[src], dtype=torch.long)
            ei = torch.cat([ei, torch.stack([loop, loop], dim=0)], dim=1)
        out[(src, rel, dst)] = ei
    return out


# ----------------------------------------------------------------------------------------------
# Segment softmax -- kgwas/conv.py:223 (torch_geometric.utils.softmax)
# ----------------------------------------------------------------------------------------------
def segment_softmax(src: torch.Tensor, index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """PyG softmax grouped by ``index`` along dim 0: subtract the (detached) per-group max, exp,
    divide by (group sum + 1e-16)."""
    shape = (num_nodes,) + tuple(src.shape[1:])
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    src_max = torch.full(shape, float('-inf'), dtype=src.dtype).scatter_reduce(
        0, idx, src.detach(), reduce='amax', include_self=True)
    out = (src - src_max.index_select(0, index)).exp()
    out_sum = torch.zeros(shape, dtype=src.dtype).index_add(0, index, out) + 1e-16
    return out / out_sum.index_select(0, index)


# ----------------------------------------------------------------------------------------------
# NeighborLoader(num_neighbors=[-1]*L) -- kgwas/kgwas.py:99-113
# ----------------------------------------------------------------------------------------------
def to_csc(edge_index: torch.Tensor, n_src: int, n_dst: int):
    """PyG ``to_csc``: edges sorted by (dst, src); returns colptr[n_dst+1], row[E], perm."""
    row, col = edge_index[0], edge_index[1]
    perm = torch.argsort(col * n_src + row, stable=True)
    row_s = row[perm]
    counts = torch.bincount(col, minlength=n_dst)
    colptr = torch.zeros(n_dst + 1, dtype=torch.long)
    colptr[1:] = torch.cumsum(counts, 0)
    return colptr, row_s, perm


class FullNeighborSampler:
    """Deterministic full-neighbourhood expansion.

    hop 0 = the seed nodes of ``input_type``; hop k: for every edge type whose dst type gained
    nodes in hop k-1, append ALL in-neighbours (directed, no replacement) to the src type's node
    list in first-seen order (seeds first) and emit the local COO edge (src_local, dst_local).
    Nodes added during hop k are not expanded before hop k+1.
    """

    def __init__(self, edge_index_dict, num_nodes: Dict[str, int], num_layers: int):
        self.edge_types: List[EdgeType] = list(edge_index_dict.keys())
        self.num_nodes = dict(num_nodes)
        self.num_layers = num_layers
        self.csc = {}
        for et, ei in edge_index_dict.items():
            s, _, d = et
            colptr, row, _ = to_csc(torch.as_tensor(ei, dtype=torch.long),
                                    self.num_nodes[s], self.num_nodes[d])
            self.csc[et] = (colptr.numpy(), row.numpy())

    def sample(self, input_type: str, seeds):
        import numpy as np
        seeds = np.asarray(seeds, dtype=np.int64)
        nodes: Dict[str, List[int]] = {t: [] for t in self.num_nodes}
        local: Dict[str, Dict[int, int]] = {t: {} for t in self.num_nodes}
        hop_of: Dict[str, List[int]] = {t: [] for t in self.num_nodes}
        for g in seeds.tolist():
            local[input_type][g] = len(nodes[input_type])
            nodes[input_type].append(g)
            hop_of[input_type].append(0)
        rows = {et: [] for et in self.edge_types}
        cols = {et: [] for et in self.edge_types}
        begin = {t: 0 for t in self.num_nodes}
        for hop in range(1, self.num_layers + 1):
            end = {t: len(nodes[t]) for t in self.num_nodes}
            for et in self.edge_types:
                s, _, d = et
                colptr, row = self.csc[et]
                for dl in range(begin[d], end[d]):
                    g = nodes[d][dl]
                    for e in range(colptr[g], colptr[g + 1]):
                        sg = int(row[e])
                        sl = local[s].get(sg)
                        if sl is None:
                            sl = len(nodes[s])
                            local[s][sg] = sl
                            nodes[s].append(sg)
                            hop_of[s].append(hop)
                        rows[et].append(sl)
                        cols[et].append(dl)
            begin = end
        n_id = {t: torch.tensor(v, dtype=torch.long) for t, v in nodes.items()}
        hops = {t: torch.tensor(v, dtype=torch.long) for t, v in hop_of.items()}
        edge_index = OrderedDict()
        for et in self.edge_types:
            edge_index[et] = torch.tensor([rows[et], cols[et]], dtype=torch.long).reshape(2, -1)
        return n_id, edge_index, hops
