"""Vectorised numpy version of oracle.pyg_semantics.FullNeighborSampler (TEST INFRASTRUCTURE /
cpu_baseline leg of bench.py).  Same semantics -- PyG NeighborLoader(num_neighbors=[-1]*L) as used at
kgwas/kgwas.py:99-113: per-relation CSC, hop-wise expansion over all in-neighbours, first-seen local
order with the seeds first -- but array-at-a-time so it can drive the CPU baseline on the full-size
graph.  Checked against the loop version in tests/test_oracle.py."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import numpy as np
import torch


class FullNeighborSamplerNP:
    def __init__(self, edge_index_dict, num_nodes: Dict[str, int], num_layers: int):
        self.edge_types = list(edge_index_dict.keys())
        self.num_nodes = dict(num_nodes)
        self.num_layers = num_layers
        self.csc = {}
        for et, ei in edge_index_dict.items():
            s, _, d = et
            ei = ei.numpy() if torch.is_tensor(ei) else np.asarray(ei)
            row, col = ei[0].astype(np.int64), ei[1].astype(np.int64)
            perm = np.argsort(col * self.num_nodes[s] + row, kind='stable')     # PyG to_csc order
            colptr = np.zeros(self.num_nodes[d] + 1, dtype=np.int64)
            np.cumsum(np.bincount(col, minlength=self.num_nodes[d]), out=colptr[1:])
            self.csc[et] = (colptr, row[perm])
        self._local = {t: np.full(n, -1, dtype=np.int64) for t, n in self.num_nodes.items()}

    def sample(self, input_type: str, seeds):
        seeds = np.asarray(seeds, dtype=np.int64)
        local = self._local
        nodes = {t: [np.zeros(0, np.int64)] for t in self.num_nodes}
        count = {t: 0 for t in self.num_nodes}
        nodes[input_type] = [seeds]
        local[input_type][seeds] = np.arange(len(seeds))
        count[input_type] = len(seeds)
        rows = {et: [] for et in self.edge_types}
        cols = {et: [] for et in self.edge_types}
        begin = {t: 0 for t in self.num_nodes}
        for _hop in range(self.num_layers):
            end = dict(count)
            flat = {t: np.concatenate(nodes[t]) for t in self.num_nodes}
            for et in self.edge_types:
                s, _, d = et
                f = flat[d][begin[d]:end[d]]
                if f.size == 0:
                    continue
                colptr, row = self.csc[et]
                starts = colptr[f]
                cnt = colptr[f + 1] - starts
                tot = int(cnt.sum())
                if tot == 0:
                    continue
                excl = np.cumsum(cnt) - cnt
                pos = np.arange(tot) - np.repeat(excl, cnt) + np.repeat(starts, cnt)
                src_g = row[pos]
                dst_l = np.repeat(np.arange(begin[d], end[d]), cnt)
                unseen = src_g[local[s][src_g] < 0]
                if unseen.size:
                    u, first = np.unique(unseen, return_index=True)
                    new = u[np.argsort(first, kind='stable')]            # first-seen order
                    local[s][new] = count[s] + np.arange(len(new))
                    nodes[s].append(new)
                    count[s] += len(new)
                rows[et].append(local[s][src_g])
                cols[et].append(dst_l)
            begin = end
        n_id = {t: torch.from_numpy(np.concatenate(nodes[t])) for t in self.num_nodes}
        for t in self.num_nodes:                                          # reset the dense map
            local[t][n_id[t].numpy()] = -1
        edge_index = OrderedDict()
        for et in self.edge_types:
            if rows[et]:
                edge_index[et] = torch.from_numpy(np.stack([np.concatenate(rows[et]), np.concatenate(cols[et])]))
            else:
                edge_index[et] = torch.zeros(2, 0, dtype=torch.long)
        return n_id, edge_index
