"""CPU restatement of the reference model (TEST INFRASTRUCTURE; see oracle/__init__.py).

Follows, op for op:
  * kgwas/conv.py:122-228   GATConv.forward / edge_update / message (+ PyG propagate = index_select
                            of x_j, alpha_j, alpha_i and scatter-add into dim_size rows)
  * kgwas/model.py:10-22    SimpleMLP
  * kgwas/model.py:24-86    HeteroGNN (+ PyG HeteroConv: per-edge-type conv, group by dst, aggr)
Parameter names equal the reference's ``state_dict`` keys so weights can be exchanged with the
product (``kgwas_amd.model.HeteroGNN``) by name.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .pyg_semantics import segment_softmax

EdgeType = Tuple[str, str, str]


def glorot_(t: torch.Tensor, gen=None):
    """torch_geometric.nn.inits.glorot (conv.py:117-119): U(-a, a), a = sqrt(6/(size(-2)+size(-1)))."""
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a, generator=gen)
    return t


class GATConvOracle(nn.Module):
    """kgwas/conv.py:36-232 restricted to what HeteroGNN instantiates (model.py:40-42):
    in_channels=(-1,-1) [two lazy bias-free Linear: lin_src, lin_dst], heads H, concat=True,
    add_self_loops=False, edge_dim=None, bias=True, sigmoid_gat=False, temperature=1, dropout=0.
    ``sigmoid_gat`` / ``temperature`` / raw-attention return are kept because they are reachable
    through the public ctor / forward kwargs (conv.py:49-50,124)."""

    def __init__(self, in_src: int, in_dst: Optional[int], out_channels: int, heads: int = 1,
                 negative_slope: float = 0.2, sigmoid_gat: bool = False, temperature: float = 1.0,
                 dtype=torch.float32, gen=None):
        super().__init__()
        H, C = heads, out_channels
        self.heads, self.out_channels = H, C
        self.negative_slope = negative_slope
        self.sigmoid_gat = sigmoid_gat
        self.temperature = temperature
        self.lin_src = nn.Linear(in_src, H * C, bias=False, dtype=dtype)       # conv.py:86
        glorot_(self.lin_src.weight, gen)
        if in_dst is not None:                                                  # conv.py:88
            self.lin_dst = nn.Linear(in_dst, H * C, bias=False, dtype=dtype)
            glorot_(self.lin_dst.weight, gen)
        else:   # same-type relation: lin_dst is never materialised in the reference (lazy)
            self.lin_dst = None
        self.att_src = nn.Parameter(glorot_(torch.empty(1, H, C, dtype=dtype), gen))  # conv.py:92
        self.att_dst = nn.Parameter(glorot_(torch.empty(1, H, C, dtype=dtype), gen))  # conv.py:93
        self.bias = nn.Parameter(torch.zeros(H * C, dtype=dtype))               # conv.py:104,120

    def forward(self, x, edge_index: torch.Tensor, return_attention_weights=None,
                return_raw_attention_weights=None):
        H, C = self.heads, self.out_channels
        if isinstance(x, torch.Tensor):                                         # conv.py:136-138
            x_src = x_dst = self.lin_src(x).view(-1, H, C)
        else:                                                                   # conv.py:139-144
            xs, xd = x
            x_src = self.lin_src(xs).view(-1, H, C)
            x_dst = self.lin_dst(xd).view(-1, H, C) if xd is not None else None
        alpha_src = (x_src * self.att_src).sum(dim=-1)                          # conv.py:150
        alpha_dst = None if x_dst is None else (x_dst * self.att_dst).sum(-1)   # conv.py:151
        n_dst = x_dst.size(0) if x_dst is not None else x_src.size(0)
        src, dst = edge_index[0], edge_index[1]
        # edge_updater -> edge_update (conv.py:177, 200-225)
        alpha = alpha_src.index_select(0, src)                                  # alpha_j
        if alpha_dst is not None:
            alpha = alpha + alpha_dst.index_select(0, dst)                      # + alpha_i, :205
        alpha = F.leaky_relu(alpha, self.negative_slope)                        # :217
        if self.sigmoid_gat:
            alpha = torch.sigmoid(alpha / self.temperature)                     # :220
        elif not return_raw_attention_weights:
            alpha = segment_softmax(alpha / self.temperature, dst, n_dst)       # :223
        # dropout p=0 (:224) -> identity
        # propagate (conv.py:182): message alpha.unsqueeze(-1) * x_j (:227-228), aggr='add'
        msg = alpha.unsqueeze(-1) * x_src.index_select(0, src)
        out = torch.zeros(n_dst, H, C, dtype=msg.dtype).index_add(0, dst, msg)
        out = out.view(-1, H * C) + self.bias                                   # :185,190
        if isinstance(return_attention_weights, bool):                          # :192-194
            return out, (edge_index, alpha)
        return out


class SimpleMLPOracle(nn.Module):
    """kgwas/model.py:10-22."""

    def __init__(self, input_dim, hidden_dim, output_dim, dtype=torch.float32):
        super().__init__()
        self.FC_hidden = nn.Linear(input_dim, hidden_dim, dtype=dtype)
        self.FC_hidden2 = nn.Linear(hidden_dim, hidden_dim, dtype=dtype)
        self.FC_output = nn.Linear(hidden_dim, output_dim, dtype=dtype)

    def forward(self, x):
        h = F.relu(self.FC_hidden(x))
        h = F.relu(self.FC_hidden2(h))
        return self.FC_output(h)


class SAGEConvOracle(nn.Module):
    """PyG SAGEConv((-1,-1), C) as built at model.py:38 (aggr='mean', root_weight=True,
    lin_l has bias, lin_r has none): out_i = lin_l(mean_{j->i} x_j) + lin_r(x_i)."""

    def __init__(self, in_src, in_dst, out_channels, dtype=torch.float32):
        super().__init__()
        self.lin_l = nn.Linear(in_src, out_channels, bias=True, dtype=dtype)
        self.lin_r = nn.Linear(in_dst, out_channels, bias=False, dtype=dtype)

    def forward(self, x, edge_index):
        xs, xd = (x, x) if isinstance(x, torch.Tensor) else x
        src, dst = edge_index[0], edge_index[1]
        n_dst = xd.size(0)
        s = torch.zeros(n_dst, xs.size(1), dtype=xs.dtype).index_add(0, dst, xs.index_select(0, src))
        cnt = torch.zeros(n_dst, dtype=xs.dtype).index_add(0, dst, torch.ones_like(dst, dtype=xs.dtype))
        mean = s / cnt.clamp(min=1).unsqueeze(-1)
        return self.lin_l(mean) + self.lin_r(xd)


def _group(xs, aggr):
    """PyG hetero_conv.group (patched variant quoted at kgwas/utils.py:53-71)."""
    if len(xs) == 0:
        return None
    if aggr is None:
        return torch.stack(xs, dim=1)
    if len(xs) == 1:
        return xs[0]
    if aggr == 'cat':
        return torch.cat(xs, dim=-1)
    out = torch.stack(xs, dim=0)
    out = getattr(torch, aggr)(out, dim=0)
    return out[0] if isinstance(out, tuple) else out


def edge_key(et: EdgeType) -> str:
    """ModuleDict key of PyG <= 2.3 HeteroConv: '__'.join(edge_type)."""
    return '__'.join(et)


class HeteroConvOracle(nn.Module):
    """PyG HeteroConv(conv_dict, aggr) as used at model.py:47,66,74."""

    def __init__(self, convs: "OrderedDict[EdgeType, nn.Module]", aggr='sum'):
        super().__init__()
        self.edge_types = list(convs.keys())
        self.convs = nn.ModuleDict({edge_key(k): v for k, v in convs.items()})
        self.aggr = aggr

    def forward(self, x_dict, edge_index_dict, return_attention_weights=False, raw=False):
        out_dict: Dict[str, list] = {}
        att: Dict[EdgeType, torch.Tensor] = {}
        for et, edge_index in edge_index_dict.items():
            key = edge_key(et)
            if key not in self.convs:
                continue
            s, _, d = et
            x = x_dict[s] if s == d else (x_dict[s], x_dict[d])
            conv = self.convs[key]
            if return_attention_weights:
                out, (_, a) = conv(x, edge_index, return_attention_weights=True,
                                   return_raw_attention_weights=raw)
                att[et] = a
            else:
                out = conv(x, edge_index)
            out_dict.setdefault(d, []).append(out)
        res = {k: _group(v, self.aggr) for k, v in out_dict.items()}
        return (res, att) if return_attention_weights else res


GO_TYPES = ('CellularComponent', 'BiologicalProcess', 'MolecularFunction')


class HeteroGNNOracle(nn.Module):
    """kgwas/model.py:24-86."""

    def __init__(self, edge_types, hidden_channels, out_channels, num_layers, gnn_backbone,
                 gnn_aggr, snp_init_dim_size, gene_init_dim_size, go_init_dim_size, gat_num_head,
                 no_relu=False, dtype=torch.float32, gen=None):
        super().__init__()
        self.edge_types = list(edge_types)
        self.convs = nn.ModuleList()
        self.snp_feat_mlp = SimpleMLPOracle(snp_init_dim_size, hidden_channels, hidden_channels, dtype)
        self.go_feat_mlp = SimpleMLPOracle(go_init_dim_size, hidden_channels, hidden_channels, dtype)
        self.gene_feat_mlp = SimpleMLPOracle(gene_init_dim_size, hidden_channels, hidden_channels, dtype)
        for _ in range(num_layers):
            layer = OrderedDict()
            for et in self.edge_types:
                s, _, d = et
                if gnn_backbone == 'GAT':
                    layer[et] = GATConvOracle(hidden_channels, None if s == d else hidden_channels,
                                              hidden_channels, heads=gat_num_head, dtype=dtype, gen=gen)
                elif gnn_backbone == 'SAGE':
                    layer[et] = SAGEConvOracle(hidden_channels, hidden_channels, hidden_channels, dtype)
                else:
                    raise NotImplementedError(gnn_backbone)  # GCN/SGC cannot run on bipartite inputs
            self.convs.append(HeteroConvOracle(layer, aggr=gnn_aggr))
        self.lin = nn.Linear(hidden_channels, out_channels, dtype=dtype)        # model.py:50
        self.no_relu = no_relu

    def forward(self, x_dict, edge_index_dict, batch_size, genotype=None, return_h=False,
                return_attention_weights=False):
        x_dict = dict(x_dict)
        x_dict['SNP'] = self.snp_feat_mlp(x_dict['SNP'])                        # model.py:56
        x_dict['Gene'] = self.gene_feat_mlp(x_dict['Gene'])                     # :57
        for t in GO_TYPES:                                                      # :58-60
            if t in x_dict:
                x_dict[t] = self.go_feat_mlp(x_dict[t])
        attention_all_layers = []
        for conv in self.convs:                                                 # :64
            if return_attention_weights:                                        # :65-72
                x_dict, att = conv(x_dict, edge_index_dict, return_attention_weights=True)
                attention_all_layers.append(torch.mean(torch.vstack([a for a in att.values()])))
            else:
                x_dict = conv(x_dict, edge_index_dict)                          # :74
            x_dict = {k: v.relu() for k, v in x_dict.items()}                   # :75
        if return_h:                                                            # :78-79
            return F.relu(self.lin(x_dict['SNP']))[:batch_size], x_dict['SNP'][:batch_size]
        if return_attention_weights:                                            # :80-81
            return F.relu(self.lin(x_dict['SNP']))[:batch_size], attention_all_layers
        if self.no_relu:                                                        # :83-84
            return self.lin(x_dict['SNP'])[:batch_size]
        return F.relu(self.lin(x_dict['SNP']))[:batch_size]                     # :86


def weighted_mse(pred: torch.Tensor, y: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """kgwas/kgwas.py:139-145: pred fp32, w float64 -> loss promoted to float64."""
    return torch.mean(w * (pred.reshape(-1) - y) ** 2)
