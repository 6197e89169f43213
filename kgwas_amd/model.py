"""HeteroGNN for KGWAS on MI355X -- same constructor, forward signature, outputs and ``state_dict`` keys
as the reference model (kgwas/model.py:24-86, kgwas/conv.py:36-232), executed by the fused HIP
kernels in kgwas_amd/csrc.

Execution plan of one HeteroConv layer (all relations at once):
  1. u_r = W_src^T att_src, v_r = W_dst^T att_dst (or W_src^T att_dst for same-type relations,
     conv.py:138) -- the destination-side linear map is only ever used through a_d = <x_d, v_r>
     (conv.py:144,151), so it is never materialised;
  2. a_d = H_d @ V_d^T for the destination rows the layer needs (pruned to hops <= L-l);
  3. kgw_gat_aggregate: Z[i, r] = sum_j softmax_j(leaky_relu(<H_s[j],u_r> + a_d[i,r])) H_s[j];
  4. out_d = relu( [Z[:,r0] | Z[:,r1] | ...] @ [W_r0^T ; W_r1^T ; ...] + sum_r bias_r )  -- the
     per-relation linear maps of conv.py:138/142, the bias of :190, the relation sum of PyG HeteroConv
     (model.py:74) and the ReLU of model.py:75 fused into one GEMM epilogue.
Only what the seeds' prediction depends on is computed (layer l on rows of hop <= L-l); rows the
reference computes and then discards (model.py:86 keeps [:batch_size]) carry zero gradient, so
parameter gradients are identical.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph import GraphSchema, HeteroGraph
from .sampler import BatchDict, SampledBatch, sample_full_graph

EdgeType = Tuple[str, str, str]
GO_TYPES = ('CellularComponent', 'BiologicalProcess', 'MolecularFunction')


def edge_key(et: EdgeType) -> str:
    """ModuleDict key of PyG <= 2.3 HeteroConv."""
    return '__'.join(et)


def _glorot_(t: torch.Tensor):
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)
    return t


class SimpleMLP(nn.Module):
    """kgwas/model.py:10-22."""

    def __init__(self, input_dim, hidden_dim, output_dim):
        super().__init__()
        self.FC_hidden = nn.Linear(input_dim, hidden_dim)
        self.FC_hidden2 = nn.Linear(hidden_dim, hidden_dim)
        self.FC_output = nn.Linear(hidden_dim, output_dim)
        self.ReLU = nn.ReLU()

    def forward(self, x):
        h = self.ReLU(self.FC_hidden(x))
        h = self.ReLU(self.FC_hidden2(h))
        return self.FC_output(h)


class GATConv(nn.Module):
    """Parameter holder of one relation's attention conv (kgwas/conv.py:81-120): bias-free
    ``lin_src`` / ``lin_dst`` (glorot), ``att_src`` / ``att_dst`` [1,H,C] (glorot), ``bias`` (zeros).
    Same-type relations never materialise ``lin_dst`` in the reference (it stays a lazy parameter)."""

    def __init__(self, in_channels: int, out_channels: int, bipartite: bool, heads: int = 1,
                 negative_slope: float = 0.2, temperature: float = 1.0):
        super().__init__()
        if heads != 1:
            raise NotImplementedError('gat_num_head > 1 breaks the reference read-out (model.py:50 expects '
                                      'hidden_channels inputs); only heads=1 is supported')
        self.heads, self.out_channels = heads, out_channels
        self.negative_slope, self.temperature = negative_slope, temperature
        self.lin_src = nn.Linear(in_channels, heads * out_channels, bias=False)
        _glorot_(self.lin_src.weight)
        if bipartite:
            self.lin_dst = nn.Linear(in_channels, heads * out_channels, bias=False)
            _glorot_(self.lin_dst.weight)
        else:
            self.lin_dst = None
        self.att_src = nn.Parameter(_glorot_(torch.empty(1, heads, out_channels)))
        self.att_dst = nn.Parameter(_glorot_(torch.empty(1, heads, out_channels)))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels))


class HeteroConv(nn.Module):
    """Container matching PyG HeteroConv's module tree: ``convs.<src__rel__dst>``."""

    def __init__(self, convs: "OrderedDict[EdgeType, GATConv]", aggr: str = 'sum'):
        super().__init__()
        self.edge_types = list(convs.keys())
        self.convs = nn.ModuleDict({edge_key(k): v for k, v in convs.items()})
        self.aggr = aggr

    def conv(self, et: EdgeType) -> GATConv:
        return self.convs[edge_key(et)]


class HeteroGNN(nn.Module):
    """kgwas/model.py:24-86 (same ctor / forward signature).  ``pyg_data`` only needs ``.edge_types``
    and ``.node_types``."""

    def __init__(self, pyg_data, hidden_channels, out_channels, num_layers, gnn_backbone, gnn_aggr,
                 snp_init_dim_size, gene_init_dim_size, go_init_dim_size, gat_num_head, no_relu=False):
        super().__init__()
        if gnn_backbone != 'GAT':
            raise NotImplementedError(f"backbone {gnn_backbone!r}: only 'GAT' (the reference default, "
                                      "kgwas.py:52) runs on the fused MI355X path")
        if gnn_aggr != 'sum':
            raise NotImplementedError("gnn_aggr: only 'sum' (the reference default) is fused")
        if hidden_channels != 128:
            raise NotImplementedError('the fused kernels are specialised for gnn_hidden_dim=128')
        self.node_types = list(pyg_data.node_types)
        self.edge_types = [tuple(e) for e in pyg_data.edge_types]
        self.schema = GraphSchema(self.node_types, self.edge_types)
        self.num_layers = num_layers
        self.hidden = hidden_channels
        self.convs = nn.ModuleList()
        self.snp_feat_mlp = SimpleMLP(snp_init_dim_size, hidden_channels, hidden_channels)
        self.go_feat_mlp = SimpleMLP(go_init_dim_size, hidden_channels, hidden_channels)
        self.gene_feat_mlp = SimpleMLP(gene_init_dim_size, hidden_channels, hidden_channels)
        self.ReLU = nn.ReLU()
        for _ in range(num_layers):
            layer = OrderedDict()
            for et in self.edge_types:
                layer[et] = GATConv(hidden_channels, hidden_channels, bipartite=(et[0] != et[2]),
                                    heads=gat_num_head)
            self.convs.append(HeteroConv(layer, aggr=gnn_aggr))
        self.lin = nn.Linear(hidden_channels, out_channels)
        self.no_relu = no_relu
        self.live_rel, self.live_types = self.schema.live_relations(num_layers, 'SNP')
        self.last_attention = None

    # ------------------------------------------------------------------------------------------
    def _mlp_for(self, t: str) -> SimpleMLP:
        if t == 'SNP':
            return self.snp_feat_mlp
        if t == 'Gene':
            return self.gene_feat_mlp
        if t in GO_TYPES:
            return self.go_feat_mlp
        raise KeyError(f'no feature MLP for node type {t!r} (kgwas/model.py:56-60)')

    def _packed(self, l: int):
        """Stack the live relations' parameters of layer l (1-based) for the fused kernels."""
        sc = self.schema
        hc: HeteroConv = self.convs[l - 1]
        live = set(self.live_rel[l])
        dev = self.lin.weight.device
        zero128 = torch.zeros(self.hidden, device=dev)
        U, V = [], []
        for r, et in enumerate(sc.edge_types):
            if r not in live:
                U.append(zero128); V.append(zero128)
                continue
            c = hc.conv(et)
            w_src = c.lin_src.weight
            w_dst = c.lin_dst.weight if c.lin_dst is not None else w_src
            U.append(c.att_src.view(-1) @ w_src)            # u_r = W_src^T att_src
            V.append(c.att_dst.view(-1) @ w_dst)            # v_r = W_dst^T att_dst
        return torch.stack(U), torch.stack(V)

    def _fused_layers(self, batch: SampledBatch, h: Dict[str, torch.Tensor], want_attention=False):
        sc = self.schema
        m = batch.meta
        NT = sc.NT
        attn = []
        for l in range(1, self.num_layers + 1):
            hc: HeteroConv = self.convs[l - 1]
            U, V = self._packed(l)
            # layer input, type-major (src_base)
            parts = []
            for t, name in enumerate(sc.node_types):
                ns = int(m.n_src[l - 1][t])
                if ns:
                    if name not in h or h[name].shape[0] < ns:
                        raise RuntimeError(f'layer {l}: node type {name!r} is a message source but has no '
                                           f'incoming relation to produce its layer-{l - 1} state')
                    parts.append(h[name][:ns])
            H = torch.cat(parts, 0) if parts else torch.zeros(0, self.hidden, device=U.device)
            # destination-side attention terms a_d[i, r]
            a_parts = []
            for t, name in enumerate(sc.node_types):
                nr = int(m.n_rows[l - 1][t])
                if nr:
                    a_parts.append((h[name][:nr] @ V[sc.rels_by_dst[t]].t()).reshape(-1))
            a_dst = torch.cat(a_parts) if a_parts else torch.zeros(0, device=U.device)
            c0 = hc.conv(sc.edge_types[0])
            Z, stat, e_edge = ops.gat_aggregate(batch, l, H, a_dst, U, c0.negative_slope, c0.temperature)
            if want_attention:
                attn.append(ops.edge_alpha(batch, l, stat, e_edge, c0.temperature))
            # per-relation linear maps + bias + relation sum + ReLU
            h_next = {}
            for t, name in enumerate(sc.node_types):
                nr = int(m.n_rows[l - 1][t])
                if not nr:
                    continue
                rels = sc.rels_by_dst[t]
                zb = int(m.z_base[l - 1][t])
                Zt = Z[zb:zb + nr * len(rels)].view(nr, len(rels) * self.hidden)
                Wcat = torch.cat([hc.conv(sc.edge_types[r]).lin_src.weight.t() for r in rels], 0)
                bsum = torch.stack([hc.conv(sc.edge_types[r]).bias for r in rels]).sum(0)
                h_next[name] = torch.relu(torch.addmm(bsum, Zt, Wcat))
            h = h_next
        return h, attn

    def forward(self, x_dict, edge_index_dict, batch_size, genotype=None, return_h=False,
                return_attention_weights=False):
        batch: Optional[SampledBatch] = getattr(x_dict, 'kgw_batch', None) or getattr(edge_index_dict, 'kgw_batch', None)
        if batch is None:
            batch = self._block_from_coo(x_dict, edge_index_dict)
        # feature MLPs (model.py:56-60); GO types share one MLP
        h = {}
        for t in self.node_types:
            if t in x_dict and x_dict[t].shape[0] > 0:
                h[t] = self._mlp_for(t)(x_dict[t])
            elif t in x_dict:
                h[t] = torch.zeros(0, self.hidden, device=self.lin.weight.device)
        h, attn = self._fused_layers(batch, h, want_attention=return_attention_weights)
        snp = h['SNP']
        out = self.lin(snp)[:batch_size]
        if return_h:                                            # model.py:78-79
            return self.ReLU(out), snp[:batch_size]
        if return_attention_weights:                            # model.py:80-81 (mean attention per layer)
            self.last_attention = attn
            return self.ReLU(out), [a.mean() if a.numel() else a.sum() for a in attn]
        if self.no_relu:                                        # model.py:83-84
            return out
        return self.ReLU(out)                                   # model.py:86

    # ------------------------------------------------------------------------------------------
    def _block_from_coo(self, x_dict, edge_index_dict) -> SampledBatch:
        """Plain (x_dict, edge_index_dict) inputs: every row of every type is computed in every layer,
        exactly like the reference on a PyG batch / the full graph (kgwas/utils.py:446-461)."""
        dev = next(iter(x_dict.values())).device
        g = HeteroGraph()
        for t in self.node_types:
            if t in x_dict:
                g[t].num_nodes_ = int(x_dict[t].shape[0])
            else:
                g[t].num_nodes_ = 0
        for et in self.edge_types:
            ei = edge_index_dict.get(et)
            g[et].edge_index = ei if ei is not None else torch.zeros(2, 0, dtype=torch.long)
        return sample_full_graph(g, self.num_layers, dev)

    # --- checkpoint compatibility (kgwas/utils.py:203-222) ---------------------------------------
    def state_dict(self, *args, **kwargs):
        sd = super().state_dict(*args, **kwargs)
        prefix = kwargs.get('prefix', args[1] if len(args) > 1 else '')
        out = OrderedDict()
        for k, v in sd.items():
            out[k] = v
        # same-type relations: PyG keeps an uninitialised lazy ``lin_dst.weight`` in the state_dict
        for l in range(self.num_layers):
            for et in self.edge_types:
                if et[0] == et[2]:
                    out[f'{prefix}convs.{l}.convs.{edge_key(et)}.lin_dst.weight'] = \
                        torch.nn.parameter.UninitializedParameter()
        return out

    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = OrderedDict()
        for k, v in state_dict.items():
            if isinstance(v, torch.nn.parameter.UninitializedParameter):
                continue                      # never-materialised lazy lin_dst of same-type relations
            k = k.replace('<', '').replace('>', '').replace('___', '__')   # PyG >= 2.4 key style
            sd[k] = v
        return super().load_state_dict(sd, strict=strict, **kw)
