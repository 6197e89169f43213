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
     (model.py:74) and the ReLU of model.py:75 as ONE GEMM per destination type.
Only what the seeds' prediction depends on is computed (layer l on rows of hop <= L-l); rows the
reference computes and then discards (model.py:86 keeps [:batch_size]) carry zero gradient, so
parameter gradients are identical.

Parameter storage is MI355X-first: the ~30 relations x 5 tensors of a layer live in a handful of packed
tensors (``RelationPack``), laid out so that step 4's concatenated weight is a free view and steps 1-2
are three batched mat-vecs -- a step issues ~100 launches instead of ~1000.  ``state_dict()`` /
``load_state_dict()`` translate to and from the reference's per-relation keys
(``convs.<l>.convs.<src__rel__dst>.{lin_src.weight,lin_dst.weight,att_src,att_dst,bias}``), so a
checkpoint written by either implementation loads in the other (kgwas/utils.py:203-222).
Relations that are structurally disconnected from the read-out (e.g. every layer-2 relation whose
destination is not SNP) are kept in a separate pack that never enters the autograd graph: like in the
reference their gradient is None and Adam (incl. its weight decay) never touches them.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .graph import GraphSchema, HeteroGraph
from .sampler import SampledBatch, gather_rows_multi, sample_full_graph

_RELVEC_ALL = os.environ.get('KGW_RELVEC_ALL', '1') != '0'      # (A/B: one relation-vector launch per layer as before)

EdgeType = Tuple[str, str, str]
GO_TYPES = ('CellularComponent', 'BiologicalProcess', 'MolecularFunction')
REL_FIELDS = ('att_src', 'att_dst', 'bias', 'lin_src.weight', 'lin_dst.weight')


def edge_key(et: EdgeType) -> str:
    """ModuleDict key of PyG <= 2.3 HeteroConv."""
    return '__'.join(et)


def _uniform_(t: torch.Tensor, a: float):
    with torch.no_grad():
        t.uniform_(-a, a)
    return t


def _padded_linear(in_l: int, out_l: int, in_p: int, out_p: int) -> nn.Linear:
    """nn.Linear(in_l, out_l) with its default initialisation, stored zero-padded as [out_p, in_p] (see HeteroGNN: a hidden width
    below 128 runs on the 128-wide kernels; the padding stays exactly zero through training)."""
    if (in_l, out_l) == (in_p, out_p):
        return nn.Linear(in_p, out_p)
    src = nn.Linear(in_l, out_l)
    lin = nn.Linear(in_p, out_p)
    with torch.no_grad():
        lin.weight.zero_(); lin.bias.zero_()
        lin.weight[:out_l, :in_l].copy_(src.weight)
        lin.bias[:out_l].copy_(src.bias)
    return lin


class SimpleMLP(nn.Module):
    """kgwas/model.py:10-22.  ``c_logical``: the model's hidden width when it is below the kernels' 128 (zero-padded storage)."""

    def __init__(self, input_dim, hidden_dim, output_dim, c_logical=None):
        super().__init__()
        cl = hidden_dim if c_logical is None else c_logical
        self.FC_hidden = _padded_linear(input_dim, cl, input_dim, hidden_dim)
        self.FC_hidden2 = _padded_linear(cl, cl, hidden_dim, hidden_dim)
        self.FC_output = _padded_linear(cl, cl, hidden_dim, output_dim)
        self.ReLU = nn.ReLU()

    def first(self, x, fixed_shape=False):
        """relu(FC_hidden(x)) -- one autograd node; weight gradient on the split-K MFMA kernel."""
        return ops.linear_relu(x, self.FC_hidden.weight, self.FC_hidden.bias, fixed_shape)

    def tail(self, h1, out=None):
        """FC_output(relu(FC_hidden2(h1))) -- one autograd node."""
        return ops.mlp_tail(h1, self.FC_hidden2.weight, self.FC_hidden2.bias, self.FC_output.weight, self.FC_output.bias,
                            out)

    def tail2(self, h1, out=None):
        """relu(FC_hidden2(h1)): the MLP without FC_output (folded into layer 1, ops.fold_fc_output_hip)."""
        return ops.mlp_tail2(h1, self.FC_hidden2.weight, self.FC_hidden2.bias, out)

    def hidden(self, x, out=None, rows_dev=None, ids=None):
        """h2 = relu(FC_hidden2(relu(FC_hidden(x)))): the MLP without FC_output (folded into layer 1).  ``ids`` (int32): the
        input rows are ``x[ids]`` of a resident feature matrix."""
        if x.requires_grad:
            if ids is not None:
                x = x[ids.long()]
            return self.tail2(self.first(x), out)
        return ops.mlp2(x, self.FC_hidden.weight, self.FC_hidden.bias, self.FC_hidden2.weight, self.FC_hidden2.bias, out, rows_dev,
                        ids)

    def forward(self, x, out=None, rows_dev=None):
        if x.requires_grad:
            return self.tail(self.first(x), out)
        return ops.mlp3(x, self.FC_hidden.weight, self.FC_hidden.bias, self.FC_hidden2.weight, self.FC_hidden2.bias,
                        self.FC_output.weight, self.FC_output.bias, out, rows_dev)




class RelationPack(nn.Module):
    """Parameters of a list of relations of one layer, packed (kgwas/conv.py:81-120 per relation):
    ``w_src_t[i]`` = lin_src.weight^T  ([in k, out c], glorot), ``w_dst_t[j]`` = lin_dst.weight^T of the
    j-th bipartite relation (same-type relations never materialise lin_dst in the reference),
    ``att_src`` / ``att_dst`` (glorot on [1,1,C]), ``bias`` (zeros)."""

    def __init__(self, edge_types: List[EdgeType], rel_ids: List[int], C: int, c_logical: int = None):
        super().__init__()
        self.n_rels_total = len(edge_types)
        self.rel_ids = list(rel_ids)
        self.rel_types = [edge_types[r] for r in rel_ids]
        n = len(rel_ids)
        bip = [i for i, et in enumerate(self.rel_types) if et[0] != et[2]]
        self.bip = bip
        self.bip_pos = {i: j for j, i in enumerate(bip)}
        cl = self.cl = C if c_logical is None else c_logical      # the model's width; storage is C (= 128) wide, zero beyond cl
        a_w = math.sqrt(6.0 / (cl + cl))        # glorot on [C_out, C_in]
        a_a = math.sqrt(6.0 / (1 + cl))         # glorot on [1, heads=1, C]

        def block(shape, a):
            if cl == C:
                return _uniform_(torch.empty(*shape), a)
            t = torch.zeros(*shape)
            with torch.no_grad():
                if len(shape) == 3:
                    t[:, :cl, :cl].uniform_(-a, a)
                else:
                    t[:, :cl].uniform_(-a, a)
            return t
        self.w_src_t = nn.Parameter(block((n, C, C), a_w))
        self.w_dst_t = nn.Parameter(block((len(bip), C, C), a_w))
        self.att_src = nn.Parameter(block((n, C), a_a))
        self.att_dst = nn.Parameter(block((n, C), a_a))
        self.bias = nn.Parameter(torch.zeros(n, C))
        self.register_buffer('rel_ids_t', torch.tensor(rel_ids, dtype=torch.long), persistent=False)
        self.register_buffer('bip_t', torch.tensor(bip, dtype=torch.long), persistent=False)
        # int32 tables of the HIP kernels: relation id of each packed slot, packed slot of each relation id (-1 =
        # not in this pack), index into w_dst_t of each packed slot (-1 = same-type relation)
        live_of = [-1] * len(edge_types)
        for i, r in enumerate(rel_ids):
            live_of[r] = i
        self._sel_cache = {}                       # relation-sum selectors of ops.layer_transform, by block layout
        self.register_buffer('rel_ids_i32', torch.tensor(rel_ids, dtype=torch.int32), persistent=False)
        self.register_buffer('live_of_rel_i32', torch.tensor(live_of, dtype=torch.int32), persistent=False)
        self.register_buffer('bip_pos_i32', torch.tensor([self.bip_pos.get(i, -1) for i in range(n)], dtype=torch.int32),
                             persistent=False)

    # reference-named view of one tensor of relation slot i (value or gradient)
    def get(self, i: int, field: str, grad: bool = False):
        def pick(p):
            if grad:
                return None if p.grad is None else p.grad
            return p.detach()
        cl = self.cl
        if field == 'lin_src.weight':
            t = pick(self.w_src_t)
            return None if t is None else t[i, :cl, :cl].t()
        if field == 'lin_dst.weight':
            if i not in self.bip_pos:
                return 'lazy'
            t = pick(self.w_dst_t)
            return None if t is None else t[self.bip_pos[i], :cl, :cl].t()
        t = pick(getattr(self, field))
        if t is None:
            return None
        return t[i, :cl].reshape(1, 1, -1) if field.startswith('att') else t[i, :cl]

    def set(self, i: int, field: str, value: torch.Tensor):
        cl = self.cl
        with torch.no_grad():
            if field == 'lin_src.weight':
                self.w_src_t[i, :cl, :cl].copy_(value.t())
            elif field == 'lin_dst.weight':
                if i in self.bip_pos:
                    self.w_dst_t[self.bip_pos[i], :cl, :cl].copy_(value.t())
            elif field.startswith('att'):
                getattr(self, field)[i, :cl].copy_(value.reshape(-1))
            else:
                getattr(self, field)[i, :cl].copy_(value)


class SagePack(nn.Module):
    """Parameters of a list of SAGEConv((-1,-1), C) relations of one layer, packed (kgwas/model.py:38; PyG SAGEConv:
    aggr='mean', root_weight=True): ``w_l_t[i]`` = lin_l.weight^T ([in, out]) and ``bias[i]`` = lin_l.bias act on the
    mean of the neighbours, ``w_r_t[i]`` = lin_r.weight^T (no bias) on the destination node itself.  Same field
    plumbing as RelationPack (``get`` / ``set`` by reference name)."""

    FIELDS = ('lin_l.weight', 'lin_l.bias', 'lin_r.weight')

    def __init__(self, edge_types: List[EdgeType], rel_ids: List[int], C: int, c_logical: int = None):
        super().__init__()
        self.n_rels_total = len(edge_types)
        self.rel_ids = list(rel_ids)
        n = len(rel_ids)
        cl = self.cl = C if c_logical is None else c_logical
        a = 1.0 / math.sqrt(cl)                    # nn.Linear's default (kaiming_uniform a=sqrt(5)) bound for fan_in = cl

        def block(shape):
            if cl == C:
                return _uniform_(torch.empty(*shape), a)
            t = torch.zeros(*shape)
            with torch.no_grad():
                (t[:, :cl, :cl] if len(shape) == 3 else t[:, :cl]).uniform_(-a, a)
            return t
        self.w_l_t = nn.Parameter(block((n, C, C)))
        self.bias = nn.Parameter(block((n, C)))
        self.w_r_t = nn.Parameter(block((n, C, C)))
        self._sel_cache = {}

    def get(self, i: int, field: str, grad: bool = False):
        p = {'lin_l.weight': self.w_l_t, 'lin_l.bias': self.bias, 'lin_r.weight': self.w_r_t}[field]
        t = p.grad if grad else p.detach()
        if t is None:
            return None
        cl = self.cl
        return t[i, :cl] if field == 'lin_l.bias' else t[i, :cl, :cl].t()

    def set(self, i: int, field: str, value: torch.Tensor):
        cl = self.cl
        with torch.no_grad():
            if field == 'lin_l.bias':
                self.bias[i, :cl].copy_(value)
            else:
                (self.w_l_t if field == 'lin_l.weight' else self.w_r_t)[i, :cl, :cl].copy_(value.t())


class HeteroGNN(nn.Module):
    """kgwas/model.py:24-86 (same ctor / forward signature).  ``pyg_data`` only needs ``.edge_types``
    and ``.node_types``."""

    def __init__(self, pyg_data, hidden_channels, out_channels, num_layers, gnn_backbone, gnn_aggr,
                 snp_init_dim_size, gene_init_dim_size, go_init_dim_size, gat_num_head, no_relu=False):
        super().__init__()
        if gnn_backbone not in ('GAT', 'SAGE'):
            raise NotImplementedError(f"backbone {gnn_backbone!r}: 'GAT' (the reference default, kgwas.py:52) and 'SAGE' "
                                      "run on the fused MI355X path; GCNConv / SGConv cannot take the bipartite "
                                      "relations HeteroConv hands them (kgwas/model.py:44-46)")
        self.backbone = gnn_backbone
        if gnn_aggr not in ('sum', 'mean', 'min', 'max'):
            raise NotImplementedError(f"gnn_aggr {gnn_aggr!r}: 'sum' (the reference default, fused), 'mean', 'min' and 'max' are "
                                      "built; 'cat' widens the hidden state to R*128 and breaks the reference's own "
                                      "read-out (kgwas/model.py:50), like gat_num_head > 1")
        self.aggr = gnn_aggr
        if not 1 <= hidden_channels <= 128:
            raise NotImplementedError('gnn_hidden_dim > 128: the fused kernels are 128 wide (narrower models run on them '
                                      'zero-padded; wider ones would need a second instantiation of every kernel -- DESIGN.md section 8)')
        if gat_num_head != 1:
            raise NotImplementedError('gat_num_head > 1 breaks the reference read-out (model.py:50 expects '
                                      'hidden_channels inputs); only heads=1 is supported')
        self.node_types = list(pyg_data.node_types)
        self.edge_types = [tuple(e) for e in pyg_data.edge_types]
        self.schema = GraphSchema(self.node_types, self.edge_types)
        sc = self.schema
        self.num_layers = num_layers
        # gnn_hidden_dim < 128 (kgwas/kgwas.py:52): the model is embedded in the 128-wide kernels -- every hidden tensor and
        # parameter zero-padded to 128.  Exact: padded channels are 0 after every Linear / ReLU, padded parameters receive a
        # gradient of exactly 0 (their inputs or their output gradients are 0) and Adam with L2 leaves 0 at 0; checkpoints,
        # state_dict and gradients are exposed at the model's own width.  Costs what 128 costs.
        self.hidden_logical = cl = int(hidden_channels)
        hidden_channels = 128
        self.hidden = hidden_channels
        self.negative_slope, self.temperature = 0.2, 1.0          # conv.py:43,50 defaults (model.py:40-42)
        self.rel_fields = REL_FIELDS if gnn_backbone == 'GAT' else SagePack.FIELDS
        self.live_rel, self.live_types = sc.live_relations(num_layers, 'SNP')
        self.live_packs = nn.ModuleList()
        self.dead_packs = nn.ModuleList()
        self._slot: List[Dict[int, Tuple[str, int]]] = []        # per layer: relation id -> (pack, index)
        self._dst_range: List[Dict[int, Tuple[int, int]]] = []   # per layer: dst type -> [lo, hi) in live pack
        for l in range(1, num_layers + 1):
            live = set(self.live_rel[l])
            order = [r for t in range(sc.NT) for r in sc.rels_by_dst[t] if r in live]   # grouped by dst type
            dead = [r for r in range(sc.NR) if r not in live]
            Pack = RelationPack if gnn_backbone == 'GAT' else SagePack
            self.live_packs.append(Pack(self.edge_types, order, hidden_channels, cl))
            self.dead_packs.append(Pack(self.edge_types, dead, hidden_channels, cl))
            slot = {r: ('live', i) for i, r in enumerate(order)}
            slot.update({r: ('dead', i) for i, r in enumerate(dead)})
            self._slot.append(slot)
            rng, lo = {}, 0
            for t in range(sc.NT):
                k = sum(1 for r in sc.rels_by_dst[t] if r in live)
                if k:
                    assert k == len(sc.rels_by_dst[t])
                    rng[t] = (lo, lo + k)
                    lo += k
            self._dst_range.append(rng)
        self.snp_feat_mlp = SimpleMLP(snp_init_dim_size, hidden_channels, hidden_channels, cl)
        self.go_feat_mlp = SimpleMLP(go_init_dim_size, hidden_channels, hidden_channels, cl)
        self.gene_feat_mlp = SimpleMLP(gene_init_dim_size, hidden_channels, hidden_channels, cl)
        self.ReLU = nn.ReLU()
        self.lin = _padded_linear(cl, out_channels, hidden_channels, out_channels)
        self.no_relu = no_relu
        self.last_attention = None
        # FC_output of the feature MLPs folded into the layer-1 relation parameters (ops.fold_fc_output_hip): exact, removes a
        # 128 x 128 Linear (forward, dX, dW) over every sampled node.  GAT with a relation SUM only: SAGE has a root term and
        # min / max are not linear in the messages.
        import os
        self.fold_fc = gnn_backbone == 'GAT' and gnn_aggr in ('sum', 'mean') and os.environ.get('KGW_FOLD_FC', '1') == '1'
        if self.fold_fc:
            mlp_of = {'SNP': 0, 'Gene': 1}
            order = self.live_packs[0].rel_ids
            try:
                sm = [mlp_of.get(self.edge_types[r][0], 2) for r in order]
                dm = [mlp_of.get(self.edge_types[r][2], 2) for r in order]
                for r in order:
                    for t in (self.edge_types[r][0], self.edge_types[r][2]):
                        self._mlp_for(t)
            except KeyError:
                self.fold_fc = False
            if self.fold_fc:
                self._fold_used = set(sm) | set(dm)
                import numpy as np
                self._fold_tab = (np.asarray(order, dtype=np.int32), np.asarray(sm, dtype=np.int32), np.asarray(dm, dtype=np.int32))
                self.register_buffer('_fold_src_m', torch.tensor(sm, dtype=torch.long), persistent=False)
                self.register_buffer('_fold_dst_m', torch.tensor(dm, dtype=torch.long), persistent=False)
        # dead packs never receive gradients; keep them out of autograd entirely
        for p in self.dead_packs.parameters():
            p.requires_grad_(False)

    # ------------------------------------------------------------------------------------------
    def _mlp_for(self, t: str) -> SimpleMLP:
        if t == 'SNP':
            return self.snp_feat_mlp
        if t == 'Gene':
            return self.gene_feat_mlp
        if t in GO_TYPES:
            return self.go_feat_mlp
        raise KeyError(f'no feature MLP for node type {t!r} (kgwas/model.py:56-60)')

    def _embed(self, batch: SampledBatch, x_dict, t: str, out=None, fold=False):
        """Feature MLP of the sampled nodes of type t (model.py:56-60).  When most of a type is in the batch
        and its features are wide (the 5120 / 57742-wide gene matrix), the first Linear runs on the RESIDENT
        matrix and the 128-wide result is sliced, instead of slicing 20 KB rows first: same values, the
        x[n_id] copy (kgwas.py:135 moves it over PCIe every step) disappears."""
        mlp = self._mlp_for(t)
        dg = batch.dg
        n = batch.n_nodes[t]
        if n == 0:
            return torch.zeros(0, self.hidden, device=self.lin.weight.device)
        if out is not None and out.n != n:
            out = None
        lazy = getattr(x_dict, 'kgw_batch', None) is batch and t in dg.x
        if lazy and not dg.full_graph:
            X = dg.x[t]
            if X.shape[1] >= 512 and 2 * n > X.shape[0]:
                i = dg.schema.type_id[t]
                g2l = batch.buf.g2l[dg.node_base[i]:dg.node_base[i] + X.shape[0]]
                if fold and ops.resident_mlp2_ok(X, mlp.FC_hidden.weight, mlp.FC_hidden2.weight, n):
                    # (static layout: the block is padded to its capacity; the count of real rows lives on the device)
                    real = batch.rows_dev(t) if n == batch.lay_src(1, i) else None
                    return ops.resident_mlp2(X, mlp.FC_hidden.weight, mlp.FC_hidden.bias, mlp.FC_hidden2.weight,
                                             mlp.FC_hidden2.bias, batch.n_id(t), g2l, out, rows_real=real)
                h1 = ops.resident_linear_relu_rows(X, mlp.FC_hidden.weight, mlp.FC_hidden.bias, batch.n_id(t), g2l)
                return mlp.tail2(h1, out) if fold else mlp.tail(h1, out)
        # static layout: the row block is padded to its capacity; the kernels skip the padding (count on the device)
        rd = batch.rows_dev(t) if n == batch.lay_src(1, dg.schema.type_id[t]) else None
        if fold and lazy and dict.get(x_dict, t) is None and dg.x[t].shape[1] <= 20:
            # narrow features (the 20-wide SNP rows): the x[n_id] slicing happens inside the fused two-layer kernel
            return mlp.hidden(dg.x[t], out, rd, ids=batch.n_id(t))
        return mlp.hidden(x_dict[t], out, rd) if fold else mlp(x_dict[t], out, rd)

    def _layer_input(self, batch: SampledBatch, l: int):
        """Preallocated type-major input matrix of layer l and one RowBlock per node type that has rows in it."""
        m, sc = batch.meta, self.schema
        total = int(m.src_base[l - 1][sc.NT])
        buf = torch.empty(max(total, 1), self.hidden, device=self.lin.weight.device)
        blocks = {name: ops.RowBlock(buf, int(m.src_base[l - 1][t]), int(m.lay_src[l - 1][t]))
                  for t, name in enumerate(sc.node_types) if int(m.lay_src[l - 1][t])}
        return buf, blocks

    def _embed_all(self, batch: SampledBatch, x_dict, blocks=None, fold=False):
        """All feature MLPs (model.py:56-60).  The three GO types share ``go_feat_mlp`` (model.py:58-60): their
        rows go through it as ONE matrix.  ``blocks``: RowBlocks of the first layer's input to write into."""
        h = {}
        blocks = blocks or {}
        # (n_nodes first: touching a lazy x_dict entry would gather that type's rows a second time)
        go = [t for t in self.node_types if t in GO_TYPES and t in x_dict and
              (batch.n_nodes[t] if t in batch.n_nodes else x_dict[t].shape[0]) > 0]
        if len(go) > 1:
            lazy = getattr(x_dict, 'kgw_batch', None) is batch and all(t in batch.dg.x for t in go) and \
                len({batch.dg.x[t].shape[1] for t in go}) == 1
            fused_jobs = None
            if lazy and fold:
                mlp = self.go_feat_mlp
                fj = [(batch.dg.x[t], batch.n_id(t)) for t in go]
                if ops.mlp2_gathered_ok(fj, mlp.FC_hidden.weight, mlp.FC_hidden2.weight):
                    fused_jobs = fj         # gather + both hidden layers in one launch (kgw_mlp2w_fwd)
            if fused_jobs is not None:
                ns = [batch.n_nodes[t] for t in go]
                xg = xs = None
            elif lazy:            # gather the three types' rows straight into one matrix (no concatenation copy)
                ns = [batch.n_nodes[t] for t in go]
                xg = torch.empty(sum(ns), batch.dg.x[go[0]].shape[1], device=self.lin.weight.device)
                off, jobs = 0, []
                for t, n in zip(go, ns):
                    jobs.append((batch.dg.x[t], batch.n_id(t), xg[off:off + n]))
                    off += n
                gather_rows_multi(jobs)                       # one launch for the three types
                xs = None
            else:
                xs = [x_dict[t] for t in go]
                ns = [x.shape[0] for x in xs]
                xg = torch.cat(xs, 0)
            out = None
            bl = [blocks.get(t) for t in go]
            if all(b is not None and b.n == n for b, n in zip(bl, ns)) and \
                    all(bl[k + 1].lo == bl[k].lo + bl[k].n for k in range(len(bl) - 1)):
                out = ops.RowBlock(bl[0].buf, bl[0].lo, sum(ns))          # the GO blocks are adjacent: one output
            if fused_jobs is not None:
                mlp = self.go_feat_mlp
                y = ops.mlp2_gathered(fused_jobs, mlp.FC_hidden.weight, mlp.FC_hidden.bias, mlp.FC_hidden2.weight,
                                      mlp.FC_hidden2.bias, out)
            else:
                y = self.go_feat_mlp.hidden(xg, out) if fold else self.go_feat_mlp(xg, out)
            for t, piece in zip(go, ops.split_rows(y, ns)):
                h[t] = piece
        for t in self.node_types:
            if t in x_dict and t not in h:
                h[t] = self._embed(batch, x_dict, t, blocks.get(t), fold)
        return h

    def _combine_relations(self, o: torch.Tensor) -> torch.Tensor:
        """HeteroConv(aggr) over the relation axis of per-relation outputs o [R, rows, C] (PyG hetero_conv.group:
        stack + reduce; kgwas/model.py:47): used for 'min' / 'max', which cannot be folded into one GEMM."""
        if o.shape[0] == 1:
            return o[0]
        return getattr(torch, self.aggr)(o, dim=0).values

    def _sage_layers(self, batch: SampledBatch, h: Dict[str, torch.Tensor], hbuf=None):
        """SAGE backbone (kgwas/model.py:38,74-75): out_d = relu(sum_r [lin_l^r(mean_{j->i} h_s[j]) + lin_r^r(h_d[i])]).
        The neighbour mean IS the attention aggregate with all logits equal: the same kernels run with zero attention
        vectors (alpha = 1/deg; rows without in-edges stay zero like PyG's mean), forward and backward; the root
        term is one more product per destination type with the relation-summed lin_r."""
        sc, m, C = self.schema, batch.meta, self.hidden
        dev = self.lin.weight.device
        zeros = torch.zeros(sc.NR, C, device=dev)
        for l in range(1, self.num_layers + 1):
            P: SagePack = self.live_packs[l - 1]
            rng = self._dst_range[l - 1]
            parts, spans = [], []
            for t, name in enumerate(sc.node_types):
                ns = int(m.lay_src[l - 1][t])
                if ns:
                    if name not in h or h[name].shape[0] < ns:
                        raise RuntimeError(f'layer {l}: node type {name!r} takes part in the layer but has no '
                                           f'incoming relation to produce its layer-{l - 1} state')
                    parts.append(h[name] if h[name].shape[0] == ns else h[name][:ns])
                    spans.append((int(m.src_base[l - 1][t]), ns))
            if hbuf is None:
                hbuf, _ = self._layer_input(batch, l)
            H = ops.join_blocks(hbuf, spans, parts) if len(parts) != 1 or parts[0].shape[0] != hbuf.shape[0] else parts[0]
            hbuf = None
            Z, _, _ = ops.gat_aggregate(batch, l, H, zeros, zeros, self.negative_slope, self.temperature)
            h_next = {}
            for t, name in enumerate(sc.node_types):
                rows = int(m.lay_rows[l - 1][t])
                if not rows:
                    continue
                lo, hi = rng[t]
                R = hi - lo
                z0 = int(m.z_base[l - 1][t])
                if self.aggr in ('min', 'max'):
                    zr = Z[z0:z0 + rows * R].view(rows, R, C).transpose(0, 1)                    # [R, rows, C]
                    if ops.LIBRARY_GEMM.own_first:          # (library-free form: one own-kernel product per relation and term)
                        hr = h[name][:rows].contiguous()
                        zero = torch.zeros(C, device=dev)
                        o = torch.stack([ops.linear_act(zr[r].contiguous(), P.w_l_t[lo + r], P.bias[lo + r], relu=False) +
                                         ops.linear_act(hr, P.w_r_t[lo + r], zero, relu=False) for r in range(R)])
                    else:
                        ops.LIBRARY_GEMM.note('sage min/max per-relation outputs', R, rows, C)
                        o = torch.baddbmm(P.bias[lo:hi].unsqueeze(1), zr, P.w_l_t[lo:hi]) + \
                            torch.matmul(h[name][:rows].unsqueeze(0), P.w_r_t[lo:hi])
                    h_next[name] = torch.relu(self._combine_relations(o))
                    continue
                x = Z[z0:z0 + rows * R].view(rows, R * C)
                y = ops.linear_act(x, P.w_l_t[lo:hi].reshape(R * C, C), P.bias[lo:hi].sum(0), relu=False)
                if ops.LIBRARY_GEMM.own_first:
                    y = y + ops.linear_act(h[name][:rows].contiguous(), P.w_r_t[lo:hi].sum(0), torch.zeros(C, device=dev), relu=False)
                else:
                    ops.LIBRARY_GEMM.note('sage root term', rows, C, C)
                    y = y + h[name][:rows] @ P.w_r_t[lo:hi].sum(0)        # root term: sum_r lin_r^r(h_d[i])
                if self.aggr == 'mean':
                    y = y * (1.0 / R)
                h_next[name] = torch.relu(y)
            h = h_next
        return h, []

    def _all_layer_params(self, batch: SampledBatch, folded: bool):
        """_layer_params for every layer with the relation vectors of ALL layers from one launch (ops.rel_vectors_all): they
        depend on the parameters only.  Their backward then is one launch too, after the first layer's."""
        sc, m = self.schema, batch.meta
        dev = self.lin.weight.device
        blocks_all, zeros = [], []
        for l in range(1, self.num_layers + 1):
            rng = self._dst_range[l - 1]
            tys = [t for t in range(sc.NT) if int(m.lay_rows[l - 1][t])]
            blocks_all.append([(rng[t][0], rng[t][1], int(m.z_base[l - 1][t]), int(m.lay_rows[l - 1][t])) for t in tys])
            zeros.append(ops.aggregate_workspace(batch, l, dev))
        rv = ops.rel_vectors_all([self.live_packs[l] for l in range(self.num_layers)], blocks_all, zeros)
        return [self._layer_params(batch, l, folded, relvec=rv[l - 1], zbuf=zeros[l - 1]) for l in range(1, self.num_layers + 1)]

    def _layer_params(self, batch: SampledBatch, l: int, folded: bool, relvec=None, zbuf=None):
        """Everything layer l needs that depends on the PARAMETERS only (and on the batch's static row counts): the destination
        blocks, the zeroed aggregate workspace, u_r / v_r / summed biases (ops.rel_vectors) and, for a folded layer 1, the
        fold of FC_output into them.  No activation enters."""
        sc, m = self.schema, batch.meta
        P: RelationPack = self.live_packs[l - 1]
        rng = self._dst_range[l - 1]
        # destination blocks of the layer: one per node type that receives messages
        tys = [t for t in range(sc.NT) if int(m.lay_rows[l - 1][t])]
        blocks = [(rng[t][0], rng[t][1], int(m.z_base[l - 1][t]), int(m.lay_rows[l - 1][t])) for t in tys]
        # u_r = W_src^T att_src ; v_r = W_dst^T att_dst (W_src^T att_dst for same-type relations) and the summed
        # bias of every destination block: one launch
        # ... and the zero fill of the aggregate's workspace rides in the same launch
        zws = ops.aggregate_workspace(batch, l, self.lin.weight.device)
        # (the weights reach the transform / the fold THROUGH that node: their gradient is added inside its backward kernel)
        if relvec is not None:                          # (all layers' vectors came out of ONE launch: _all_layer_params)
            U, V, bsum, Wv = relvec
            zws = zbuf
        else:
            U, V, bsum, Wv = ops.rel_vectors(P, blocks, zero=zws, pass_weights=True)
        Wp = gam = kap = None
        if folded and l == 1:
            # (an MLP no live layer-1 relation touches -- the GO one of a 1-layer model -- stays out of the graph: like
            # in the reference its parameters get no gradient and Adam skips them)
            mlps = (self.snp_feat_mlp, self.gene_feat_mlp, self.go_feat_mlp)
            fc = []
            for k, mm in enumerate(mlps):
                used = k in self._fold_used
                fc += [mm.FC_output.weight if used else mm.FC_output.weight.detach(),
                       mm.FC_output.bias if used else mm.FC_output.bias.detach()]
            U, V, kap, Wp, gam = ops.fold_fc_output_hip(P, U, V, fc, self._fold_tab, weight=Wv)
        return tys, blocks, zws, U, V, bsum, Wv, kap, Wp, gam

    def _fused_layers(self, batch: SampledBatch, h: Dict[str, torch.Tensor], want_attention=False, hbuf=None,
                      last_premasked=False, folded=False, prep=None):
        """``folded``: h holds the feature MLPs' hidden state h2 (``_embed_all(fold=True)``), not their output: layer 1 runs
        with FC_output folded into its relation parameters (ops.fold_fc_output_hip).  ``prep``: per layer, what
        ``_layer_params`` returns, computed ahead by the caller."""
        if self.backbone == 'SAGE':
            if want_attention:
                raise NotImplementedError('attention weights exist for the GAT backbone only (kgwas/model.py:65-72)')
            return self._sage_layers(batch, h, hbuf)
        sc = self.schema
        m = batch.meta
        C = self.hidden
        dev = self.lin.weight.device
        attn = []
        if prep is None and self.num_layers > 1 and _RELVEC_ALL:
            prep = self._all_layer_params(batch, folded)
        for l in range(1, self.num_layers + 1):
            P: RelationPack = self.live_packs[l - 1]
            tys, blocks, zws, U, V, bsum, Wv, kap, Wp, gam = prep[l - 1] if prep is not None else self._layer_params(batch, l, folded)
            # layer input, type-major (src_base): every type that sends or receives messages in this layer
            parts, spans = [], []
            for t, name in enumerate(sc.node_types):
                ns = int(m.lay_src[l - 1][t])
                if ns:
                    if name not in h or h[name].shape[0] < ns:
                        raise RuntimeError(f'layer {l}: node type {name!r} takes part in the layer but has no '
                                           f'incoming relation to produce its layer-{l - 1} state')
                    parts.append(h[name] if h[name].shape[0] == ns else h[name][:ns])
                    spans.append((int(m.src_base[l - 1][t]), ns))
            if hbuf is None:
                hbuf, _ = self._layer_input(batch, l)
            H = ops.join_blocks(hbuf, spans, parts) if len(parts) != 1 or parts[0].shape[0] != hbuf.shape[0] else parts[0]
            # (from layer 2 on, H is the previous layer's ReLU output: with the fused transform its backward is folded
            # into this node's)
            fused = self.aggr in ('sum', 'mean')
            Z, stat, e_edge = ops.gat_aggregate(batch, l, H, U, V, self.negative_slope, self.temperature,
                                                relu_input=((l > 1 or folded) and fused), zbuf=zws, logit_bias=kap)
            if want_attention:
                attn.append(ops.edge_alpha(batch, l, stat, e_edge, self.temperature))
            if not fused:
                # 'min' / 'max' over relations: per-relation outputs (lin_src + bias, conv.py:138-190) as one batched
                # product per destination type, then the reduction over the relation axis and the ReLU (model.py:74-75)
                hbuf = None
                outs = []
                for (lo, hi, z0, rows) in blocks:
                    R = hi - lo
                    zr = Z[z0:z0 + rows * R].view(rows, R, C).transpose(0, 1)
                    if ops.LIBRARY_GEMM.own_first:
                        o = torch.stack([ops.linear_act(zr[r].contiguous(), Wv[lo + r], P.bias[lo + r], relu=False) for r in range(R)])
                    else:
                        ops.LIBRARY_GEMM.note('gat min/max per-relation outputs', R, rows, C)
                        o = torch.baddbmm(P.bias[lo:hi].unsqueeze(1), zr, Wv[lo:hi])
                    outs.append(torch.relu(self._combine_relations(o)))
                h = {sc.node_types[t]: o for t, o in zip(tys, outs)}
                continue
            # per-relation linear maps + bias + relation sum + ReLU: one GEMM per destination type, one autograd node
            hbuf, nxt = self._layer_input(batch, l + 1) if l < self.num_layers else (None, {})
            outs = ops.layer_transform(P, Z, blocks, [nxt.get(sc.node_types[t]) for t in tys],
                                       premasked=(l < self.num_layers) or last_premasked, bias_sum=bsum,
                                       weight=Wp if Wp is not None else Wv, gamma=gam, stat=stat if gam is not None else None)
            if self.aggr == 'mean':
                # mean over the relations of a destination type = the sum scaled by 1/R; relu(s/R) = relu(s)/R, and the
                # positive scale commutes with the folded ReLU masks
                hbuf = None
                outs = [o * (1.0 / (hi - lo)) for o, (lo, hi, _, _) in zip(outs, blocks)]
            h_next = {sc.node_types[t]: o for t, o in zip(tys, outs)}
            h = h_next
        return h, attn

    def forward(self, x_dict, edge_index_dict, batch_size, genotype=None, return_h=False,
                return_attention_weights=False):
        if return_attention_weights and not return_h:           # model.py:65-72,80-81 (mean attention per layer)
            return self._forward_with_attention(x_dict, edge_index_dict, batch_size)
        batch: Optional[SampledBatch] = getattr(x_dict, 'kgw_batch', None) or getattr(edge_index_dict, 'kgw_batch', None)
        if batch is None:
            batch = self._block_from_coo(x_dict, edge_index_dict)
        hbuf, blocks = self._layer_input(batch, 1)
        h = self._embed_all(batch, x_dict, blocks, fold=self.fold_fc)
        h, attn = self._fused_layers(batch, h, hbuf=hbuf, folded=self.fold_fc)
        snp = h['SNP']
        out = self._readout(snp[:batch_size])
        if return_h:                                            # model.py:78-79
            return self.ReLU(out), snp[:batch_size, :self.hidden_logical]
        if self.no_relu:                                        # model.py:83-84
            return out
        return self.ReLU(out)                                   # model.py:86

    def _readout(self, h):
        """self.lin(h) (kgwas/model.py:50,83-86).  For the reference's out_channels == 1 (kgwas.py:52) the Linear(128 -> 1) is a
        row-wise dot product: elementwise multiply + row sum, not a library GEMV."""
        if self.lin.out_features == 1:
            return (h * self.lin.weight.view(1, -1)).sum(1, keepdim=True) + self.lin.bias
        ops.LIBRARY_GEMM.note('read-out', h.shape[0], h.shape[1], self.lin.out_features)
        return self.lin(h)

    @torch.no_grad()
    def hot_path_attention(self, batch: SampledBatch):
        """Per layer, the softmax attention weight of every edge the TRAINING path aggregates (live relations, hop-pruned
        destination rows; local edge order of ``batch``) -- what the fused kernels actually use, as opposed to the
        reference-shaped ``forward(return_attention_weights=True)`` which runs all relations over all rows."""
        hbuf, blocks = self._layer_input(batch, 1)
        h = self._embed_all(batch, batch.x_dict, blocks, fold=self.fold_fc)
        _, attn = self._fused_layers(batch, h, want_attention=True, hbuf=hbuf, folded=self.fold_fc)
        return attn

    def forward_loss(self, x_dict, edge_index_dict, batch_size, n_id, y_all, w_all, mlp_out=None, unit_grad=False):
        """The training step's forward (kgwas/kgwas.py:137-145): HeteroGNN.forward followed by
        mean(w_all[n_id] * (pred - y_all[n_id])**2), with the read-out Linear + ReLU (model.py:86) and the loss fused
        into one node.  Returns (loss [float64 scalar], pred [batch_size]).  ``mlp_out`` (list): receives the feature MLPs'
        output tensors -- the cut between the two halves of a backward pass whose first half's gradients are all-reduced
        while the second half runs (multi-GPU GraphTrainStep).  ``unit_grad``: the caller will backpropagate exactly
        ``loss.backward()`` (gradient 1): the read-out node then does its forward and backward in two launches."""
        batch: Optional[SampledBatch] = getattr(x_dict, 'kgw_batch', None) or getattr(edge_index_dict, 'kgw_batch', None)
        if batch is None:
            batch = self._block_from_coo(x_dict, edge_index_dict)
        if self.lin.out_features != 1:
            raise NotImplementedError('the fused read-out + loss is for out_channels == 1 (kgwas/kgwas.py:52)')
        hbuf, blocks = self._layer_input(batch, 1)
        gat = self.backbone == 'GAT' and self.aggr in ('sum', 'mean')
        prep = None
        # what depends on the parameters only -- the relation vectors of all layers, the FC_output fold -- is prepared FIRST and
        # handed to the first gene Linear's kgw_gemm3 launch as rider blocks (ops.ParamRiders); whatever no launch took is
        # launched the ordinary way when the scope closes, before the layers read it
        riders = ops.ParamRiders() if (ops._G3_RIDERS and self.backbone == 'GAT' and self.num_layers > 1 and _RELVEC_ALL) else None
        with ops.param_riders_scope(riders):
            if riders is not None:
                prep = self._all_layer_params(batch, self.fold_fc)
            h = self._embed_all(batch, x_dict, blocks, fold=self.fold_fc)
        self.last_riders_taken = riders.taken if riders is not None else 0
        if mlp_out is not None:
            mlp_out.extend(h.values())
        h, _ = self._fused_layers(batch, h, hbuf=hbuf, last_premasked=gat, folded=self.fold_fc, prep=prep)
        return ops.readout_weighted_mse(h['SNP'], self.lin.weight, self.lin.bias, n_id, y_all, w_all, batch_size,
                                        relu=not self.no_relu, h_is_relu=gat, unit_grad=unit_grad)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _forward_all_relations(self, batch: SampledBatch, raw: bool):
        """Both layers over EVERY relation and every row of a full-graph block (no structural pruning), the way the
        reference runs a batch (kgwas/model.py:64-75): returns (final x_dict, [per layer: per-edge attention of the whole
        block, local edge order]).  ``raw``: the export's variant (kgwas/utils.py:446-461) -- raw leaky_relu logits as
        message weights (conv.py:221-228) and no ReLU between the layers (utils.py:460)."""
        sc, m, C = self.schema, batch.meta, self.hidden
        dev = self.lin.weight.device
        h = self._embed_all(batch, batch.x_dict)
        per_layer = []
        for l in range(1, self.num_layers + 1):
            U = torch.zeros(sc.NR, C, device=dev)
            V = torch.zeros(sc.NR, C, device=dev)
            for pack in (self.live_packs[l - 1], self.dead_packs[l - 1]):
                if len(pack.rel_ids):
                    u, v = ops.rel_vectors(pack)           # rows of the pack's relations, zeros elsewhere
                    U += u; V += v
            parts = []
            for t, name in enumerate(sc.node_types):
                ns = int(m.lay_src[l - 1][t])
                if ns:
                    if name not in h or h[name].shape[0] < ns:
                        raise RuntimeError(f'layer {l}: node type {name!r} has no layer-{l - 1} state')
                    parts.append(h[name][:ns])
            H = torch.cat(parts, 0)
            Z, stat, e_edge = ops.gat_aggregate(batch, l, H, U, V, self.negative_slope, self.temperature, raw_weights=raw)
            per_layer.append(e_edge if raw else ops.edge_alpha(batch, l, stat, e_edge, self.temperature))
            h_next = {}
            for t, name in enumerate(sc.node_types):
                n, R = int(m.lay_rows[l - 1][t]), int(sc.R_dst[t])
                if n == 0 or R == 0:
                    continue
                z0 = int(m.z_base[l - 1][t])
                x = Z[z0:z0 + n * R].view(n, R * C)
                ws, bs = [], 0
                for r in sc.rels_by_dst[t]:                # slot order of the Z columns
                    which, i = self._slot[l - 1][r]
                    pack = self.live_packs[l - 1] if which == 'live' else self.dead_packs[l - 1]
                    ws.append(pack.w_src_t[i])
                    bs = bs + pack.bias[i]
                y = ops.linear(x, torch.cat(ws, 0), bs, w_kn=True)
                h_next[name] = y if raw else torch.relu(y)                   # model.py:75 / no ReLU at utils.py:460
            h = h_next
        return h, per_layer

    @torch.no_grad()
    def raw_attention_full_graph(self, graph: HeteroGraph, device=None):
        """Per layer, per relation: (edge_index [2,E] global ids, raw attention [E]) over the WHOLE graph, the way
        the reference's export computes them (kgwas/utils.py:437-461): every relation of every layer takes part,
        the attention is leaky_relu(alpha_j + alpha_i) without softmax (conv.py:217-223 under
        return_raw_attention_weights), the layer output fed to the next layer is the sum of messages weighted by
        those raw values (conv.py:227-228) plus bias, summed over relations, and -- unlike HeteroGNN.forward -- NO
        ReLU is applied between the layers (utils.py:460)."""
        from .sampler import BatchBuffers, DeviceGraph, finish_sample, sample_into
        if self.backbone != 'GAT':
            raise NotImplementedError('attention weights exist for the GAT backbone only (kgwas/utils.py:437-461)')
        if self.aggr != 'sum':
            raise NotImplementedError("the attention export is built for gnn_aggr='sum' (the reference default)")
        dev = torch.device(device) if device is not None else self.lin.weight.device
        sc = self.schema
        dg = DeviceGraph.get(graph, self.num_layers, dev, full_graph=True).with_all_relations_live()
        buf = BatchBuffers(dg)
        sample_into(dg, buf, None, 0)
        batch = finish_sample(dg, buf, sc.node_types[0], dg.n_nodes[0])
        m = batch.meta
        _, per_layer = self._forward_all_relations(batch, raw=True)
        ei = batch.edge_index_dict                         # full graph: local ids are the global ids
        seg_ptr = buf.seg_ptr
        layers = []
        for e_edge in per_layer:
            att = OrderedDict()
            for r, et in enumerate(self.edge_types):
                a, b = int(m.seg_off[0][r]), int(m.seg_off[0][r + 1])
                e0, e1 = (int(seg_ptr[a]), int(seg_ptr[b])) if b > a else (0, 0)
                att[et] = (ei[et], e_edge[e0:e1])
            layers.append(att)
        return layers

    @torch.no_grad()
    def _forward_with_attention(self, x_dict, edge_index_dict, batch_size):
        """forward(return_attention_weights=True) (kgwas/model.py:65-72,80-81): the reference runs EVERY edge type over
        every edge of the batch in both layers and returns the mean attention of all of them per layer, so this path
        does too -- the batch's subgraph as a full block with all relations live, no hop pruning (inference only)."""
        from .sampler import BatchBuffers, DeviceGraph, finish_sample, sample_into
        if self.backbone != 'GAT':
            raise NotImplementedError('attention weights exist for the GAT backbone only (kgwas/model.py:65-72)')
        if self.aggr != 'sum':
            raise NotImplementedError("attention weights are built for gnn_aggr='sum' (the reference default)")
        dev = next(iter(x_dict.values())).device if len(x_dict) else self.lin.weight.device
        g = HeteroGraph()
        for t in self.node_types:
            if t in x_dict:
                g[t].x = x_dict[t]
            else:
                g[t].num_nodes_ = 0
        for et in self.edge_types:
            ei = edge_index_dict.get(et)
            g[et].edge_index = ei if ei is not None else torch.zeros(2, 0, dtype=torch.long)
        dg = DeviceGraph(g, self.num_layers, dev, full_graph=True).with_all_relations_live()
        buf = BatchBuffers(dg)
        sample_into(dg, buf, None, 0)
        sc = self.schema
        batch = finish_sample(dg, buf, sc.node_types[0], dg.n_nodes[0])
        h, per_layer = self._forward_all_relations(batch, raw=False)
        self.last_attention = per_layer
        out = self.ReLU(self._readout(h['SNP'][:batch_size]))
        return out, [a.mean() if a.numel() else a.sum() for a in per_layer]

    # ------------------------------------------------------------------------------------------
    def _block_from_coo(self, x_dict, edge_index_dict) -> SampledBatch:
        """Plain (x_dict, edge_index_dict) inputs: every row of every type is computed in every layer,
        exactly like the reference on a PyG batch / the full graph (kgwas/utils.py:446-461)."""
        dev = next(iter(x_dict.values())).device
        g = HeteroGraph()
        for t in self.node_types:
            g[t].num_nodes_ = int(x_dict[t].shape[0]) if t in x_dict else 0
        for et in self.edge_types:
            ei = edge_index_dict.get(et)
            g[et].edge_index = ei if ei is not None else torch.zeros(2, 0, dtype=torch.long)
        return sample_full_graph(g, self.num_layers, dev)

    # --- reference-named parameters (checkpoints, tests) ------------------------------------------
    def _rel_items(self, grad: bool = False):
        """Yield (reference key, tensor | None | 'lazy') for every relation tensor of every layer."""
        for l in range(self.num_layers):
            for r, et in enumerate(self.edge_types):
                which, i = self._slot[l][r]
                pack = self.live_packs[l] if which == 'live' else self.dead_packs[l]
                for f in self.rel_fields:
                    yield f'convs.{l}.convs.{edge_key(et)}.{f}', pack.get(i, f, grad)

    def _dense_items(self, grad: bool = False, keep_vars: bool = False):
        """(reference key, tensor at the model's own width) of the feature MLPs and the read-out."""
        cl = self.hidden_logical
        for prefix, mod in (('snp_feat_mlp', self.snp_feat_mlp), ('go_feat_mlp', self.go_feat_mlp),
                            ('gene_feat_mlp', self.gene_feat_mlp), ('lin', self.lin)):
            for n, p in mod.named_parameters():
                t = p.grad if grad else (p if keep_vars else p.detach())
                if t is not None and cl != self.hidden:
                    if prefix == 'lin':
                        t = t[:, :cl] if n == 'weight' else t
                    elif n == 'FC_hidden.weight':
                        t = t[:cl]
                    elif n.endswith('weight'):
                        t = t[:cl, :cl]
                    else:
                        t = t[:cl]
                yield f'{prefix}.{n}', t

    def named_reference_tensors(self, grad: bool = False) -> "OrderedDict[str, Optional[torch.Tensor]]":
        """Parameters (or their gradients) under the reference's names.  Lazy lin_dst of same-type relations
        is omitted; structurally dead relations have gradient None."""
        out = OrderedDict()
        for k, v in self._rel_items(grad):
            if isinstance(v, str):
                continue
            out[k] = v
        for k, t in self._dense_items(grad):
            out[k] = t
        return out

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        out = OrderedDict() if destination is None else destination
        for k, v in self._rel_items():
            out[prefix + k] = torch.nn.parameter.UninitializedParameter() if isinstance(v, str) else v.clone()
        for k, t in self._dense_items(keep_vars=keep_vars):
            out[prefix + k] = t
        return out

    def load_state_dict(self, state_dict, strict=True, **kw):
        want = {}
        for l in range(self.num_layers):
            for r, et in enumerate(self.edge_types):
                for f in self.rel_fields:
                    want[f'convs.{l}.convs.{edge_key(et)}.{f}'] = (l, r, f)
        rest = OrderedDict()
        seen = set()
        for k, v in state_dict.items():
            k2 = k.replace('<', '').replace('>', '').replace('___', '__')     # PyG >= 2.4 key style
            if k2 in want:
                l, r, f = want[k2]
                seen.add(k2)
                if isinstance(v, torch.nn.parameter.UninitializedParameter):
                    continue                  # never-materialised lazy lin_dst of same-type relations
                which, i = self._slot[l][r]
                (self.live_packs[l] if which == 'live' else self.dead_packs[l]).set(i, f, v)
            else:
                rest[k2] = v
        missing = [k for k, (l, r, f) in want.items() if k not in seen and
                   not (f == 'lin_dst.weight' and self.edge_types[r][0] == self.edge_types[r][2])]
        unexpected = []
        known = ('snp_feat_mlp.', 'go_feat_mlp.', 'gene_feat_mlp.', 'lin.')
        with torch.no_grad():
            for k, dst in self._dense_items():              # (views of the padded storage at the model's own width)
                if k in rest:
                    if tuple(rest[k].shape) != tuple(dst.shape):
                        raise RuntimeError(f'load_state_dict: {k} has shape {tuple(rest[k].shape)}, the model expects {tuple(dst.shape)}')
                    dst.copy_(rest[k])
                else:
                    missing.append(k)
        have = {k for k, _ in self._dense_items()}
        unexpected += [k for k in rest if k.startswith(known) and k not in have]
        unexpected += [k for k in rest if not k.startswith(known)]
        if strict and (missing or unexpected):
            raise RuntimeError(f'load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}')
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)
