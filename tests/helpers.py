"""Shared test plumbing: run the CPU oracle on exactly the subgraph / weights the HIP path used."""
from collections import OrderedDict

import numpy as np
import torch

from oracle.gat_oracle import HeteroGNNOracle


def product_dims(model):
    return (model.snp_feat_mlp.FC_hidden.in_features, model.gene_feat_mlp.FC_hidden.in_features,
            model.go_feat_mlp.FC_hidden.in_features)


def oracle_from_product(model, dtype=torch.float64):
    snp, gene, go = product_dims(model)
    o = HeteroGNNOracle(model.edge_types, getattr(model, 'hidden_logical', model.hidden), model.lin.out_features, model.num_layers,
                        getattr(model, 'backbone', 'GAT'), getattr(model, 'aggr', 'sum'),
                        snp, gene, go, 1, no_relu=model.no_relu, dtype=dtype)
    sd = OrderedDict()
    for k, v in model.state_dict().items():
        if isinstance(v, torch.nn.parameter.UninitializedParameter):
            continue
        sd[k] = v.detach().cpu().to(dtype)
    missing = o.load_state_dict(sd, strict=True)
    return o


def batch_cpu(batch, dtype=torch.float64):
    x = {t: v.detach().cpu().to(dtype) for t, v in batch.x_dict.items()}
    ei = OrderedDict((k, v.cpu()) for k, v in batch.edge_index_dict.items())
    return x, ei


def grads_by_name(model):
    """Gradients under the reference's parameter names (the product keeps packed tensors internally)."""
    if hasattr(model, 'named_reference_tensors'):
        return {n: (g.detach().cpu() if g is not None else None)
                for n, g in model.named_reference_tensors(grad=True).items()}
    return {n: (p.grad.detach().cpu() if p.grad is not None else None) for n, p in model.named_parameters()}


def params_by_name(model):
    if hasattr(model, 'named_reference_tensors'):
        return {n: p.detach().cpu().double().clone() for n, p in model.named_reference_tensors().items()}
    return {n: p.detach().cpu().double().clone() for n, p in model.named_parameters()}


def assert_close(a, b, rtol, atol, what='', rel_to_max=1e-5):
    """|a - b| <= atol + rtol*|b| + rel_to_max*max|b|.  The last term: an fp32 sum whose terms cancel carries
    round-off proportional to its largest terms (~1e-7 * sum|terms|), not to the (possibly tiny) result, and
    the summation order on the GPU (chunks, half-waves, src-major lists) differs from the oracle's."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.numel() == 0:
        return
    err = (a - b).abs()
    tol = atol + rtol * b.abs() + rel_to_max * float(b.abs().max())
    bad = err > tol
    assert not bad.any(), (f'{what}: {int(bad.sum())}/{a.numel()} elements out of tolerance; max abs err '
                           f'{float(err.max()):.3e}, max |ref| {float(b.abs().max()):.3e}')


def global_edge_set(batch, et):
    """Set of (global src, global dst) pairs of relation ``et`` in a sampled batch (multiset as sorted array)."""
    ei = batch.edge_index_dict[et].cpu().numpy()
    s, _, d = et
    ns = batch.n_id(s).cpu().numpy().astype(np.int64)
    nd = batch.n_id(d).cpu().numpy().astype(np.int64)
    pairs = np.stack([ns[ei[0]], nd[ei[1]]], axis=1) if ei.shape[1] else np.zeros((0, 2), np.int64)
    order = np.lexsort((pairs[:, 0], pairs[:, 1]))
    return pairs[order]


def fold_fc_output_reference(pack, U, V, T3, c3, src_m, dst_m):
    """Test reference of kgwas_amd.ops.fold_fc_output_hip (kgw_fold_fwd / kgw_fold_bwd) in framework ops + autograd.
    Fold the last Linear of the feature MLPs, H = h2 T_m + c_m (FC_output, kgwas/model.py:15,21; m = the MLP of the node's
    type), into the layer-1 relation parameters -- exact, like aggregate-then-transform: H enters GATConv (which has no
    root term) only linearly, as the message sum_j alpha_ij H_j and through the logit projections <H_j, u_r>, <H_i, v_r>
    (kgwas/conv.py:150-152,227-228).  With layer 1 running on h2:
        U'_r = T_src U_r,  V'_r = T_dst V_r,  kappa_r = <c_src, U_r> + <c_dst, V_r>         (logits)
        W'_r = T_src W_r^T (the packed [in, out] form),  gamma_r = c_src W_r^T                (transform; gamma is added
        wherever the segment is not empty: sum_j alpha_ij = 1)
    ``U``, ``V`` [n_rels, C] by relation id (rel_vectors); ``T3`` [n_mlp, C, C] = FC_output.weight^T, ``c3`` [n_mlp, C];
    ``src_m`` / ``dst_m`` [n] long: MLP index of the source / destination type of every packed relation.
    Returns (U' [n_rels,C], V' [n_rels,C], kappa [n_rels] by relation id; W' [n,C,C], gamma [n,C] by packed slot)."""
    ids = pack.rel_ids_t
    Ui, Vi = U[ids], V[ids]
    Ts, Td = T3[src_m], T3[dst_m]
    cs, cd = c3[src_m], c3[dst_m]
    Wp = torch.bmm(Ts, pack.w_src_t)
    Up = torch.bmm(Ts, Ui.unsqueeze(-1)).squeeze(-1)
    Vp = torch.bmm(Td, Vi.unsqueeze(-1)).squeeze(-1)
    kap = (cs * Ui).sum(-1) + (cd * Vi).sum(-1)
    gam = torch.bmm(cs.unsqueeze(1), pack.w_src_t).squeeze(1)
    NR = U.shape[0]
    Uf = torch.zeros_like(U).index_copy(0, ids, Up)
    Vf = torch.zeros_like(V).index_copy(0, ids, Vp)
    kf = torch.zeros(NR, device=U.device, dtype=U.dtype).index_copy(0, ids, kap)
    return Uf, Vf, kf, Wp, gam
