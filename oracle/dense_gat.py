"""Second, INDEPENDENT derivation of the reference model: dense masked-softmax graph attention in numpy float64
(TEST INFRASTRUCTURE; see oracle/__init__.py).

``oracle/gat_oracle.py`` restates the reference op for op (index_select / scatter, the way PyG executes
kgwas/conv.py:122-228).  This file shares no code and no formulation with it: it starts from the equations of the GAT
paper the operator implements (Velickovic et al., ICLR 2018, eqs. 1-4, single head) --

    e_ij    = LeakyReLU(a_src . (W_src h_j) + a_dst . (W_dst h_i))        j in N(i)
    alpha_ij = exp(e_ij) / sum_{k in N(i)} exp(e_ik)
    h'_i    = sum_{j in N(i)} alpha_ij W_src h_j + b

-- written with a dense [n_dst, n_src] edge-MULTIPLICITY matrix A per relation (duplicate edges count twice, like PyG's
edge list), plus the pieces the reference adds around it: the 1e-16 in the softmax denominator (PyG ``softmax``,
kgwas/conv.py:223), same-type relations using lin_src for both roles (conv.py:138), the relation sum and ReLU of the
stack (kgwas/model.py:74-75), the three feature MLPs (model.py:10-22,56-60), the read-out (model.py:86) and the
LD-weighted loss (kgwas/kgwas.py:145).  Gradients are not derived here at all: ``directional_derivative`` differentiates
the dense loss numerically (central differences in float64), which checks the autograd of the scatter formulation
against an implementation that has no backward pass to get wrong.

Only usable on tiny graphs (dense n_dst x n_src matrices): tests/golden/gat_case.py.
"""
from __future__ import annotations

import numpy as np

GO_TYPES = ('CellularComponent', 'BiologicalProcess', 'MolecularFunction')


def _key(et):
    return '__'.join(et)


def _mlp(P, prefix, x):
    h = np.maximum(x @ P[f'{prefix}.FC_hidden.weight'].T + P[f'{prefix}.FC_hidden.bias'], 0.0)
    h = np.maximum(h @ P[f'{prefix}.FC_hidden2.weight'].T + P[f'{prefix}.FC_hidden2.bias'], 0.0)
    return h @ P[f'{prefix}.FC_output.weight'].T + P[f'{prefix}.FC_output.bias']


def multiplicity(edge_index, n_dst, n_src):
    A = np.zeros((n_dst, n_src), dtype=np.float64)
    ei = np.asarray(edge_index, dtype=np.int64).reshape(2, -1)
    np.add.at(A, (ei[1], ei[0]), 1.0)
    return A


def gat_relation(h_src, h_dst, A, W_src, W_dst, att_src, att_dst, bias, slope=0.2, temperature=1.0):
    """One relation: returns (out [n_dst, C], alpha [n_dst, n_src] = attention of ONE edge j -> i, 0 where no edge)."""
    zs = h_src @ W_src.T
    zd = zs if W_dst is None else h_dst @ W_dst.T          # same-type: lin_src plays both roles (conv.py:138)
    e = (zd @ att_dst)[:, None] + (zs @ att_src)[None, :]
    e = np.where(e > 0, e, slope * e) / temperature
    mask = A > 0
    emax = np.where(mask, e, -np.inf).max(axis=1, initial=-np.inf)
    emax = np.where(np.isfinite(emax), emax, 0.0)
    w = np.where(mask, np.exp(np.where(mask, e - emax[:, None], 0.0)), 0.0)
    den = (A * w).sum(axis=1, keepdims=True) + 1e-16
    alpha = w / den                                         # per single edge; a (i, j) pair of multiplicity m gets m of them
    return (A * alpha) @ zs + bias, alpha


def forward(P, x_dict, edge_index_dict, n_nodes, num_layers, batch_size, slope=0.2, collect=None):
    """The stacked model on raw features.  P: parameters under the reference's state_dict names (numpy float64).
    ``collect`` (dict): filled with per-layer activations ``x{l}`` and per-edge attention ``alpha{l}`` ({edge type: [E]})."""
    h = {'SNP': _mlp(P, 'snp_feat_mlp', x_dict['SNP']), 'Gene': _mlp(P, 'gene_feat_mlp', x_dict['Gene'])}
    for t in GO_TYPES:
        if t in x_dict:
            h[t] = _mlp(P, 'go_feat_mlp', x_dict[t])
    if collect is not None:
        collect['x0'] = {k: v.copy() for k, v in h.items()}
    for l in range(num_layers):
        out = {}
        alphas = {}
        for et, ei in edge_index_dict.items():
            s, _, d = et
            pre = f'convs.{l}.convs.{_key(et)}.'
            A = multiplicity(ei, n_nodes[d], n_nodes[s])
            W_dst = P.get(pre + 'lin_dst.weight') if s != d else None
            o, alpha = gat_relation(h[s], h[d], A, P[pre + 'lin_src.weight'], W_dst, P[pre + 'att_src'].reshape(-1),
                                    P[pre + 'att_dst'].reshape(-1), P[pre + 'bias'], slope)
            out[d] = out[d] + o if d in out else o
            ei = np.asarray(ei, dtype=np.int64).reshape(2, -1)
            alphas[et] = alpha[ei[1], ei[0]]
        h = {k: np.maximum(v, 0.0) for k, v in out.items()}
        if collect is not None:
            collect[f'x{l + 1}'] = {k: v.copy() for k, v in h.items()}
            collect[f'alpha{l + 1}'] = alphas
    pred = np.maximum(h['SNP'] @ P['lin.weight'].T + P['lin.bias'], 0.0)
    return pred[:batch_size]


def loss(P, x_dict, edge_index_dict, n_nodes, num_layers, batch_size, y, w):
    pred = forward(P, x_dict, edge_index_dict, n_nodes, num_layers, batch_size).reshape(-1)
    return float(np.mean(w * (pred - y) ** 2))


def directional_derivative(P, direction, eps, *args):
    """(L(P + eps v) - L(P - eps v)) / (2 eps) for a direction v given as {name: array} (missing names = 0)."""
    plus = {k: v + eps * direction[k] if k in direction else v for k, v in P.items()}
    minus = {k: v - eps * direction[k] if k in direction else v for k, v in P.items()}
    return (loss(plus, *args) - loss(minus, *args)) / (2.0 * eps)
