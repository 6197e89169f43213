#!/usr/bin/env python
"""Generate tests/golden/gat_small.npz: committed forward / backward / sampler vectors of the tiny case in
tests/golden/gat_case.py.

PyG is not installable here, so the vectors come from the CPU restatement (oracle/gat_oracle.py, float64) -- but only
after a SECOND, independently derived implementation (oracle/dense_gat.py: dense masked-softmax attention from the GAT
paper's equations in numpy, gradients by numerical differentiation) reproduced them: the script refuses to write the
file otherwise.  Once committed, the file pins BOTH restatements and the HIP path: a silent regression of the oracle no
longer moves checker and checked together.  To cross-check the same case against real PyG on a machine that has it:
tools/dump_for_pyg.py.

    python tests/golden/make_gat_golden.py
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import dense_gat                                                           # noqa: E402
from oracle.gat_oracle import HeteroGNNOracle, weighted_mse                            # noqa: E402
from oracle.pyg_semantics import FullNeighborSampler, add_self_loops_hetero, to_undirected_hetero   # noqa: E402
from tests.golden import gat_case as gc                                                # noqa: E402

GRAD_STRIDE = 37            # [128,128] gradient matrices are stored as every 37th element + their sum and norm


def transformed_edges():
    e0 = OrderedDict((k, torch.from_numpy(v)) for k, v in gc.original_edges().items())
    return add_self_loops_hetero(to_undirected_hetero(e0, dict(gc.NODES)), dict(gc.NODES))


def build_oracle(edge_types, dtype=torch.float64):
    o = HeteroGNNOracle(edge_types, gc.HIDDEN, 1, gc.NUM_LAYERS, 'GAT', 'sum', gc.DIMS['SNP'], gc.DIMS['Gene'],
                        gc.DIMS['GO'], 1, dtype=dtype)
    sd = OrderedDict((k, torch.from_numpy(v).to(dtype)) for k, v in gc.parameters(edge_types).items())
    o.load_state_dict(sd, strict=True)
    return o


def layerwise(oracle, x, ei, batch_size):
    """The loop of kgwas/model.py:56-75 with every intermediate kept."""
    from oracle.gat_oracle import GO_TYPES
    h = dict(x)
    h['SNP'] = oracle.snp_feat_mlp(h['SNP'])
    h['Gene'] = oracle.gene_feat_mlp(h['Gene'])
    for t in GO_TYPES:
        h[t] = oracle.go_feat_mlp(h[t])
    xs, alphas = [dict(h)], []
    for conv in oracle.convs:
        h, att = conv(h, ei, return_attention_weights=True)
        h = {k: v.relu() for k, v in h.items()}
        xs.append(dict(h)); alphas.append(att)
    pred = torch.relu(oracle.lin(h['SNP']))[:batch_size]
    return pred, xs, alphas


def sorted_pairs(n_id, ei, et):
    s, _, d = et
    src = n_id[s][ei[0]].numpy(); dst = n_id[d][ei[1]].numpy()
    order = np.lexsort((src, dst))
    return np.stack([src[order], dst[order]], axis=1), order


def main():
    und = transformed_edges()
    edge_types = list(und.keys())
    feats = gc.features()
    y_all, w_all = gc.labels_and_weights()
    out = OrderedDict()
    out['edge_type_names'] = np.array(['|'.join(et) for et in edge_types])
    for k, et in enumerate(edge_types):
        out[f'und_edges_{k}'] = und[et].numpy()

    # --- sampler (kgwas/kgwas.py:99-113 with num_neighbors=[-1,-1]) ---------------------------------------------
    smp = FullNeighborSampler(und, dict(gc.NODES), gc.NUM_LAYERS)
    n_id, ei, hops = smp.sample('SNP', gc.SEEDS)
    for t in gc.NODES:
        out[f'nid_{t}'] = n_id[t].numpy()                 # PyG first-seen order (seeds first)
        out[f'hop_{t}'] = hops[t].numpy()
    pairs = {}
    for k, et in enumerate(edge_types):
        pairs[et], _ = sorted_pairs(n_id, ei[et], et)
        out[f'pairs_{k}'] = pairs[et]

    # --- forward / backward in float64 ------------------------------------------------------------------------
    oracle = build_oracle(edge_types)
    x = {t: torch.from_numpy(feats[t])[n_id[t]].double() for t in gc.NODES}
    pred, xs, alphas = layerwise(oracle, x, ei, gc.BATCH)
    assert torch.equal(pred, oracle(x, ei, gc.BATCH))
    seeds_t = torch.from_numpy(gc.SEEDS)
    yb = torch.from_numpy(y_all)[seeds_t].double()
    wb = torch.from_numpy(w_all)[seeds_t]
    loss = weighted_mse(pred, yb, wb)
    loss.backward()
    grads = {n: p.grad for n, p in oracle.named_parameters()}
    # product-style names: 'convs.0.convs.<key>.lin_src.weight' -- identical to the oracle's parameter names

    # --- second derivation: dense masked softmax + numerical differentiation ------------------------------------
    P = {k: v.astype(np.float64) for k, v in gc.parameters(edge_types).items()}
    n_local = {t: int(n_id[t].numel()) for t in gc.NODES}
    x_np = {t: x[t].numpy() for t in gc.NODES}
    ei_np = OrderedDict((et, ei[et].numpy()) for et in edge_types)
    col = {}
    pred_d = dense_gat.forward(P, x_np, ei_np, n_local, gc.NUM_LAYERS, gc.BATCH, collect=col)
    err = float(np.abs(pred_d - pred.detach().numpy()).max())
    assert err < 1e-11, f'dense vs scatter formulation: prediction differs by {err}'
    for l in range(gc.NUM_LAYERS + 1):
        for t in gc.NODES:
            e = float(np.abs(col[f'x{l}'][t] - xs[l][t].detach().numpy()).max())
            assert e < 1e-10, (l, t, e)
    for l in range(gc.NUM_LAYERS):
        for et in edge_types:
            if ei[et].shape[1]:
                e = float(np.abs(col[f'alpha{l + 1}'][et] - alphas[l][et].detach().numpy().reshape(-1)).max())
                assert e < 1e-12, (l, et, e)
    args = (x_np, ei_np, n_local, gc.NUM_LAYERS, gc.BATCH, yb.numpy(), wb.numpy())
    assert abs(dense_gat.loss(P, *args) - float(loss)) < 1e-12
    dd = []
    for trial in range(4):
        direction = {k: (2.0 * gc.hash01(v.size, 5000 + 97 * trial + i) - 1.0).reshape(v.shape)
                     for i, (k, v) in enumerate(P.items())}
        num = dense_gat.directional_derivative(P, direction, 1e-6, *args)
        ana = sum(float((grads[k].numpy() * direction[k]).sum()) for k in P if grads[k] is not None)
        assert abs(num - ana) <= 1e-6 * max(1.0, abs(ana)), (trial, num, ana)
        dd.append((num, ana))
    print('dense masked-softmax derivation agrees: pred err %.1e, directional derivatives %s' % (err, dd))

    # --- what the file holds ----------------------------------------------------------------------------------
    out['pred'] = pred.detach().numpy().reshape(-1)
    out['h_seed'] = xs[-1]['SNP'][:gc.BATCH].detach().numpy()
    out['loss'] = np.float64(loss.item())
    for l in range(gc.NUM_LAYERS + 1):
        for t in gc.NODES:
            order = np.argsort(n_id[t].numpy(), kind='stable')          # rows by ascending global id
            out[f'x{l}_{t}'] = xs[l][t].detach().numpy()[order]
    for l in range(gc.NUM_LAYERS):
        for k, et in enumerate(edge_types):
            _, order = sorted_pairs(n_id, ei[et], et)
            out[f'alpha{l + 1}_{k}'] = alphas[l][et].detach().numpy().reshape(-1)[order]
    none = []
    for n, g in grads.items():
        if g is None:
            none.append(n)
            continue
        g = g.numpy()
        if g.size >= 128 * 128:
            out[f'gs_{n}'] = g.reshape(-1)[::GRAD_STRIDE].copy()
            out[f'gn_{n}'] = np.array([g.sum(), np.sqrt((g ** 2).sum())])
        else:
            out[f'g_{n}'] = g
    out['grad_none'] = np.array(none)
    out['grad_stride'] = np.int64(GRAD_STRIDE)

    # --- minibatch == full graph for the seeds (survey fact 6) ---------------------------------------------------
    with torch.no_grad():
        full = oracle({t: torch.from_numpy(feats[t]).double() for t in gc.NODES}, und, gc.NODES['SNP']).reshape(-1)
    assert float((full[seeds_t] - pred.detach().reshape(-1)).abs().max()) < 1e-12
    out['pred_full_graph'] = full.numpy()

    # float32 twin of the same computation (what the reference would produce on CPU in its own precision)
    o32 = build_oracle(edge_types, torch.float32)
    with torch.no_grad():
        p32 = o32({t: v.float() for t, v in x.items()}, ei, gc.BATCH)
        out['pred_fp32'] = p32.numpy().reshape(-1)
        out['loss_fp32'] = np.float64(weighted_mse(p32, yb.float(), wb).item())
    path = os.path.join(HERE, 'gat_small.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.2f MB' % (os.path.getsize(path) / 1e6), 'loss', float(loss), 'pred[:4]', out['pred'][:4],
          'sampled nodes', n_local, 'edges', int(sum(v.shape[1] for v in ei.values())))


if __name__ == '__main__':
    main()
