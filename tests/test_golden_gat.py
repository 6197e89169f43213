"""CPU: the committed forward / backward / sampler vectors of the tiny GAT case (tests/golden/gat_small.npz, SURVEY.md 8c
items 1-3) against BOTH restatements of the reference -- the op-for-op scatter formulation (oracle/gat_oracle.py) and the
independently derived dense masked-softmax one (oracle/dense_gat.py, numerical gradients) -- and against the product's
host-side graph transforms.  The HIP path is held to the same file in tests/test_gpu_golden.py."""
from collections import OrderedDict

import numpy as np
import torch

from tests.golden import gat_case as gc
from tests.golden_io import build_oracle, golden, sampled_inputs, transformed_edges


def test_graph_transforms_match_the_committed_edge_lists():
    """ToUndirected + AddSelfLoops (kgwas_data.py:271-272): the oracle's restatement and the product's numpy build."""
    from kgwas_amd.graph import add_self_loops, to_undirected
    G = golden()
    names = [tuple(s.split('|')) for s in G['edge_type_names']]
    und_o = transformed_edges()
    und_p = add_self_loops(to_undirected(gc.original_edges(), dict(gc.NODES)), dict(gc.NODES))
    assert list(und_o.keys()) == names == list(und_p.keys())
    assert len(names) == 13 and sum(1 for et in names if et[1].startswith('rev_')) == 6
    for k, et in enumerate(names):
        assert np.array_equal(und_o[et].numpy(), G[f'und_edges_{k}']), et
        assert np.array_equal(np.asarray(und_p[et]), G[f'und_edges_{k}']), et
    g2g = G[f'und_edges_{names.index(("Gene", "G2G", "Gene"))}']
    assert (g2g[0] == g2g[1]).sum() == 10 + 2          # 10 appended loops + the two that were already there


def test_oracle_sampler_matches_the_committed_sets():
    from oracle.pyg_semantics import FullNeighborSampler
    from oracle.sampler_np import FullNeighborSamplerNP
    G = golden()
    und = transformed_edges()
    n_id, ei, hops = FullNeighborSampler(und, dict(gc.NODES), gc.NUM_LAYERS).sample('SNP', gc.SEEDS)
    n_id2, ei2 = FullNeighborSamplerNP(und, dict(gc.NODES), gc.NUM_LAYERS).sample('SNP', gc.SEEDS)
    for t in gc.NODES:
        assert np.array_equal(n_id[t].numpy(), G[f'nid_{t}']) and np.array_equal(hops[t].numpy(), G[f'hop_{t}'])
        assert np.array_equal(np.sort(np.asarray(n_id2[t])), np.sort(G[f'nid_{t}']))
    assert np.array_equal(G['nid_SNP'][:gc.BATCH], gc.SEEDS)                     # seeds first, in seed order
    for k, et in enumerate(und):
        s, _, d = et
        for nid, e in ((n_id, ei[et]), (n_id2, ei2[et])):
            src = np.asarray(nid[s])[np.asarray(e[0])]; dst = np.asarray(nid[d])[np.asarray(e[1])]
            o = np.lexsort((src, dst))
            assert np.array_equal(np.stack([src[o], dst[o]], 1).reshape(-1, 2), G[f'pairs_{k}'].reshape(-1, 2)), et


def _check_grads(G, grads, rtol, atol_scale):
    # (absolute floor 1e-6 * atol_scale: a gradient that is mathematically zero -- e.g. d att_dst of a relation whose segments all
    # hold one edge -- is 1e-19-sized float64 round-off whose value depends on the host's BLAS summation order)
    none = set(G['grad_none'].tolist())
    stride = int(G['grad_stride'])
    n = 0
    for name, g in grads.items():
        if name in none:
            assert g is None or float(np.abs(g).max()) == 0.0, name
            continue
        assert g is not None, name
        g = np.asarray(g, dtype=np.float64)
        if f'g_{name}' in G.files:
            ref = G[f'g_{name}']
            assert np.allclose(g, ref, rtol=rtol, atol=atol_scale * max(np.abs(ref).max(), 1e-6)), name
        else:
            ref = G[f'gs_{name}']
            assert np.allclose(g.reshape(-1)[::stride], ref, rtol=rtol, atol=atol_scale * max(np.abs(ref).max(), 1e-6)), name
            s, nrm = G[f'gn_{name}']
            assert abs(np.sqrt((g ** 2).sum()) - nrm) <= rtol * nrm + 1e-3 * atol_scale, name        # (norm of a matrix of round-off noise)
        n += 1
    assert n > 60
    return n


def test_scatter_oracle_reproduces_the_committed_vectors():
    G = golden()
    x, ei, n_id, yb, wb = sampled_inputs()
    oracle = build_oracle(list(ei.keys()))
    from tests.golden.make_gat_golden import layerwise, sorted_pairs
    from oracle.gat_oracle import weighted_mse
    pred, xs, alphas = layerwise(oracle, x, ei, gc.BATCH)
    loss = weighted_mse(pred, yb, wb)
    loss.backward()
    assert np.allclose(pred.detach().numpy().reshape(-1), G['pred'], rtol=0, atol=1e-13)
    assert abs(float(loss) - float(G['loss'])) < 1e-12
    for l in range(gc.NUM_LAYERS + 1):
        for t in gc.NODES:
            order = np.argsort(n_id[t].numpy(), kind='stable')
            assert np.allclose(xs[l][t].detach().numpy()[order], G[f'x{l}_{t}'], rtol=0, atol=1e-12), (l, t)
    for l in range(gc.NUM_LAYERS):
        for k, et in enumerate(ei):
            _, order = sorted_pairs(n_id, ei[et], et)
            a = alphas[l][et].detach().numpy().reshape(-1)[order]
            assert np.allclose(a, G[f'alpha{l + 1}_{k}'], rtol=0, atol=1e-13), (l, et)
    _check_grads(G, {n: (p.grad.numpy() if p.grad is not None else None) for n, p in oracle.named_parameters()}, 1e-10, 1e-12)
    # float32 twin
    o32 = build_oracle(list(ei.keys()), torch.float32)
    with torch.no_grad():
        p32 = o32({t: v.float() for t, v in x.items()}, ei, gc.BATCH).numpy().reshape(-1)
    assert np.allclose(p32, G['pred_fp32'], rtol=1e-5, atol=1e-6) and np.allclose(p32, G['pred'], rtol=1e-4, atol=1e-5)


def test_dense_masked_softmax_derivation_reproduces_the_committed_vectors():
    """Independent of the scatter formulation: dense attention from the GAT paper's equations, gradients by central
    differences along random directions against the COMMITTED gradient samples' parent tensors (via the oracle)."""
    from oracle import dense_gat
    G = golden()
    x, ei, n_id, yb, wb = sampled_inputs()
    P = {k: v.astype(np.float64) for k, v in gc.parameters(list(ei.keys())).items()}
    n_local = {t: int(n_id[t].numel()) for t in gc.NODES}
    x_np = {t: x[t].numpy() for t in gc.NODES}
    ei_np = OrderedDict((et, ei[et].numpy()) for et in ei)
    col = {}
    pred = dense_gat.forward(P, x_np, ei_np, n_local, gc.NUM_LAYERS, gc.BATCH, collect=col).reshape(-1)
    assert np.allclose(pred, G['pred'], rtol=0, atol=1e-12)
    assert np.allclose(col['x2']['SNP'][:gc.BATCH], G['h_seed'], rtol=0, atol=1e-11)
    args = (x_np, ei_np, n_local, gc.NUM_LAYERS, gc.BATCH, yb.numpy(), wb.numpy())
    assert abs(dense_gat.loss(P, *args) - float(G['loss'])) < 1e-12
    for l in range(gc.NUM_LAYERS):
        for k, et in enumerate(ei):
            s, _, d = et
            e = ei_np[et]
            src = n_id[s].numpy()[e[0]]; dst = n_id[d].numpy()[e[1]]
            order = np.lexsort((src, dst))
            assert np.allclose(col[f'alpha{l + 1}'][et][order], G[f'alpha{l + 1}_{k}'], rtol=0, atol=1e-13), (l, et)
    # numerical derivative along a direction that only moves the SMALL tensors (stored whole in the file)
    small = [k for k in P if f'g_{k}' in G.files]
    direction = {k: (2.0 * gc.hash01(P[k].size, 9000 + i) - 1.0).reshape(P[k].shape) for i, k in enumerate(small)}
    num = dense_gat.directional_derivative(P, direction, 1e-6, *args)
    ana = sum(float((G[f'g_{k}'] * direction[k]).sum()) for k in small)
    assert abs(num - ana) <= 2e-6 * max(1.0, abs(ana)), (num, ana)


def test_minibatch_prediction_equals_full_graph_prediction():
    G = golden()
    assert np.allclose(G['pred_full_graph'][gc.SEEDS], G['pred'], rtol=0, atol=1e-12)
    und = transformed_edges()
    oracle = build_oracle(list(und.keys()))
    feats = gc.features()
    with torch.no_grad():
        full = oracle({t: torch.from_numpy(feats[t]).double() for t in gc.NODES}, und, gc.NODES['SNP']).reshape(-1)
    assert np.allclose(full.numpy(), G['pred_full_graph'], rtol=0, atol=1e-12)
