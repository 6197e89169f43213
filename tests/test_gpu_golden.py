"""-m gpu: the HIP path against the COMMITTED vectors of the tiny GAT case (tests/golden/gat_small.npz; inputs regenerated
bit-exactly by tests/golden/gat_case.py) -- sampler node / edge sets (exact), attention, prediction, loss, every
parameter gradient (fp32 kernels vs float64 vectors: |a-b| <= 1e-5 + 1e-4|b| + 1e-5 max|b|).  Unlike the live
oracle comparisons of tests/test_gpu_model.py, the expected values here cannot drift with the oracle."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from tests.golden import gat_case as gc
from tests.golden_io import case_graph, golden
from tests.helpers import assert_close, grads_by_name

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5


@pytest.fixture(scope='module')
def case():
    from kgwas_amd.model import HeteroGNN
    from kgwas_amd.sampler import NeighborLoader
    data, w_all = case_graph()
    model = HeteroGNN(data, gc.HIDDEN, 1, gc.NUM_LAYERS, 'GAT', 'sum', gc.DIMS['SNP'], gc.DIMS['Gene'], gc.DIMS['GO'], 1).cuda()
    sd = OrderedDict((k, torch.from_numpy(v)) for k, v in gc.parameters(data.edge_types).items())
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    batch = next(iter(NeighborLoader(data, [-1] * gc.NUM_LAYERS, ('SNP', gc.SEEDS), batch_size=gc.BATCH, device='cuda:0')))
    return data, w_all, model, batch


def test_sampler_matches_committed_node_and_edge_sets(case):
    data, _, _, batch = case
    G = golden()
    L = gc.NUM_LAYERS
    for i, t in enumerate(gc.NODES):
        nid = batch.n_id(t).cpu().numpy()
        assert np.array_equal(np.sort(nid), np.sort(G[f'nid_{t}'])), t
        # hop membership: the local order is hop-major, so the hop of a node is given by the meta offsets
        off = [int(batch.meta.node_off[i][k]) for k in range(L + 2)]
        hop_of = {int(g): int(h) for g, h in zip(G[f'nid_{t}'], G[f'hop_{t}'])}
        for k in range(L + 1):
            assert all(hop_of[int(g)] == k for g in nid[off[k]:off[k + 1]]), (t, k)
    assert np.array_equal(batch.n_id('SNP').cpu().numpy()[:gc.BATCH], gc.SEEDS)
    ei = batch.edge_index_dict
    for k, et in enumerate(data.edge_types):
        s, _, d = et
        e = ei[et].cpu().numpy()
        src = batch.n_id(s).cpu().numpy()[e[0]]; dst = batch.n_id(d).cpu().numpy()[e[1]]
        o = np.lexsort((src, dst))
        assert np.array_equal(np.stack([src[o], dst[o]], 1).reshape(-1, 2), G[f'pairs_{k}'].reshape(-1, 2)), et


def test_forward_loss_and_gradients_match_committed_vectors(case):
    data, w_all, model, batch = case
    G = golden()
    model.train()
    for p in model.parameters():
        p.grad = None
    loss, pred = model.forward_loss(batch.x_dict, batch.edge_index_dict, gc.BATCH, batch.n_id('SNP'),
                                    data['SNP'].y.cuda(), w_all.cuda())
    loss.backward()
    assert_close(pred, G['pred'], RTOL, ATOL, 'pred')
    assert abs(float(loss) - float(G['loss'])) <= 1e-5 * float(G['loss'])
    with torch.no_grad():
        p2, h = model(batch.x_dict, batch.edge_index_dict, gc.BATCH, return_h=True)
    assert_close(p2.reshape(-1), G['pred'], RTOL, ATOL, 'pred (forward)')
    assert_close(h, G['h_seed'], RTOL, ATOL, 'h of the seeds')
    # identical ranking of the seeds on integer indexing (north_star) wherever the committed values are separated
    ref = G['pred']
    order = np.argsort(-ref, kind='stable')
    gaps = np.abs(np.diff(ref[order]))
    got = np.argsort(-pred.detach().cpu().numpy().astype(np.float64), kind='stable')
    if gaps.min() > 1e-5:
        assert np.array_equal(got, order)
    none = set(G['grad_none'].tolist())
    stride = int(G['grad_stride'])
    n = 0
    for name, g in grads_by_name(model).items():
        if name in none:
            assert g is None or float(g.abs().max()) == 0.0, name
            continue
        assert g is not None, name
        if f'g_{name}' in G.files:
            ref_g = G[f'g_{name}']
            assert_close(g.reshape(ref_g.shape), ref_g, RTOL, max(ATOL, 1e-4 * float(np.abs(ref_g).max())), f'grad {name}')
        else:
            ref_g = G[f'gs_{name}']
            assert_close(g.reshape(-1)[::stride], ref_g, RTOL, max(ATOL, 1e-4 * float(np.abs(ref_g).max())), f'grad {name}')
            nrm = float(G[f'gn_{name}'][1])
            assert abs(float(g.double().norm()) - nrm) <= 1e-4 * nrm + 1e-7, name
        n += 1
    assert n > 40


def test_attention_matches_committed_vectors(case):
    """Per-edge softmax attention of the hot path (live relations, pruned rows) and the reference-shaped per-layer means
    over ALL relations and edges (kgwas/model.py:65-72)."""
    data, _, model, batch = case
    G = golden()
    L = gc.NUM_LAYERS
    att = model.hot_path_attention(batch)
    m = batch.meta
    seg_ptr = batch.buf.seg_ptr.cpu().numpy()
    col = batch.buf.col_local.cpu().numpy()
    sc = batch.dg.schema
    checked = 0
    for l in range(1, L + 1):
        alpha = att[l - 1].cpu().numpy()
        live = set(model.live_rel[l])
        for hop in range(L - l + 1):                       # layer l aggregates destination hops 0 .. L - l
            for r, et in enumerate(sc.edge_types):
                if r not in live:
                    continue
                a, b = int(m.seg_off[hop][r]), int(m.seg_off[hop][r + 1])
                if b <= a:
                    continue
                s, d = et[0], et[2]
                nid_s = batch.n_id(s).cpu().numpy(); nid_d = batch.n_id(d).cpu().numpy()
                ref_pairs = G[f'pairs_{r}'].reshape(-1, 2); ref_alpha = G[f'alpha{l}_{r}']
                lut = {}
                for (sg, dgl), v in zip(map(tuple, ref_pairs), ref_alpha):
                    lut[(int(sg), int(dgl))] = float(v)           # duplicate edges carry the same attention
                d_i = sc.type_id[d]
                for sg_i in range(a, b):
                    row = int(m.node_off[d_i][hop]) + (sg_i - a)
                    for e in range(int(seg_ptr[sg_i]), int(seg_ptr[sg_i + 1])):
                        want = lut[(int(nid_s[col[e]]), int(nid_d[row]))]
                        assert abs(float(alpha[e]) - want) <= 1e-5 + 1e-4 * want, (l, et, e)
                        checked += 1
    assert checked > 300
    with torch.no_grad():
        _, means = model(batch.x_dict, batch.edge_index_dict, gc.BATCH, return_attention_weights=True)
    for l in range(L):
        ref = np.concatenate([G[f'alpha{l + 1}_{k}'] for k in range(len(data.edge_types))]).mean()
        assert abs(float(means[l]) - ref) <= 1e-5 * ref + 1e-7, (l, float(means[l]), ref)


def test_full_graph_forward_matches_committed_vector(case):
    data, _, model, batch = case
    G = golden()
    with torch.no_grad():
        xf = {t: data[t].x.cuda() for t in data.node_types}
        eif = {et: data[et].edge_index.cuda() for et in data.edge_types}
        full = model(xf, eif, gc.NODES['SNP']).reshape(-1)
        mini = model(batch.x_dict, batch.edge_index_dict, gc.BATCH).reshape(-1)
    assert_close(full, G['pred_full_graph'], RTOL, ATOL, 'full-graph prediction')
    assert_close(mini, full[torch.as_tensor(gc.SEEDS)], RTOL, ATOL, 'minibatch vs full graph')
