"""-m gpu: HeteroGNN on the fused path vs the reference restatement (oracle/gat_oracle.py) with the same
weights on the same sampled subgraph; training-step parity; API round trips.

The oracle runs the reference's UNPRUNED computation (both layers on every sampled node and edge,
kgwas/model.py:64-86); the product computes only what the seeds depend on -- outputs and parameter
gradients must still agree."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle.gat_oracle import weighted_mse
from tests.helpers import assert_close, batch_cpu, grads_by_name, oracle_from_product, params_by_name

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5


def _model(data, dims, seed=0, L=2, backbone='GAT', aggr='sum', **kw):
    from kgwas_amd.model import HeteroGNN
    torch.manual_seed(seed)
    m = HeteroGNN(data, 128, 1, L, backbone, aggr, dims[0], dims[1], dims[2], 1, **kw).cuda()
    # biases start at zero in the reference (conv.py:120); randomise so their path is exercised
    with torch.no_grad():
        for pack in list(m.live_packs) + list(m.dead_packs):
            pack.bias.normal_(0, 0.1)
    return m


@pytest.fixture(params=['default-routes', 'own-kernels-only', 'library-allowed'])
def gemm_routing(request):
    """'default-routes': what a user gets -- this package's kernel whenever one can take the shape (round 4), the library only
    for shapes none takes.  'own-kernels-only': ops.LIBRARY_GEMM.strict -- in addition, a library GEMM is an error.
    'library-allowed': the opt-in of KGW_ALLOW_LIBRARY=1 -- problems of a few hundred rows go to hipBLASLt (the default of
    rounds 1-3).  All three must match the restatement."""
    from kgwas_amd import ops
    was = ops.LIBRARY_GEMM.strict, ops.LIBRARY_GEMM.allow_library
    ops.LIBRARY_GEMM.strict = request.param == 'own-kernels-only'
    ops.LIBRARY_GEMM.allow_library = request.param == 'library-allowed'
    ops.LIBRARY_GEMM.reset()
    yield request.param
    own, calls = ops.LIBRARY_GEMM.strict, ops.LIBRARY_GEMM.calls
    ops.LIBRARY_GEMM.strict, ops.LIBRARY_GEMM.allow_library = was
    assert not own or calls == 0
    if request.param == 'default-routes':
        assert calls == 0, 'the small oracle cases have no product that needs the library'


def _loader(data, ids, bs, L=2):
    from kgwas_amd.sampler import NeighborLoader
    return NeighborLoader(data, [-1] * L, ('SNP', ids), batch_size=bs, device='cuda:0')


@pytest.mark.parametrize('which', ['small', 'edge'])
@pytest.mark.parametrize('L', [1, 2, 3])
def test_forward_backward_matches_reference_restatement(small_kg, edge_case_graph, which, L, gemm_routing):
    if which == 'small':
        data, dims = small_kg.data, (small_kg.snp_init_dim_size, small_kg.gene_init_dim_size, small_kg.go_init_dim_size)
    else:
        data, d = edge_case_graph
        dims = (d['SNP'], d['Gene'], 16)
    model = _model(data, dims, L=L)
    ids = np.random.default_rng(L).choice(data['SNP'].x.shape[0], size=40, replace=False)
    batch = next(iter(_loader(data, ids, 40, L)))
    bs = batch['SNP'].batch_size
    out = model(batch.x_dict, batch.edge_index_dict, bs)
    assert out.shape == (40, 1)
    y = torch.rand(40, dtype=torch.float64)
    w = torch.rand(40, dtype=torch.float64) + 0.5
    loss = weighted_mse(out, y.cuda(), w.cuda())
    loss.backward()

    oracle = oracle_from_product(model)
    x, ei = batch_cpu(batch)
    out_o = oracle(x, ei, bs)
    loss_o = weighted_mse(out_o, y, w)
    loss_o.backward()

    assert_close(out, out_o.detach(), RTOL, ATOL, 'pred')
    assert_close(loss.detach(), loss_o.detach(), RTOL, ATOL, 'loss')
    go = grads_by_name(oracle)
    n_live = 0
    for name, g in grads_by_name(model).items():
        ref = go[name]
        if g is None:
            # structurally dead parameter: the reference gives it grad None (or exactly zero)
            assert ref is None or float(ref.abs().max()) == 0.0, f'{name}: product has no grad, oracle has'
            continue
        n_live += 1
        assert ref is not None, name
        scale = float(ref.abs().max())
        assert_close(g, ref, RTOL, max(ATOL, 1e-4 * scale), f'grad {name}')
    assert n_live > 10
    # identical top-k SNP ranking on fixed weights (north_star)
    k = 10
    assert torch.equal(torch.topk(out.flatten().cpu().double(), k).indices, torch.topk(out_o.flatten(), k).indices) \
        or float((out.flatten().cpu().double() - out_o.flatten()).abs().max()) < 1e-6


@pytest.mark.parametrize('hc', [64, 20])
def test_hidden_width_below_128_matches_the_restatement_at_that_width(small_kg, hc):
    """gnn_hidden_dim (kgwas/kgwas.py:52) below the kernels' 128: the model runs zero-padded on the 128-wide kernels and must
    equal the reference restatement BUILT AT THAT WIDTH -- prediction, loss, every gradient (exposed at the model's own width) --
    and stay inside its block through Adam steps with weight decay (the padding receives exactly zero gradient)."""
    from kgwas_amd.kgwas import KGWAS
    run = KGWAS(small_kg, device='cuda:0', seed=13)
    run.initialize_model(gnn_hidden_dim=hc)
    model = run.model
    assert model.hidden_logical == hc and run.config['gnn_hidden_dim'] == hc
    with torch.no_grad():
        for pack in list(model.live_packs) + list(model.dead_packs):
            pack.bias[:, :hc].normal_(0, 0.1)
    oracle = oracle_from_product(model)
    assert oracle.lin.in_features == hc
    ids = np.asarray(small_kg.train_input_nodes[1][:3 * 64])
    ld_w = run._ld_weight_vector()
    y_all = small_kg.data['SNP'].y.double()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4)
    model.train()
    for k, batch in enumerate(_loader(small_kg.data, ids, 64)):
        if k == 0:
            out = model(batch.x_dict, batch.edge_index_dict, 64)
            n_id = batch.n_id('SNP')[:64].long()
            loss = weighted_mse(out, y_all[n_id.cpu()].cuda(), ld_w[n_id])
            loss.backward()
            x, ei = batch_cpu(batch)
            out_o = oracle(x, ei, 64)
            loss_o = weighted_mse(out_o, y_all[n_id.cpu()], ld_w[n_id].cpu())
            loss_o.backward()
            assert_close(out, out_o.detach(), RTOL, ATOL, 'pred')
            go = grads_by_name(oracle)
            n_live = 0
            for name, g in grads_by_name(model).items():
                if g is None:
                    continue
                assert g.shape == go[name].shape, name
                assert_close(g, go[name], RTOL, max(ATOL, 1e-4 * float(go[name].abs().max())), f'grad {name}')
                n_live += 1
            assert n_live > 10
            model.zero_grad(set_to_none=True)
        run.train_step(batch, opt, ld_w)
    for p in model.parameters():                      # three Adam steps later the padding is still exactly zero
        if p.dim() == 3 and p.shape[-1] == 128:
            assert float(p[:, hc:, :].abs().sum()) == 0.0 and float(p[:, :, hc:].abs().sum()) == 0.0
        elif p.dim() == 2 and p.shape[-1] == 128 and p.shape[0] != 128:
            assert float(p[:, hc:].abs().sum()) == 0.0
    for mlp in (model.snp_feat_mlp, model.gene_feat_mlp, model.go_feat_mlp):
        assert float(mlp.FC_hidden.weight[hc:].abs().sum()) == 0.0 and float(mlp.FC_hidden2.weight[hc:].abs().sum()) == 0.0
        assert float(mlp.FC_hidden2.weight[:, hc:].abs().sum()) == 0.0 and float(mlp.FC_output.bias[hc:].abs().sum()) == 0.0
    p, h = model(batch.x_dict, batch.edge_index_dict, 64, return_h=True)
    assert h.shape == (64, hc)


def test_minibatch_equals_full_graph(edge_case_graph):
    """the seeds' minibatch output equals the full-graph 2-layer output (generic
    (x_dict, edge_index_dict) entry of HeteroGNN.forward = every row, every layer)."""
    data, d = edge_case_graph
    model = _model(data, (d['SNP'], d['Gene'], 16), seed=4)
    ids = np.random.default_rng(1).choice(3000, size=64, replace=False)
    batch = next(iter(_loader(data, ids, 64)))
    with torch.no_grad():
        mini = model(batch.x_dict, batch.edge_index_dict, 64)
        xf = {t: data[t].x.cuda() for t in data.node_types}
        eif = {et: data[et].edge_index.cuda() for et in data.edge_types}
        full = model(xf, eif, 3000)
    assert full.shape == (3000, 1)
    assert_close(mini, full[torch.as_tensor(ids)], RTOL, ATOL, 'minibatch vs full graph')
    # and the full-graph path agrees with the oracle on the full graph
    oracle = oracle_from_product(model)
    with torch.no_grad():
        full_o = oracle({t: data[t].x.double() for t in data.node_types}, data.edge_index_dict, 3000)
    assert_close(full, full_o, RTOL, ATOL, 'full graph vs oracle')


def test_return_h_no_relu_attention(edge_case_graph):
    data, d = edge_case_graph
    model = _model(data, (d['SNP'], d['Gene'], 16), seed=5)
    oracle = oracle_from_product(model)
    ids = np.arange(100, 132)
    batch = next(iter(_loader(data, ids, 32)))
    x, ei = batch_cpu(batch)
    with torch.no_grad():
        p, h = model(batch.x_dict, batch.edge_index_dict, 32, return_h=True)
        po, ho = oracle(x, ei, 32, return_h=True)
        assert_close(p, po, RTOL, ATOL, 'pred(return_h)')
        assert_close(h, ho, RTOL, ATOL, 'h')
        model.no_relu = oracle.no_relu = True
        assert_close(model(batch.x_dict, batch.edge_index_dict, 32), oracle(x, ei, 32), RTOL, ATOL, 'no_relu')
        model.no_relu = oracle.no_relu = False
        # model.py:65-72: the mean attention over ALL edge types and ALL edges of the batch, per layer
        p2, att = model(batch.x_dict, batch.edge_index_dict, 32, return_attention_weights=True)
        p2o, atto = oracle(x, ei, 32, return_attention_weights=True)
        assert_close(p2, po, RTOL, ATOL, 'pred(attention)')
        assert len(att) == len(atto) == 2
        for l, (a, ao) in enumerate(zip(att, atto)):
            assert abs(float(a) - float(ao)) <= 1e-5 * abs(float(ao)) + 1e-7, (l, float(a), float(ao))


def test_training_steps_track_the_reference(small_kg):
    """A few Adam steps (lr 1e-4, weight_decay 5e-4 as L2, kgwas.py:116,142-151): parameters of the HIP
    path follow the oracle trained on identical batches / init."""
    from kgwas_amd.kgwas import KGWAS
    run = KGWAS(small_kg, device='cuda:0', seed=7)
    run.initialize_model()
    oracle = oracle_from_product(run.model, dtype=torch.float64)
    opt = torch.optim.Adam(run.model.parameters(), lr=1e-3, weight_decay=5e-4)
    opt_o = torch.optim.Adam(oracle.parameters(), lr=1e-3, weight_decay=5e-4)
    ld_w = run._ld_weight_vector()
    p0 = params_by_name(run.model)
    ids = np.asarray(small_kg.train_input_nodes[1][:4 * 64])
    y_all = small_kg.data['SNP'].y.double()
    run.model.train()
    for batch in _loader(small_kg.data, ids, 64):
        loss = run.train_step(batch, opt, ld_w)
        x, ei = batch_cpu(batch)
        n_id = batch.n_id('SNP')[:64].long().cpu()
        opt_o.zero_grad()
        loss_o = weighted_mse(oracle(x, ei, 64), y_all[n_id], ld_w.cpu()[n_id])
        loss_o.backward()
        opt_o.step()
        assert_close(loss.detach(), loss_o.detach(), 1e-4, 1e-6, 'loss')
    # Adam normalises every coordinate's step to ~lr, so coordinates whose gradient is fp32 noise can differ
    # by O(lr) between an fp32 and an fp64 run; compare the parameter UPDATE in norm, per tensor and overall.
    po = dict(oracle.named_parameters())
    num = den = 0.0
    for n, p in params_by_name(run.model).items():
        d_hip = p - p0[n]
        d_ref = po[n].detach() - p0[n]
        num += float((d_hip - d_ref).pow(2).sum()); den += float(d_ref.pow(2).sum())
        assert float((d_hip - d_ref).abs().max()) <= 2.5 * 1e-3 * 4, n          # never more than ~lr per step
    assert den > 0 and (num / den) ** 0.5 < 2e-2, f'relative update error {(num / den) ** 0.5:.3e}'


def test_checkpoint_roundtrip(small_kg, tmp_path):
    """model.pt / config.pkl as written by kgwas/utils.py:203-207; keys are the reference's."""
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.utils import save_model
    run = KGWAS(small_kg, device='cuda:0', seed=1)
    run.initialize_model()
    save_model(run.model, run.config, str(tmp_path / 'ck'))
    sd = torch.load(str(tmp_path / 'ck' / 'model.pt'), weights_only=False)
    assert 'snp_feat_mlp.FC_hidden.weight' in sd and 'lin.bias' in sd
    assert 'convs.0.convs.SNP__ABC__Gene.lin_src.weight' in sd
    assert 'convs.1.convs.Gene__rev_ABC__SNP.att_dst' in sd
    run2 = KGWAS(small_kg, device='cuda:0', seed=2)
    run2.load_pretrained(str(tmp_path / 'ck'))
    for (n, a), (_, b) in zip(run.model.named_parameters(), run2.model.named_parameters()):
        assert torch.equal(a, b), n


@pytest.mark.parametrize('backbone', ['GAT', 'SAGE'])
def test_end_to_end_train_api(tiny_kg, backbone):
    """KGWAS.train() end to end on a tiny graph: loaders, epochs, best model, inference column."""
    from kgwas_amd.kgwas import KGWAS
    run = KGWAS(tiny_kg, device='cuda:0', seed=3)
    run.initialize_model(gnn_backbone=backbone)
    run.train(batch_size=32, epoch=2, save_best_model=False)     # val set (52 SNPs) must hold one full batch
    assert len(run.train_loader) == len(tiny_kg.train_input_nodes[1]) // 32
    assert 'pred' in run.data.lr_uni.columns and np.isfinite(run.data.lr_uni['pred'].values).all()
    res = run.kgwas_res                                          # kgwas.py:196-212 outputs
    assert {'P_weighted', 'KGWAS_P'} <= set(res.columns)
    assert float(res['KGWAS_P'].min()) >= 0.0 and float(res['KGWAS_P'].max()) <= 1.0
    assert np.isfinite(run.val_metrics['mse'])


def test_validation_pearson_matches_oracle_after_training(small_kg):
    """north_star: val-set Pearson r of the MI355X path within 1e-3 of the reference restatement trained with
    identical init / data / batch order (one epoch of kgwas/kgwas.py:126-164 on the small synthetic KG; the HIP
    side runs the captured-graph step, the oracle the unpruned fp64 reference computation)."""
    from scipy.stats import pearsonr
    from kgwas_amd.graph_step import GraphTrainStep
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.utils import evaluate_minibatch_clean
    from oracle.sampler_np import FullNeighborSamplerNP
    bs = 64
    torch.set_num_threads(min(16, torch.get_num_threads()))      # the CPU oracle degrades badly when oversubscribed
    run = KGWAS(small_kg, device='cuda:0', seed=21)
    run.initialize_model()
    oracle = oracle_from_product(run.model, dtype=torch.float64)
    ids = np.asarray(small_kg.train_input_nodes[1])
    nb = min(len(ids) // bs, 32)
    ids = ids[:nb * bs]
    lr, wd = 1e-3, 5e-4
    gs = GraphTrainStep(run, ('SNP', ids), bs, lr=lr, weight_decay=wd)
    run.model.train()
    for i in range(nb):
        gs.step(i)
    gs.check()
    # oracle: same batches, same order, torch Adam
    g = small_kg.data
    smp = FullNeighborSamplerNP(g.edge_index_dict, g.num_nodes_dict, 2)
    opt = torch.optim.Adam(oracle.parameters(), lr=lr, weight_decay=wd)
    y_all = g['SNP'].y.double()
    w_all = run._ld_weight_vector().cpu()
    for i in range(nb):
        seeds = ids[i * bs:(i + 1) * bs]
        n_id, ei = smp.sample('SNP', seeds)
        x = {t: g[t].x[n_id[t]].double() for t in g.node_types}
        opt.zero_grad()
        weighted_mse(oracle(x, ei, bs), y_all[n_id['SNP'][:bs]], w_all[n_id['SNP'][:bs]]).backward()
        opt.step()
    # validation predictions of both on the val split
    from kgwas_amd.sampler import NeighborLoader
    val_ids = np.asarray(small_kg.val_input_nodes[1])[:192]
    res = evaluate_minibatch_clean(NeighborLoader(g, [-1, -1], ('SNP', val_ids), batch_size=bs, device='cuda:0'),
                                   run.model, 'cuda:0')
    preds_o = []
    with torch.no_grad():
        for i in range(0, len(val_ids), bs):
            seeds = val_ids[i:i + bs]
            n_id, ei = smp.sample('SNP', seeds)
            x = {t: g[t].x[n_id[t]].double() for t in g.node_types}
            preds_o.append(oracle(x, ei, len(seeds)).reshape(-1))
    pred_o = torch.cat(preds_o).numpy()
    truth = y_all[torch.as_tensor(val_ids)].numpy()
    assert np.allclose(res['truth'], truth.astype(np.float32))
    r_hip = pearsonr(res['pred'], res['truth'])[0]
    r_ref = pearsonr(pred_o, truth)[0]
    assert np.std(pred_o) > 0 and np.isfinite(r_hip)
    assert abs(r_hip - r_ref) < 1e-3, (r_hip, r_ref)
    rel = float(np.linalg.norm(res['pred'] - pred_o) / max(np.linalg.norm(pred_o), 1e-12))
    assert rel < 2e-2, f'relative L2 difference of the validation predictions {rel:.3e} (r_hip {r_hip:.5f}, r_ref {r_ref:.5f})'


def test_network_weight_export_matches_reference_procedure(small_kg):
    """Row f-2: KGWAS.get_network_weight() (kgwas/utils.py:437-494) vs the same procedure on the fp64 oracle: MLPs,
    then every HeteroConv with raw (un-normalised) attention returned AND propagated, no ReLU between layers."""
    from kgwas_amd.kgwas import KGWAS
    from oracle.gat_oracle import GO_TYPES
    run = KGWAS(small_kg, device='cuda:0', seed=5)
    run.initialize_model()
    with torch.no_grad():                       # make bias / dead relations non-trivial
        for p in run.model.parameters():
            if p.dim() == 2 and p.shape[-1] == 128 and float(p.abs().sum()) == 0.0:
                p.uniform_(-0.1, 0.1)
    run.best_model = run.model
    df = run.get_network_weight()
    assert list(df.columns) == ['h_idx', 't_idx', 'weight', 'h_type', 'rel_type', 't_type', 'layer']
    assert set(df.layer.unique()) == {'l1', 'l2'}

    g = small_kg.data
    o = oracle_from_product(run.model)
    with torch.no_grad():
        x = {t: g[t].x.double() for t in g.node_types}
        x['SNP'] = o.snp_feat_mlp(x['SNP']); x['Gene'] = o.gene_feat_mlp(x['Gene'])
        for t in GO_TYPES:
            x[t] = o.go_feat_mlp(x[t])
        ei = {et: g[et].edge_index for et in g.edge_types}
        n_checked = 0
        for k, conv in enumerate(o.convs):
            x, att = conv(x, ei, return_attention_weights=True, raw=True)          # utils.py:452-460 (no relu)
            for et, a in att.items():
                sub = df[(df.layer == f'l{k + 1}') & (df.h_type == et[0]) & (df.rel_type == et[1]) & (df.t_type == et[2])]
                e = ei[et].numpy()
                # the reference drops duplicate (h, t) pairs per relation and layer: compare on unique pairs
                key_ref = e[0].astype(np.int64) * (1 << 32) + e[1]
                _, first = np.unique(key_ref, return_index=True)
                ref = dict(zip(key_ref[first].tolist(), a.reshape(-1).numpy()[first].tolist()))
                key_my = sub.h_idx.values.astype(np.int64) * (1 << 32) + sub.t_idx.values.astype(np.int64)
                assert len(key_my) == len(ref) and set(key_my.tolist()) == set(ref)
                mine = torch.tensor(sub.weight.values)
                want = torch.tensor([ref[q] for q in key_my.tolist()])
                assert_close(mine, want, 2e-4, 1e-5, f'raw attention l{k + 1} {et}', rel_to_max=1e-5)
                n_checked += len(ref)
    assert n_checked > 0


def test_gradients_are_bit_reproducible(small_kg):
    """Same batch, same parameters, two independent sample + forward + backward passes: every gradient is bitwise
    identical (no atomics in the arithmetic; the sampler's src-major rows are put in a fixed order)."""
    from kgwas_amd.kgwas import KGWAS
    run = KGWAS(small_kg, device='cuda:0', seed=3)
    run.initialize_model()
    ld_w = run._ld_weight_vector()
    ids = np.asarray(small_kg.train_input_nodes[1][:256])
    grads = []
    for _ in range(2):
        batch = next(iter(_loader(small_kg.data, ids, 256)))
        run.model.zero_grad(set_to_none=True)
        out = run.model(batch.x_dict, batch.edge_index_dict, 256)
        from kgwas_amd import ops
        loss = ops.weighted_mse(out.reshape(-1), batch.n_id('SNP'), batch.dg.y['SNP'], ld_w)
        loss.backward()
        grads.append([p.grad.clone() for p in run.model.parameters() if p.grad is not None])
    assert len(grads[0]) == len(grads[1]) > 0
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_wide_gene_features_take_the_resident_first_layer(tmp_path):
    """Gene features >= 512 wide and most genes in the batch: the first Linear runs on the RESIDENT matrix (no x[n_id]
    copy of 20 KB rows) through the library GEMM with the tuned fixed-shape solution; same values / gradients as the
    oracle on the sliced features."""
    from kgwas_amd.kgwas_data import KGWAS_Data
    kg = KGWAS_Data.from_synthetic(scale=0.002, seed=4, feat_dims={'Gene': 640}, data_path=str(tmp_path))
    data = kg.data
    dims = (kg.snp_init_dim_size, kg.gene_init_dim_size, kg.go_init_dim_size)
    assert dims[1] == 640
    model = _model(data, dims, L=2)
    ids = np.random.default_rng(0).choice(data['SNP'].x.shape[0], size=256, replace=False)
    batch = next(iter(_loader(data, ids, 256)))
    assert 2 * batch.n_nodes['Gene'] > data['Gene'].x.shape[0], 'test graph should put most genes in the batch'
    out = model(batch.x_dict, batch.edge_index_dict, 256)
    y = torch.rand(256, dtype=torch.float64); w = torch.rand(256, dtype=torch.float64) + 0.5
    weighted_mse(out, y.cuda(), w.cuda()).backward()
    oracle = oracle_from_product(model)
    x, ei = batch_cpu(batch)
    out_o = oracle(x, ei, 256)
    weighted_mse(out_o, y, w).backward()
    assert_close(out, out_o.detach(), RTOL, ATOL, 'pred')
    go = grads_by_name(oracle)
    for name, g in grads_by_name(model).items():
        if 'gene_feat_mlp' in name:
            assert_close(g, go[name], RTOL, max(ATOL, 1e-4 * float(go[name].abs().max())), f'grad {name}')


@pytest.mark.parametrize('no_relu', [False, True])
def test_fused_readout_loss_equals_forward_plus_loss(small_kg, no_relu):
    """HeteroGNN.forward_loss (read-out Linear + ReLU + LD-weighted MSE as one node) == forward() followed by the
    reference's loss expression (kgwas/kgwas.py:137-145), values and every gradient."""
    data = small_kg.data
    dims = (small_kg.snp_init_dim_size, small_kg.gene_init_dim_size, small_kg.go_init_dim_size)
    model = _model(data, dims, L=2, no_relu=no_relu)
    ids = np.random.default_rng(2).choice(data['SNP'].x.shape[0], size=96, replace=False)
    batch = next(iter(_loader(data, ids, 96)))
    g = torch.Generator().manual_seed(0)
    w_all = (torch.rand(data['SNP'].x.shape[0], generator=g, dtype=torch.float64) + 0.5).cuda()
    y_all = batch.dg.y['SNP']
    loss, pred = model.forward_loss(batch.x_dict, batch.edge_index_dict, 96, batch.n_id('SNP'), y_all, w_all)
    loss.backward()
    g1 = {n: t.clone() for n, t in grads_by_name(model).items() if t is not None}
    model.zero_grad(set_to_none=True)
    out = model(batch.x_dict, batch.edge_index_dict, 96)
    n_id = batch.n_id('SNP')[:96].long()
    loss2 = torch.mean(w_all[n_id] * (out.reshape(-1) - y_all[n_id]) ** 2)
    loss2.backward()
    g2 = {n: t for n, t in grads_by_name(model).items() if t is not None}
    assert_close(pred, out.reshape(-1).detach(), 1e-6, 1e-7, 'pred')
    assert abs(float(loss) - float(loss2)) <= 1e-9 + 1e-6 * abs(float(loss2))
    assert set(g1) == set(g2)
    for n in g1:
        assert_close(g1[n], g2[n], 1e-4, 1e-7, 'grad ' + n, rel_to_max=1e-5)
    # the training step's variant (loss gradient known to be 1: forward + backward of the node in two launches): bit-identical
    model.zero_grad(set_to_none=True)
    loss3, pred3 = model.forward_loss(batch.x_dict, batch.edge_index_dict, 96, batch.n_id('SNP'), y_all, w_all, unit_grad=True)
    loss3.backward()
    g3 = {n: t for n, t in grads_by_name(model).items() if t is not None}
    assert torch.equal(loss3, loss) and torch.equal(pred3, pred) and set(g3) == set(g1)
    for n in g1:
        assert torch.equal(g3[n], g1[n]), n
    # ... and a caller that scales the loss anyway (gradient accumulation, loss * k) gets the scaled gradients, not those of k = 1
    from kgwas_amd import ops
    for scale, how in ((4.0, 'scaled'), (1.0, 'resident unit')):
        model.zero_grad(set_to_none=True)
        loss4, _ = model.forward_loss(batch.x_dict, batch.edge_index_dict, 96, batch.n_id('SNP'), y_all, w_all, unit_grad=True)
        if how == 'scaled':
            (loss4 * scale).backward()
        else:
            loss4.backward(gradient=ops.unit_gradient(loss4.device))
        g4 = {n: t for n, t in grads_by_name(model).items() if t is not None}
        for n in g1:
            assert torch.equal(g4[n], scale * g1[n]), (how, n)


@pytest.mark.parametrize('which', ['small', 'edge'])
@pytest.mark.parametrize('L', [1, 2])
def test_sage_backbone_matches_reference_restatement(small_kg, edge_case_graph, which, L, gemm_routing):
    """Row f-4: gnn_backbone='SAGE' (kgwas/model.py:38: SAGEConv((-1,-1), 128) per relation, HeteroConv sum, ReLU) on
    the same kernels -- the neighbour mean is the attention aggregate with zero attention vectors -- against the
    op-for-op oracle (oracle.gat_oracle.SAGEConvOracle): prediction, loss, every parameter gradient, checkpoint keys."""
    if which == 'small':
        data, dims = small_kg.data, (small_kg.snp_init_dim_size, small_kg.gene_init_dim_size, small_kg.go_init_dim_size)
    else:
        data, d = edge_case_graph
        dims = (d['SNP'], d['Gene'], 16)
    model = _model(data, dims, L=L, backbone='SAGE')
    keys = [k for k in model.state_dict() if k.startswith('convs.0.')]
    assert any(k.endswith('lin_l.weight') for k in keys) and any(k.endswith('lin_l.bias') for k in keys) and \
        any(k.endswith('lin_r.weight') for k in keys) and not any('att_src' in k for k in keys)
    ids = np.random.default_rng(L).choice(data['SNP'].x.shape[0], size=40, replace=False)
    batch = next(iter(_loader(data, ids, 40, L)))
    out = model(batch.x_dict, batch.edge_index_dict, 40)
    y = torch.rand(40, dtype=torch.float64); w = torch.rand(40, dtype=torch.float64) + 0.5
    loss = weighted_mse(out, y.cuda(), w.cuda())
    loss.backward()
    oracle = oracle_from_product(model)
    x, ei = batch_cpu(batch)
    out_o = oracle(x, ei, 40)
    loss_o = weighted_mse(out_o, y, w)
    loss_o.backward()
    assert_close(out, out_o.detach(), RTOL, ATOL, 'pred')
    assert_close(loss.detach(), loss_o.detach(), RTOL, ATOL, 'loss')
    go = grads_by_name(oracle)
    n_live = 0
    for name, g in grads_by_name(model).items():
        ref = go[name]
        if g is None:
            assert ref is None or float(ref.abs().max()) == 0.0, f'{name}: product has no grad, oracle has'
            continue
        n_live += 1
        assert_close(g, ref, RTOL, max(ATOL, 1e-4 * float(ref.abs().max())), f'grad {name}')
    assert n_live > 10
    # the fused training forward (read-out + loss) works for this backbone too
    model.zero_grad(set_to_none=True)
    w_all = torch.ones(data['SNP'].x.shape[0], dtype=torch.float64).cuda()
    l2, pred = model.forward_loss(batch.x_dict, batch.edge_index_dict, 40, batch.n_id('SNP'), batch.dg.y['SNP'], w_all)
    l2.backward()
    assert_close(pred, out.reshape(-1).detach(), 1e-6, 1e-7, 'forward_loss pred')


@pytest.mark.parametrize('backbone,aggr', [('GAT', 'mean'), ('GAT', 'max'), ('GAT', 'min'), ('SAGE', 'mean'), ('SAGE', 'max')])
def test_relation_aggregation_modes_match_reference_restatement(edge_case_graph, backbone, aggr, gemm_routing):
    """gnn_aggr in {mean, min, max} (HeteroConv(aggr), kgwas/model.py:47; README options): 'mean' rides the fused sum
    path (scaled by 1/R), 'min' / 'max' reduce per-relation outputs; prediction, loss and every gradient vs the oracle,
    through forward() and through the fused training forward."""
    data, d = edge_case_graph
    dims = (d['SNP'], d['Gene'], 16)
    model = _model(data, dims, L=2, backbone=backbone, aggr=aggr)
    ids = np.random.default_rng(3).choice(data['SNP'].x.shape[0], size=48, replace=False)
    batch = next(iter(_loader(data, ids, 48)))
    y_all = batch.dg.y['SNP']
    w_all = (torch.rand(data['SNP'].x.shape[0], generator=torch.Generator().manual_seed(1), dtype=torch.float64) + 0.5).cuda()
    n_id = batch.n_id('SNP')[:48].long()
    oracle = oracle_from_product(model)
    x, ei = batch_cpu(batch)
    out_o = oracle(x, ei, 48)
    loss_o = torch.mean(w_all[n_id].cpu() * (out_o.reshape(-1) - y_all[n_id].cpu().double()) ** 2)
    loss_o.backward()
    go = grads_by_name(oracle)
    for fused in (False, True):
        model.zero_grad(set_to_none=True)
        if fused:
            loss, pred = model.forward_loss(batch.x_dict, batch.edge_index_dict, 48, batch.n_id('SNP'), y_all, w_all)
        else:
            out = model(batch.x_dict, batch.edge_index_dict, 48)
            pred = out.reshape(-1)
            loss = torch.mean(w_all[n_id] * (pred - y_all[n_id]) ** 2)
        loss.backward()
        assert_close(pred, out_o.reshape(-1).detach(), RTOL, ATOL, f'pred fused={fused}')
        assert_close(loss.detach(), loss_o.detach(), RTOL, ATOL, 'loss')
        n = 0
        for name, g in grads_by_name(model).items():
            ref = go[name]
            if g is None:
                assert ref is None or float(ref.abs().max()) == 0.0, name
                continue
            n += 1
            assert_close(g, ref, RTOL, max(ATOL, 1e-4 * float(ref.abs().max())), f'grad {name} fused={fused}')
        assert n > 10


def test_more_than_32_relations_match_reference_restatement():
    """42 relations after ToUndirected (12 SNP->Gene, 6 Gene-Gene, 6 Gene->GO + their mirrors): above the 32 relation
    ids one wavefront-wide store of the source-side backward used to cover."""
    from collections import OrderedDict
    from kgwas_amd.graph import HeteroGraph, add_self_loops, to_undirected
    rng = np.random.default_rng(42)
    n = OrderedDict([('SNP', 900), ('Gene', 40), ('CellularComponent', 6), ('BiologicalProcess', 9), ('MolecularFunction', 5)])
    e = OrderedDict()
    for k in range(12):
        m = int(rng.integers(30, 400))
        e[('SNP', f'v2g{k}', 'Gene')] = np.stack([rng.integers(0, 900, m), rng.integers(0, 40, m)])
    for k in range(6):
        m = int(rng.integers(20, 120))
        e[('Gene', f'g2g{k}', 'Gene')] = np.stack([rng.integers(0, 40, m), rng.integers(0, 40, m)])
    for k, t in enumerate(['CellularComponent', 'BiologicalProcess', 'MolecularFunction'] * 2):
        m = int(rng.integers(10, 60))
        e[('Gene', f'g2go{k}', t)] = np.stack([rng.integers(0, 40, m), rng.integers(0, n[t], m)])
    g = torch.Generator().manual_seed(1)
    data = HeteroGraph()
    for t, k in n.items():
        data[t].x = torch.rand(k, 20 if t == 'SNP' else 24 if t == 'Gene' else 16, generator=g)
    for et, ei in add_self_loops(to_undirected(e, n), n).items():
        data[et].edge_index = torch.from_numpy(np.ascontiguousarray(ei))
    assert len(data.edge_types) == 42
    model = _model(data, (20, 24, 16), seed=9)
    ids = rng.choice(900, size=48, replace=False)
    batch = next(iter(_loader(data, ids, 48)))
    out = model(batch.x_dict, batch.edge_index_dict, 48)
    y = torch.rand(48, dtype=torch.float64)
    w = torch.rand(48, dtype=torch.float64) + 0.5
    weighted_mse(out, y.cuda(), w.cuda()).backward()
    oracle = oracle_from_product(model)
    x, ei = batch_cpu(batch)
    out_o = oracle(x, ei, 48)
    weighted_mse(out_o, y, w).backward()
    assert_close(out, out_o.detach(), RTOL, ATOL, 'pred')
    go = grads_by_name(oracle)
    n_live = 0
    for name, gr in grads_by_name(model).items():
        ref = go[name]
        if gr is None:
            assert ref is None or float(ref.abs().max()) == 0.0, name
            continue
        n_live += 1
        assert_close(gr, ref, RTOL, max(ATOL, 1e-4 * float(ref.abs().max())), f'grad {name}')
    assert n_live > 100


def test_relation_vectors_of_all_layers_in_one_launch_equal_per_layer_launches(small_kg):
    """ops.rel_vectors_all (round 4): the relation vectors of every layer from ONE kgw_relvec_fwd_multi launch, their backward from
    ONE kgw_relvec_bwd_multi launch after the first layer's -- against the per-layer nodes (KGW_RELVEC_ALL=0): the same kernels
    on the same operands, so predictions and every gradient are bit-identical."""
    from kgwas_amd import model as kmodel
    data = small_kg.data
    dims = (small_kg.snp_init_dim_size, small_kg.gene_init_dim_size, small_kg.go_init_dim_size)
    ids = np.random.default_rng(3).choice(data['SNP'].x.shape[0], size=48, replace=False)
    batch = next(iter(_loader(data, ids, 48, 2)))
    res = {}
    was = kmodel._RELVEC_ALL
    try:
        for flag in (True, False):
            kmodel._RELVEC_ALL = flag
            m = _model(data, dims, seed=5, L=2)
            out = m(batch.x_dict, batch.edge_index_dict, 48)
            (out ** 2).sum().backward()
            res[flag] = (out.detach().clone(), {k: (v.clone() if v is not None else None) for k, v in grads_by_name(m).items()})
    finally:
        kmodel._RELVEC_ALL = was
    assert torch.equal(res[True][0], res[False][0])
    for k, g in res[True][1].items():
        g0 = res[False][1][k]
        assert (g is None) == (g0 is None), k
        if g is not None:
            assert torch.equal(g, g0), k
