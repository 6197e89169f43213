"""-m gpu: the training step with kgw_gemm3 (k_g3_gemm) IN THE LOOP, against the float64 restatement and against committed
vectors (tests/golden/gat_wide.npz).

The first gene Linear of the reference (kgwas/model.py:13,19 over the 5 120-wide features of kgwas_data.py:236) and its weight
gradient are the two largest products of the benchmark step; the small oracle cases have 24 - 640-wide gene matrices and never
reach the kernel that computes them.  The case of tests/golden/gat_wide_case.py does: 4 613 genes x 1 050 features (neither a
multiple of 32: row AND column padding), every gene in a 512-seed batch, row counts at which every Linear of the step runs on
this package's own kernels -- the tests run with the library-GEMM fallback FORBIDDEN (ops.LIBRARY_GEMM.strict) and assert that
kgw_gemm3 was launched."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle.gat_oracle import weighted_mse
from tests.golden import gat_wide_case as wc
from tests.golden.make_gat_wide_golden import build_oracle, oracle_steps, transformed_edges
from tests.helpers import assert_close, batch_cpu, grads_by_name

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5
HERE = os.path.dirname(os.path.abspath(__file__))


def _golden():
    return np.load(os.path.join(HERE, 'golden', 'gat_wide.npz'), allow_pickle=False)


def _case_graph():
    from kgwas_amd.graph import HeteroGraph
    data = HeteroGraph()
    feats = wc.features()
    for t in wc.NODES:
        data[t].x = torch.from_numpy(feats[t])
    und = transformed_edges()
    for et, ei in und.items():
        data[et].edge_index = ei
    y_all, w_all = wc.labels_and_weights()
    data['SNP'].y = torch.from_numpy(y_all)
    return data, torch.from_numpy(w_all), list(und.keys())


def _product_model(data, edge_types):
    from kgwas_amd.model import HeteroGNN
    m = HeteroGNN(data, wc.HIDDEN, 1, wc.NUM_LAYERS, 'GAT', 'sum', wc.DIMS['SNP'], wc.DIMS['Gene'], wc.DIMS['GO'], 1)
    sd = OrderedDict((k, torch.from_numpy(v)) for k, v in wc.parameters(edge_types).items())
    m.load_state_dict(sd, strict=True)
    return m.cuda()


class _Strict:
    """Library GEMMs forbidden inside the block; kgw_gemm3 launches counted."""

    def __enter__(self):
        from kgwas_amd import ops
        self.ops = ops
        self.was = ops.LIBRARY_GEMM.strict
        ops.LIBRARY_GEMM.strict = True
        ops.LIBRARY_GEMM.reset()
        self.g3 = ops.ROUTES.get('kgw_gemm3', 0)
        return self

    def gemm3_launches(self):
        return self.ops.ROUTES.get('kgw_gemm3', 0) - self.g3

    def __exit__(self, *exc):
        self.ops.LIBRARY_GEMM.strict = self.was
        return False


def _check_inputs(G, und, edge_types):
    chk = np.array([float(sum(int(v.sum()) for v in und.values())),
                    float(sum(float(v.astype(np.float64).sum()) for v in wc.features().values())),
                    float(sum(float(v.astype(np.float64).sum()) for v in wc.parameters(edge_types).values()))])
    assert np.array_equal(chk, G['input_checksum']), 'the case inputs were not regenerated bit for bit'


def test_step_with_gemm3_in_the_loop_matches_restatement_and_golden():
    """forward_loss + backward (the training path: resident first gene Linear, fused feature MLPs, folded FC_output) on
    batch 0 of the case vs the live float64 oracle (every element of every gradient) and vs the committed vectors."""
    from kgwas_amd.sampler import NeighborLoader
    G = _golden()
    data, w_all, edge_types = _case_graph()
    _check_inputs(G, {et: data[et].edge_index.numpy() for et in edge_types}, edge_types)
    model = _product_model(data, edge_types)
    ids = wc.seeds()[:wc.BATCH]
    batch = next(iter(NeighborLoader(data, [-1, -1], ('SNP', ids), batch_size=wc.BATCH, device='cuda:0')))
    assert batch.n_nodes['Gene'] == wc.NODES['Gene'] and batch.n_nodes['SNP'] >= 16384
    assert sum(batch.n_nodes[t] for t in wc.GO_TYPES) >= 4096
    ld_w = w_all.cuda()
    model.train()
    with _Strict() as st:
        loss, pred = model.forward_loss(batch.x_dict, batch.edge_index_dict, wc.BATCH, batch.n_id('SNP'), batch.dg.y['SNP'], ld_w)
        loss.backward()
        torch.cuda.synchronize()
        assert st.gemm3_launches() == 2, 'first gene Linear forward + its weight gradient must both run on kgw_gemm3'
        assert st.ops.LIBRARY_GEMM.calls == 0

    # --- committed vectors ------------------------------------------------------------------------------------------
    assert_close(pred, G['pred'], RTOL, ATOL, 'pred vs golden')
    assert abs(float(loss) - float(G['loss'])) <= 1e-4 * abs(float(G['loss']))
    gp = grads_by_name(model)
    stride = int(G['grad_stride'])
    n_gold = 0
    for name, g in gp.items():
        if g is None:
            assert name in set(G['grad_none'].tolist()) or f'g_{name}' not in G.files, name
            continue
        if f'gs_{name}' in G.files:
            ref = torch.from_numpy(G[f'gs_{name}'])
            scale = float(G[f'gn_{name}'][1]) / max(g.numel(), 1) ** 0.5          # rms of the gradient
            assert_close(g.reshape(-1)[::stride], ref, RTOL, max(ATOL, 1e-3 * scale), f'golden grad {name}')
            n_gold += 1
        elif f'g_{name}' in G.files:
            ref = torch.from_numpy(G[f'g_{name}'])
            assert_close(g.reshape(ref.shape), ref, RTOL, max(ATOL, 1e-4 * float(ref.abs().max())), f'golden grad {name}')
            n_gold += 1
    assert n_gold > 40 and 'gs_gene_feat_mlp.FC_hidden.weight' in G.files

    # --- live oracle on the product's own sampled subgraph: every gradient element ----------------------------------------
    oracle = build_oracle(edge_types)
    x, ei = batch_cpu(batch)
    out_o = oracle(x, ei, wc.BATCH)
    s = torch.as_tensor(ids)
    loss_o = weighted_mse(out_o, data['SNP'].y[s].double(), w_all[s])
    loss_o.backward()
    assert_close(pred, out_o.detach().reshape(-1), RTOL, ATOL, 'pred vs oracle')
    go = grads_by_name(oracle)
    for name, g in gp.items():
        ref = go[name]
        if g is None:
            assert ref is None or float(ref.abs().max()) == 0.0, name
            continue
        assert_close(g, ref, RTOL, max(ATOL, 1e-4 * float(ref.abs().max())), f'grad {name}')
    # the wide layer specifically, relative to its own magnitude
    gw, rw = gp['gene_feat_mlp.FC_hidden.weight'].double(), go['gene_feat_mlp.FC_hidden.weight']
    assert float((gw - rw).norm() / rw.norm()) < 1e-5
    # identical top-k ranking of the seeds (north_star)
    k = 20
    assert torch.equal(torch.topk(pred.cpu().double(), k).indices, torch.topk(out_o.detach().reshape(-1), k).indices)


def test_captured_adam_steps_with_gemm3_follow_the_committed_trajectory():
    """N_STEPS captured training steps (GraphTrainStep: sampling graph + step graph, the benchmark's execution mode) on the
    case's batches vs the committed float64 losses and the committed update of the wide first gene layer's weight."""
    from kgwas_amd.graph_step import GraphTrainStep
    from kgwas_amd.kgwas import KGWAS
    G = _golden()
    data, w_all, edge_types = _case_graph()

    class _D:            # the slice of KGWAS_Data the trainer reads (kgwas/kgwas.py:65-67,101-109)
        pass
    d = _D()
    d.data, d.data_path = data, '/tmp/kgwas_wide_case'
    d.snp_init_dim_size, d.gene_init_dim_size, d.go_init_dim_size = wc.DIMS['SNP'], wc.DIMS['Gene'], wc.DIMS['GO']
    run = KGWAS(d, device='cuda:0', seed=1)
    run.initialize_model()
    run.model.load_state_dict(OrderedDict((k, torch.from_numpy(v)) for k, v in wc.parameters(edge_types).items()), strict=True)
    run._ld_weight_vector = lambda: w_all.cuda()
    w0 = run.model.gene_feat_mlp.FC_hidden.weight.detach().double().cpu().clone()
    ids = wc.seeds()
    with _Strict() as st:
        gs = GraphTrainStep(run, ('SNP', ids), wc.BATCH, lr=wc.LR, weight_decay=wc.WEIGHT_DECAY)
        run.model.train()
        losses = [float(gs.step(i)) for i in range(wc.N_STEPS)]
        gs.check()
        assert st.gemm3_launches() >= 2 and st.ops.LIBRARY_GEMM.calls == 0
    for i, (a, b) in enumerate(zip(losses, G['losses'])):
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-7, (i, a, b)
    dw = (run.model.gene_feat_mlp.FC_hidden.weight.detach().double().cpu() - w0).reshape(-1)[::int(G['grad_stride'])]
    ref = torch.from_numpy(G['dW_gene_first_strided'])
    # Adam turns fp32-noise gradients into +-lr steps: compare the update in norm (as tests/test_gpu_model.py does)
    assert float((dw - ref).norm() / ref.norm()) < 2e-2
    assert float((dw - ref).abs().max()) <= 2.5 * wc.LR * wc.N_STEPS


def test_library_fallback_is_logged_and_strict_mode_raises():
    """A shape none of the package's kernels takes is a recorded library call by default and an error under KGW_STRICT."""
    from kgwas_amd import ops
    x = torch.randn(100, 3000, device='cuda')                 # K > 2304 on few rows: kgw_linear does not take it
    w = torch.randn(128, 3000, device='cuda')
    ops.LIBRARY_GEMM.reset()
    was = ops.LIBRARY_GEMM.strict
    try:
        ops.LIBRARY_GEMM.strict = False
        y = ops.linear(x, w)
        assert ops.LIBRARY_GEMM.calls == 1 and list(ops.LIBRARY_GEMM.by_site)[0][0] == 'linear'
        assert_close(y, x.double() @ w.double().t(), 1e-4, 1e-3, 'library-routed linear')
        ops.LIBRARY_GEMM.strict = True
        with pytest.raises(RuntimeError, match='KGW_STRICT'):
            ops.linear(x, w)
    finally:
        ops.LIBRARY_GEMM.strict = was
