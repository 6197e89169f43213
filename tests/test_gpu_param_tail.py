"""-m gpu: the parameter-only END of a captured step's backward pass as ONE launch (round 5, kgw_param_tail, ops.GradSink.flush)
and the second launches of the relation transforms' products and of the read-out node riding in later launches (KgwTnReducePlan,
KgwReadoutFold; ops._DEFER_REDUCE, ops._DEFER_READOUT_FOLD):
the deferred weight-gradient products of the MLPs, the backward of the FC_output fold (kgw_fold_bwd; kgwas/model.py:15,21 folded
into layer 1) and the backward of the relation vectors of every layer (kgw_relvec_bwd_multi; kgwas/conv.py:138-151) as the blocks
of one grid.  Every block computes with the expressions and in the order of the kernel it replaces => every gradient, every
optimiser state and every parameter must be BIT-identical to the step that issues the three launches one after the other."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 2048], ids=['two_launch_products', 'direct_products'])
def direct_rows(request):
    """Layer 2's transform products have a second launch (the one that rides in layer 1's kgw_transform_bwd) only when products of a
    few hundred rows are NOT run as one row block: both settings of kgw_tn_direct_rows (the shipped 0; 2 048)."""
    from kgwas_amd import _lib
    was = _lib.lib().kgw_tn_direct_rows(request.param)
    yield request.param
    _lib.lib().kgw_tn_direct_rows(was)


@pytest.fixture(scope='module')
def wide_kg():
    from kgwas_amd.kgwas_data import KGWAS_Data
    # (the shape of tests/test_gpu_riders.py: the first gene Linear on the resident kgw_gemm3 route, deferred gene / GO products)
    return KGWAS_Data.from_synthetic(scale=0.23, seed=2, feat_dims={'Gene': 1024}, data_path='/tmp/kgwas_synth_riders')


def _fresh(kg, sd0, seed=21):
    from kgwas_amd.kgwas import KGWAS
    run = KGWAS(kg, device='cuda:0', seed=seed)
    run.initialize_model()
    m = run.model
    if sd0 is None:
        with torch.no_grad():                   # away from the symmetric initial values: every term of the tail is alive
            for pack in list(m.live_packs):
                pack.bias.normal_(0, 0.1)
            for mlp in (m.snp_feat_mlp, m.gene_feat_mlp, m.go_feat_mlp):
                mlp.FC_output.bias.normal_(0, 0.1)
            m.lin.bias.fill_(0.5)
        sd0 = copy.deepcopy(m.state_dict())
    else:
        m.load_state_dict(sd0)
    return run, sd0


def _eager_fused_steps(kg, bs, tail, monkeypatch, sd0, n_steps=3, no_products=False, defer_reduce=None):
    """``n_steps`` eager steps the way the captured step issues them (GradSink + FusedAdam.step_fused) on one batch."""
    from kgwas_amd import ops
    from kgwas_amd.optim import FusedAdam
    from kgwas_amd.sampler import NeighborLoader
    monkeypatch.setattr(ops, '_PARAM_TAIL', tail)
    monkeypatch.setattr(ops, '_DEFER_REDUCE', tail if defer_reduce is None else defer_reduce)
    monkeypatch.setattr(ops, '_DEFER_READOUT_FOLD', tail if defer_reduce is None else defer_reduce)
    if no_products:
        monkeypatch.setattr(ops.GradSink, 'defer_product', lambda self, *a, **k: None)
    run, sd0 = _fresh(kg, sd0)
    m = run.model
    ids = np.asarray(kg.train_input_nodes[1][:bs])
    batch = next(iter(NeighborLoader(kg.data, [-1, -1], ('SNP', ids), batch_size=bs, device='cuda:0')))
    ld_w = run._ld_weight_vector()
    opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=5e-4)
    m.train()
    losses, taken, grads, ridden, folds = [], 0, None, 0, 0
    for _ in range(n_steps):
        opt.zero_grad()
        with ops.readout_fold_deferred():        # (as GraphTrainStep issues it: the read-out's fold is left to the backward pass)
            loss, _ = m.forward_loss(batch.x_dict, batch.edge_index_dict, bs, batch.n_id('SNP'), batch.dg.y['SNP'], ld_w, unit_grad=True)
        sink = ops.GradSink()
        with ops.grad_sink_scope(sink):
            loss.backward(gradient=ops.unit_gradient(loss.device))
        assert (sink.fold_bwd is not None and sink.relvec_bwd is not None) == tail
        opt.step_fused(sink)
        assert not sink.records and sink.fold_bwd is None and sink.relvec_bwd is None
        taken += sink.tail_taken
        ridden += sink.reduces_ridden
        folds += sink.folds_ridden
        assert sink.pending_reduce is None and sink.pending_fold is None
        losses.append(float(loss.detach()))
        if grads is None:
            torch.cuda.synchronize()
            grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
    state = {n: (p.detach().clone(), opt.state[p]['exp_avg'].clone(), opt.state[p]['exp_avg_sq'].clone())
             for n, p in m.named_parameters() if p in opt.state}
    return losses, grads, state, (taken, ridden, folds), sd0


@pytest.mark.parametrize('which,bs,no_products', [('small', 64, False), ('wide', 600, False), ('small', 64, True)])
def test_merged_tail_is_bit_identical_to_the_three_launches(small_kg, wide_kg, which, bs, no_products, monkeypatch, direct_rows):
    kg = small_kg if which == 'small' else wide_kg
    la, ga, sa, ta, sd0 = _eager_fused_steps(kg, bs, True, monkeypatch, None, no_products=no_products)
    lb, gb, sb, tb, _ = _eager_fused_steps(kg, bs, False, monkeypatch, sd0, no_products=no_products)
    assert ta[0] == 3 and ta[2] == 3 and tb == (0, 0, 0), (ta, tb)      # (merged tail and the read-out fold as a rider, every step)
    if bs > 512 and direct_rows == 0:   # (more than two row blocks of 256 seeds: layer 2's products have a second launch, and it must
        assert ta[1] == 3, ta       #  have ridden in layer 1's kgw_transform_bwd, every step)
    assert la == lb and any(l > 0 for l in la)
    assert ga.keys() == gb.keys()
    # the tensors the merged launch writes must be among them, and alive
    for key in ('FC_output.weight', 'FC_output.bias', 'att_src', 'att_dst', 'w_src_t'):
        hit = [n for n in ga if key in n]
        assert hit, key
        assert any(float(ga[n].abs().max()) > 0 for n in hit), key
    for n in ga:
        assert torch.equal(ga[n], gb[n]), (n, float((ga[n] - gb[n]).abs().max()))
    for n in sa:
        for x, y, what in zip(sa[n], sb[n], ('parameter', 'exp_avg', 'exp_avg_sq')):
            assert torch.equal(x, y), (n, what)


@pytest.mark.parametrize('tail,reduce_', [(False, True), (True, False)])
def test_each_merge_alone_is_bit_identical_too(wide_kg, tail, reduce_, monkeypatch, direct_rows):
    """The pending second launches without the merged tail (kgw_fold_bwd / kgw_relvec_bwd_multi then launch them first: they read
    what the sums finish) and the merged tail without pending second launches."""
    la, ga, sa, ta, sd0 = _eager_fused_steps(wide_kg, 600, tail, monkeypatch, None, defer_reduce=reduce_)
    lb, gb, sb, tb, _ = _eager_fused_steps(wide_kg, 600, False, monkeypatch, sd0, defer_reduce=False)
    assert ta[0] == (3 if tail else 0) and ta[1] == (3 if reduce_ and direct_rows == 0 else 0) and ta[2] == (3 if reduce_ else 0) and tb == (0, 0, 0), (ta, tb)
    assert la == lb
    for n in ga:
        assert torch.equal(ga[n], gb[n]), n
    for n in sa:
        for x, y in zip(sa[n], sb[n]):
            assert torch.equal(x, y), n


def test_a_clone_of_a_tail_gradient_is_noticed(small_kg, monkeypatch):
    """The merged launch writes its gradients AFTER autograd has handed them to the parameters: a gradient that was copied on the
    way (here: a hook that returns a new tensor) would reach the optimiser unwritten -- step_fused must refuse the step."""
    from kgwas_amd import ops
    from kgwas_amd.optim import FusedAdam
    from kgwas_amd.sampler import NeighborLoader
    monkeypatch.setattr(ops, '_PARAM_TAIL', True)
    run, _ = _fresh(small_kg, None)
    m = run.model
    ids = np.asarray(small_kg.train_input_nodes[1][:64])
    batch = next(iter(NeighborLoader(small_kg.data, [-1, -1], ('SNP', ids), batch_size=64, device='cuda:0')))
    opt = FusedAdam(m.parameters(), lr=1e-3)
    m.train()
    p = m.gene_feat_mlp.FC_output.weight
    h = p.register_hook(lambda g: g.clone())
    try:
        loss, _ = m.forward_loss(batch.x_dict, batch.edge_index_dict, 64, batch.n_id('SNP'), batch.dg.y['SNP'], run._ld_weight_vector())
        sink = ops.GradSink()
        with ops.grad_sink_scope(sink):
            loss.backward()
        with pytest.raises(ops.GradSinkMismatch):
            opt.step_fused(sink)
    finally:
        h.remove()
        torch.cuda.synchronize()


@pytest.mark.parametrize('which,bs', [('small', 64), ('wide', 600)])
def test_captured_step_with_the_merged_tail_equals_the_step_without(small_kg, wide_kg, which, bs, monkeypatch, direct_rows):
    from kgwas_amd import ops
    from kgwas_amd.graph_step import GraphTrainStep
    from tests.helpers import params_by_name
    kg = small_kg if which == 'small' else wide_kg
    ids = np.asarray(kg.train_input_nodes[1][:bs * 6])
    outs, sd0 = [], None
    for tail in (True, False):
        monkeypatch.setattr(ops, '_PARAM_TAIL', tail)
        monkeypatch.setattr(ops, '_DEFER_REDUCE', tail)
        monkeypatch.setattr(ops, '_DEFER_READOUT_FOLD', tail)
        run, sd0 = _fresh(kg, sd0, seed=13)
        gs = GraphTrainStep(run, ('SNP', ids), bs, lr=1e-3, weight_decay=5e-4)
        assert gs.fused_adam
        run.model.train()
        losses = [float(gs.step(i)) for i in range(5)]
        totals = gs.check()
        assert gs.tail_taken == (1 if tail else 0), gs.tail_taken
        assert gs.reduces_ridden == (1 if tail and bs > 512 and direct_rows == 0 else 0), gs.reduces_ridden
        assert gs.folds_ridden == (1 if tail else 0), gs.folds_ridden
        outs.append((losses, params_by_name(run.model), totals))
    assert outs[0][0] == outs[1][0] and outs[0][2] == outs[1][2]
    assert any(l > 0 for l in outs[0][0])
    for n in outs[0][1]:
        assert torch.equal(outs[0][1][n], outs[1][1][n]), n
