"""-m gpu: the HIP-graph training step (static layout, kgwas_amd/graph_step.py) against the eager path on the
same batches: identical losses, identical parameter trajectory (up to fp32 summation order -- padded GEMMs
may tile differently), and the static-layout padding rows carry no gradient."""
import copy

import numpy as np
import pytest
import torch

from tests.helpers import assert_close, params_by_name

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('layers', [2, 1])
def test_graph_step_equals_eager(small_kg, layers):
    """(layers = 1: the model whose relation vectors go through the per-layer node _RelVectors, which must launch whatever the
    captured step's sink still holds pending -- the FC_output fold's backward, a transform's second launch -- before it reads
    their outputs)"""
    from kgwas_amd.graph_step import GraphTrainStep
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.sampler import NeighborLoader
    bs, nsteps = 64, 5
    ids = np.asarray(small_kg.train_input_nodes[1][:bs * 8])
    run_e = KGWAS(small_kg, device='cuda:0', seed=11)
    run_e.initialize_model(gnn_num_layers=layers)
    run_g = KGWAS(small_kg, device='cuda:0', seed=12)
    run_g.initialize_model(gnn_num_layers=layers)
    run_g.model.load_state_dict(run_e.model.state_dict())
    p0 = params_by_name(run_e.model)

    gs = GraphTrainStep(run_g, ('SNP', ids), bs, lr=1e-3, weight_decay=5e-4)
    assert gs.n_batches == 8
    # capture warm-up must not have moved the weights
    for n, p in params_by_name(run_g.model).items():
        assert torch.equal(p, p0[n]), n

    opt = torch.optim.Adam(run_e.model.parameters(), lr=1e-3, weight_decay=5e-4)
    ld_w = run_e._ld_weight_vector()
    run_e.model.train()
    run_g.model.train()
    it = iter(NeighborLoader(small_kg.data, [-1] * layers, ('SNP', ids), batch_size=bs, drop_last=True, device='cuda:0'))
    edges_eager = 0
    for i in range(nsteps):
        batch = next(it)
        le = run_e.train_step(batch, opt, ld_w)
        lg = gs.step(i)
        edges_eager += sum(batch.n_edges_per_layer)
        assert_close(lg.detach().clone(), le.detach(), 1e-5, 1e-7, f'loss step {i}')
    stats = gs.check()
    assert sum(stats[:layers]) == edges_eager                  # same edges aggregated, counted on the device
    pe, pg = params_by_name(run_e.model), params_by_name(run_g.model)
    num = den = 0.0
    for n in pe:
        num += float((pe[n] - pg[n]).pow(2).sum())
        den += float((pe[n] - p0[n]).pow(2).sum())
    assert den > 0 and (num / den) ** 0.5 < 5e-3, f'relative update difference {(num / den) ** 0.5:.3e}'


def test_cached_epochs_are_bit_identical_to_sampled_ones(small_kg):
    """the loader's batch order is fixed (kgwas/kgwas.py:93-101), so KGWAS.train keeps the batches sampled in
    epoch 1 (graph_step.BatchCache, kgw_segments_copy) and puts them back in later epochs instead of sampling them again.  Three
    passes over the loader with the cache against three passes that sample every batch: every loss, every parameter, every
    optimiser moment and the device-side edge counts bit for bit -- for an even and an odd number of batches (the odd one starts
    every pass on the other buffer and goes through the out-of-sequence path)."""
    from kgwas_amd.graph_step import GraphTrainStep
    from kgwas_amd.kgwas import KGWAS
    bs = 64
    for nb in (6, 5):
        ids = np.asarray(small_kg.train_input_nodes[1][:bs * nb])
        outs = []
        for cached in (True, False):
            run = KGWAS(small_kg, device='cuda:0', seed=17)
            run.initialize_model()
            if outs:
                run.model.load_state_dict(sd0)
            else:
                sd0 = copy.deepcopy(run.model.state_dict())
            gs = GraphTrainStep(run, ('SNP', ids), bs, lr=1e-3, weight_decay=5e-4, cache_batches=cached)
            assert gs.n_batches == nb and (gs.cache is not None) == cached
            run.model.train()
            losses = []
            for ep in range(3):
                for i in range(nb):
                    losses.append(gs.step(i).detach().clone())
                stats = gs.check()
            torch.cuda.synchronize()
            if cached:
                # every batch sampled exactly once (epoch 1), everything afterwards put back from its slot
                assert gs.cache.saved == nb and all(gs.cache.filled), (gs.cache.saved, gs.cache.filled)
                assert gs.cache.restored >= 2 * nb - 1, gs.cache.restored
            state = {n: (p.detach().clone(), gs.opt.state[p]['exp_avg'].clone(), gs.opt.state[p]['exp_avg_sq'].clone())
                     for n, p in run.model.named_parameters() if p in gs.opt.state}
            outs.append((torch.stack(losses), state, stats))
        assert torch.equal(outs[0][0], outs[1][0]) and float(outs[0][0].abs().sum()) > 0
        assert outs[0][2] == outs[1][2]
        assert outs[0][1].keys() == outs[1][1].keys()
        for n in outs[0][1]:
            for a, b in zip(outs[0][1][n], outs[1][1][n]):
                assert torch.equal(a, b), n


def test_kgwas_train_with_kept_batches_equals_train_that_samples_every_epoch(small_kg, monkeypatch):
    """The same through the reference's API: KGWAS.train(epoch=3) (kgwas/kgwas.py:85-212) keeps the batches of epoch 1 by default;
    with KGW_EPOCH_CACHE=0 it samples every batch of every epoch.  Validation metrics of the last epoch, test metrics, the
    whole-genome predictions and every parameter of the best model: bit for bit."""
    from kgwas_amd import graph_step
    from kgwas_amd.kgwas import KGWAS
    outs, made = [], []
    real_init = graph_step.BatchCache.__init__

    def counting_init(self, *a, **k):
        real_init(self, *a, **k)
        made.append(self)
    monkeypatch.setattr(graph_step.BatchCache, '__init__', counting_init)
    for keep in ('1', '0'):
        monkeypatch.setenv('KGW_EPOCH_CACHE', keep)
        run = KGWAS(small_kg, device='cuda:0', seed=23)
        run.initialize_model()
        if outs:
            run.model.load_state_dict(sd0)
        else:
            sd0 = copy.deepcopy(run.model.state_dict())
        n_before = len(made)
        run.train(batch_size=64, epoch=3, save_best_model=False, save_name='keep' + keep)
        assert (len(made) > n_before) == (keep == '1')
        if keep == '1':
            c = made[-1]
            assert c.saved == c.n_batches and c.restored >= 2 * c.n_batches - 1, (c.saved, c.restored, c.n_batches)
        outs.append((dict(run.val_metrics), dict(run.test_metrics), np.asarray(run.kgwas_res['pred'].values).copy(),
                     {k: v.detach().clone() for k, v in run.best_model.named_reference_tensors().items()}))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    assert np.array_equal(outs[0][2], outs[1][2]) and float(np.abs(outs[0][2]).sum()) > 0
    for k in outs[0][3]:
        assert torch.equal(outs[0][3][k], outs[1][3][k]), k


def test_static_capacity_overflow_is_reported(small_kg):
    """A batch that needs more rows than the static layout holds must raise, never silently truncate."""
    from kgwas_amd import _lib
    from kgwas_amd.graph_step import GraphTrainStep
    from kgwas_amd.kgwas import KGWAS
    ids = np.asarray(small_kg.train_input_nodes[1][:64 * 4])
    run = KGWAS(small_kg, device='cuda:0', seed=3)
    run.initialize_model()
    gs = GraphTrainStep(run, ('SNP', ids), 64, margin=1.0)
    # shrink a capacity below what batch 1 needs by capturing on caps measured from batch 0 only
    gs2 = GraphTrainStep(run, ('SNP', ids[:64]), 64, margin=1.0)
    gs2.ids = gs.ids
    gs2.n_batches = 4
    overflow = False
    for i in range(4):
        gs2.step(i)
    try:
        gs2.check()
    except _lib.KgwasHipError:
        overflow = True
    caps_equal = gs2.caps.node_off == gs.caps.node_off
    assert overflow or caps_equal
    # the non-blocking poll KGWAS.train uses inside an epoch reports the same thing, one poll late at most
    if overflow:
        import torch as _t
        seen = False
        for _ in range(3):
            try:
                gs2.poll()
            except _lib.KgwasHipError:
                seen = True
                break
            _t.cuda.synchronize()
        assert seen


@pytest.mark.parametrize('eval_batch', [64, 1 << 20])
@pytest.mark.parametrize('drop_last', [False, True])
def test_graph_eval_equals_eager_eval(small_kg, drop_last, eval_batch, monkeypatch):
    """evaluate_minibatch_clean through the captured forward (GraphEvalStep) == the eager per-batch loop, including
    the partial last batch of a drop_last=False loader -- in batches of the loader's own size (padded last batch, two buffers,
    side sampler) and as ONE batch of all nodes (the default: a seed's prediction does not depend on its batch mates)."""
    from kgwas_amd import graph_step
    from kgwas_amd.kgwas import KGWAS
    monkeypatch.setattr(graph_step, 'EVAL_BATCH', eval_batch)
    from kgwas_amd.sampler import NeighborLoader
    from kgwas_amd.utils import evaluate_minibatch_clean
    run = KGWAS(small_kg, device='cuda:0', seed=4)
    run.initialize_model()
    ids = np.asarray(small_kg.test_input_nodes[1])[:64 * 5 + 17]
    mk = lambda: NeighborLoader(small_kg.data, [-1, -1], ('SNP', ids), batch_size=64, drop_last=drop_last, device='cuda:0')
    a = evaluate_minibatch_clean(mk(), run.model, 'cuda:0')
    monkeypatch.setenv('KGW_EVAL_EAGER', '1')
    b = evaluate_minibatch_clean(mk(), run.model, 'cuda:0')
    n = (len(ids) // 64) * 64 if drop_last else len(ids)
    assert len(ids) % 64 != 0 and len(ids) >= 3 * 64
    assert a['pred'].shape == b['pred'].shape == (n,)
    assert np.array_equal(a['truth'], b['truth'])
    np.testing.assert_allclose(a['pred'], b['pred'], rtol=1e-5, atol=1e-6)


def test_measure_overlap_leaves_the_training_state_alone(small_kg):
    """GraphTrainStep.measure_overlap (what bench.py calls before its epoch measurement) runs real training steps, half
    of them on stale batches -- parameters, optimiser moments, step counter and running totals must be what they were, and the
    trajectory after it the one of a trainer that never measured."""
    from kgwas_amd.graph_step import GraphTrainStep
    from kgwas_amd.kgwas import KGWAS
    bs = 64
    ids = np.asarray(small_kg.train_input_nodes[1][:bs * 8])
    runs, outs = [], []
    for measure in (True, False):
        run = KGWAS(small_kg, device='cuda:0', seed=31)
        run.initialize_model()
        if runs:
            run.model.load_state_dict(sd0)
        else:
            sd0 = copy.deepcopy(run.model.state_dict())
        gs = GraphTrainStep(run, ('SNP', ids), bs, lr=1e-3, weight_decay=5e-4)
        runs.append(run)
        run.model.train()
        for i in range(3):
            gs.step(i)
        if measure:
            before = params_by_name(run.model)
            step_before, stats_before = int(gs.opt.step_dev[0]), gs.stats.clone()
            res = gs.measure_overlap(4)
            assert 'overlap_ratio' in res
            for n, p in params_by_name(run.model).items():
                assert torch.equal(p, before[n]), n
            assert int(gs.opt.step_dev[0]) == step_before and torch.equal(gs.stats, stats_before)
        for i in range(3, 6):
            gs.step(i)
        outs.append((params_by_name(run.model), gs.check()))
    for n in outs[0][0]:
        assert torch.equal(outs[0][0][n], outs[1][0][n]), n
    assert outs[0][1] == outs[1][1]
