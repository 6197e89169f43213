"""-m gpu: the fused optimiser launch (kgw_adam_fused; kgwas/kgwas.py:116,151 = torch.optim.Adam with L2 weight decay).

The weight-gradient products that feed only Adam (the Linears of kgwas/model.py:13-21) may stop after their first launch and
leave per-block partial sums; kgw_adam_fused finishes the sums in the producers' own order while it updates the parameters.  The
bar is BIT-IDENTITY with the unfused path (kgw_tn_gemm_ex / kgw_mlp2_bwd_first + kgw_adam): gradients left in the gradient
tensors, parameters, both Adam moments, the step counter -- and, one level up, the whole captured training step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _adam_pair(shapes, seed):
    from kgwas_amd.optim import FusedAdam
    g = torch.Generator(device='cpu').manual_seed(seed)
    a = [torch.randn(*s, generator=g).to(DEV).requires_grad_() for s in shapes]
    b = [p.detach().clone().requires_grad_() for p in a]
    return a, b, FusedAdam(a, lr=1e-2, weight_decay=5e-4), FusedAdam(b, lr=1e-2, weight_decay=5e-4)


def _same_state(pa, pb, oa, ob):
    for k, (x, y) in enumerate(zip(pa, pb)):
        assert torch.equal(x.grad, y.grad), (k, float((x.grad - y.grad).abs().max()), int((x.grad != y.grad).sum()))
        assert torch.equal(x, y), (k, float((x - y).abs().max()), int((x != y).sum()))
        assert torch.equal(oa.state[x]['exp_avg'], ob.state[y]['exp_avg'])
        assert torch.equal(oa.state[x]['exp_avg_sq'], ob.state[y]['exp_avg_sq'])
    assert int(oa.step_dev) == int(ob.step_dev)


# rows x M x N of C = A^T B: the step's own shapes (122 k sampled SNPs, 20 k genes: 128 x 128 on the <2,2> tiling), a narrow B
# (<2,1>), a narrow A (<1,4>, <1,2>), odd sizes (<1,1>), a product with a single row block (finished by its producer)
@pytest.mark.parametrize('rows,M,N,tr', [(122880, 128, 128, False), (20032, 128, 128, True), (40000, 128, 20, False),
                                         (30000, 6, 128, False), (30000, 6, 66, True), (9000, 5, 7, False), (200, 128, 128, False)])
def test_tn_partial_sums_finished_by_adam(rows, M, N, tr):
    from kgwas_amd import ops
    g = torch.Generator(device='cpu').manual_seed(rows + M)
    A = torch.randn(rows, M, generator=g).to(DEV)
    B = torch.randn(rows, N, generator=g).to(DEV)
    shape = (N, M) if tr else (M, N)
    pa, pb, oa, ob = _adam_pair([shape, (M,)], 3)
    for step in range(3):
        ref, ref_cs = ops.tn_gemm(A, B, colsum=True, transpose_out=tr)
        sink = ops.GradSink()
        with ops.grad_sink_scope(sink):
            out, cs = ops.tn_gemm(A, B, colsum=True, transpose_out=tr, defer=True)
        if rows > 1024:
            assert len(sink.records) == 2                 # both gradients wait for the optimiser
        pa[0].grad, pa[1].grad = out, cs
        pb[0].grad, pb[1].grad = ref, ref_cs
        oa.step_fused(sink)
        ob.step()
        assert not sink.records
        _same_state(pa, pb, oa, ob)
        A = A * 0.5 + 0.1
    assert int(oa.step_dev) == 3 and int(oa.done_dev.abs().sum()) == 0


def test_grouped_products_and_leftover_record(monkeypatch):
    from kgwas_amd import ops
    monkeypatch.setattr(ops, '_DEFER_PRODUCTS', False)
    g = torch.Generator(device='cpu').manual_seed(5)
    rows = 9000
    dY1, X1 = torch.randn(rows, 128, generator=g).to(DEV), torch.randn(rows, 128, generator=g).to(DEV)
    dY2, X2 = torch.randn(rows, 128, generator=g).to(DEV), torch.randn(rows, 64, generator=g).to(DEV)
    pa, pb, oa, ob = _adam_pair([(128, 128), (128,), (128, 64), (128,)], 9)
    ref = ops.weight_grads([(dY1, X1), (dY2, X2)])
    sink = ops.GradSink()
    with ops.grad_sink_scope(sink):
        got = ops.weight_grads([(dY1, X1), (dY2, X2)])
    assert len(sink.records) == 4
    for k in range(2):
        pa[2 * k].grad, pa[2 * k + 1].grad = got[k]
        pb[2 * k].grad, pb[2 * k + 1].grad = ref[k]
    oa.step_fused(sink)
    ob.step()
    _same_state(pa, pb, oa, ob)
    # a record nobody claims (autograd copied the gradient, or it never reached a parameter): loud, and nothing is launched
    sink = ops.GradSink()
    with ops.grad_sink_scope(sink):
        out, cs = ops.tn_gemm(dY1, X1, colsum=True, defer=True)
    before = [p.detach().clone() for p in pa]
    pa[0].grad, pa[1].grad = out.clone(), cs
    with pytest.raises(ops.GradSinkMismatch):
        oa.step_fused(sink)
    for p, q in zip(pa, before):
        assert torch.equal(p, q)


@pytest.mark.parametrize('rows,K1,defer_products', [(40000, 20, False), (16384, 4, False), (122880, 20, False), (20000, 20, True),
                                                   (200, 20, True)])
def test_mlp2_first_layer_partials_finished_by_adam(rows, K1, defer_products, monkeypatch):
    """_MLP2's backward (kgwas/model.py:18-20 on the 20-wide SNP features): d W1 / d b1 from kgw_mlp2_bwd_first's block partials,
    d W2 / d b2 from the split-K product's -- all four finished inside kgw_adam_fused."""
    from kgwas_amd import ops
    # (defer_products: a SHORT d W2 product -- under 32 768 rows -- is launched by step_fused, grouped with whatever else was
    #  deferred; 200 rows: a product with one row block, complete after its first launch, still matched)
    monkeypatch.setattr(ops, '_DEFER_PRODUCTS', defer_products)
    g = torch.Generator(device='cpu').manual_seed(rows)
    x = torch.rand(rows, K1, generator=g).to(DEV)
    dh2 = (torch.randn(rows, 128, generator=g) * (torch.rand(rows, 128, generator=g) > 0.5)).to(DEV)
    shapes = [(128, K1), (128,), (128, 128), (128,)]
    pa, pb, oa, ob = _adam_pair(shapes, 21)
    with torch.no_grad():
        for ps in (pa, pb):
            ps[0].mul_(0.3); ps[2].mul_(0.1)
    for step in range(2):
        for ps, opt, fused in ((pa, oa, True), (pb, ob, False)):
            opt.zero_grad(set_to_none=True)
            h2 = ops.mlp2(x, *ps)
            sink = ops.GradSink() if fused else None
            with ops.grad_sink_scope(sink):
                h2.backward(dh2)
            if fused:
                if rows >= 16384:
                    assert len(sink.records) + 2 * len(sink.products) == 4    # (products: only with KGW_DEFER_PRODUCTS=1)
                    assert len(sink.products) == (1 if defer_products else 0)
                opt.step_fused(sink)
            else:
                opt.step()
        _same_state(pa, pb, oa, ob)


@pytest.mark.parametrize('size', ['small', 'medium'])
def test_fused_step_equals_unfused_step(small_kg, monkeypatch, size):
    """The captured training step with the fused optimiser launch == the same step with the folds, Adam and the statistics as
    launches of their own: losses, every parameter, the running totals -- bit for bit.  ``medium`` (21 % of the benchmark graph,
    256 seeds: 4 200 genes with the 5 120-wide features, tens of thousands of sampled SNP rows) puts the first gene Linear on
    kgw_gemm3 and makes the MLPs' weight gradients multi-block products: the deferred sums are really taken by the optimiser's
    launch and the gene weight's operand image is written by it; ``small`` has single-block products (only the counters move into
    the launch).  Half way the gene weight is changed behind the trainers' backs: the image must follow."""
    from kgwas_amd import ops
    from kgwas_amd.graph_step import GraphTrainStep
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.kgwas_data import KGWAS_Data
    from tests.helpers import params_by_name
    if size == 'medium':
        small_kg = KGWAS_Data.from_synthetic(scale=0.21, seed=1, data_path='/tmp/kgwas_synth_medium')
    bs, nsteps = (64, 6) if size == 'small' else (256, 6)
    ids = np.asarray(small_kg.train_input_nodes[1][:bs * 8])
    runs, steps = [], []
    for fused in (True, False):
        monkeypatch.setattr(ops, '_FUSED_ADAM', fused)
        run = KGWAS(small_kg, device=DEV, seed=11)
        run.initialize_model()
        if runs:
            run.model.load_state_dict(runs[0].model.state_dict())
        runs.append(run)
        gs = GraphTrainStep(run, ('SNP', ids), bs, lr=1e-3, weight_decay=5e-4)
        assert gs.fused_adam == fused
        if fused and size == 'medium':
            assert gs.deferred_gradients >= 5, gs.deferred_gradients          # (incl. the gene layer's kgw_gemm3 weight gradient)
            assert len(gs._image_params) == 1 and gs._image_params[0] is run.model.gene_feat_mlp.FC_hidden.weight
        if not fused:
            assert not gs._image_params
        steps.append(gs)
    # (the first trainer was built before the second copied its weights: both start from the same state)
    losses = [[], []]
    for k, gs in enumerate(steps):
        for i in range(nsteps):
            if i == nsteps // 2:
                with torch.no_grad():
                    runs[k].model.gene_feat_mlp.FC_hidden.weight.mul_(1.03125)
            losses[k].append(float(gs.step(i)))
    assert losses[0] == losses[1]
    assert len(set(losses[0])) == nsteps
    assert steps[0].check() == steps[1].check()
    assert int(steps[0].opt.step_dev) == int(steps[1].opt.step_dev) == nsteps
    pa, pb = params_by_name(runs[0].model), params_by_name(runs[1].model)
    for n in pa:
        assert torch.equal(pa[n], pb[n]), n
    if size == 'medium':
        # the image the optimiser launch left == what kgw_gemm3_pack makes of the final weight
        gs = steps[0]
        W = gs._image_params[0]
        ref = ops.gemm3_pack(W.detach(), W.shape[1], False)
        assert torch.equal(ref, gs._images[W.data_ptr()])


@pytest.mark.parametrize('M,K', [(5120, 20032), (1024, 4096), (96, 2048)])
def test_gemm3_weight_gradient_finished_by_adam(M, K):
    """d W1 [128, M] = (A [M, K] B [K, 128])^T on kgw_gemm3 with its K ranges added by kgw_adam_fused (KGW_GRAD_G3T) + the
    updated weight's operand image written by the same launch: gradient, parameter, moments and image bit-identical to
    kgw_gemm3(transpose_out) + kgw_adam + kgw_gemm3_pack."""
    from kgwas_amd import _lib, ops
    g = torch.Generator(device='cpu').manual_seed(M)
    A = torch.randn(M, K, generator=g).to(DEV)
    B = (torch.randn(K, 128, generator=g) * 0.1).to(DEV)
    pa, pb, oa, ob = _adam_pair([(128, M)], 4)
    img = torch.empty(int(_lib.lib().kgw_gemm3_packed_bytes(M)), dtype=torch.uint8, device=DEV)
    oa.packed_images[pa[0]] = img
    for step in range(3):
        packed = ops.gemm3_pack(B, K, True)
        ref = ops.gemm3(A, packed, transpose_out=True)
        sink = ops.GradSink()
        with ops.grad_sink_scope(sink):
            out = ops.gemm3(A, packed, transpose_out=True, defer=True)
        assert len(sink.records) == 1
        pa[0].grad, pb[0].grad = out, ref
        oa.step_fused(sink)
        ob.step()
        _same_state(pa, pb, oa, ob)
        assert torch.equal(img, ops.gemm3_pack(pb[0].detach(), M, False))
        B = B * 0.5


def test_finish_into_flat_bucket():
    """kgw_grad_finish (multi-GPU step: the gradients must be complete BEFORE the all-reduce, so the optimiser's launch cannot take the
    last sums): one launch writes every finished gradient into its slot of a flat bucket -- bit-identical to the producers' own second
    launches followed by the concatenation, for copied (complete) gradients, split-K sums, first-layer partials and the kgw_gemm3
    weight gradient alike."""
    from kgwas_amd import ops
    from kgwas_amd.optim import FusedAdam
    g = torch.Generator(device='cpu').manual_seed(77)
    rows, K1, M3, K3 = 40000, 20, 1024, 4096
    x = torch.rand(rows, K1, generator=g).to(DEV)
    dh2 = (torch.randn(rows, 128, generator=g) * (torch.rand(rows, 128, generator=g) > 0.5)).to(DEV)
    A3 = torch.randn(M3, K3, generator=g).to(DEV)
    B3 = (torch.randn(K3, 128, generator=g) * 0.1).to(DEV)
    shapes = [(128, K1), (128,), (128, 128), (128,), (128, M3), (33, 7)]
    ps = [torch.randn(*s_, generator=g).to(DEV).requires_grad_() for s_ in shapes]
    with torch.no_grad():
        ps[0].mul_(0.3); ps[2].mul_(0.1)
    extra = torch.randn(33, 7, generator=g).to(DEV)                 # a complete gradient: copied

    def grads(sink):
        for p in ps:
            p.grad = None
        h2 = ops.mlp2(x, *ps[:4])
        with ops.grad_sink_scope(sink):
            h2.backward(dh2)
            ps[4].grad = ops.gemm3(A3, ops.gemm3_pack(B3, K3, True), transpose_out=True, defer=True)
        ps[5].grad = extra.clone()

    grads(None)
    ref = torch.cat([p.grad.reshape(-1) for p in ps])
    sink = ops.GradSink()
    grads(sink)
    assert len(sink.records) == 5
    flat = torch.full_like(ref, float('nan'))
    views, off = [], 0
    for p in ps:
        views.append(flat[off:off + p.numel()].view_as(p)); off += p.numel()
    FusedAdam(ps).finish_into(sink, list(zip(ps, views)))
    assert not sink.records
    assert torch.equal(flat, ref)
    assert torch.equal(torch.cat([p.grad.reshape(-1) for p in ps]), ref)        # the gradient tensors hold the sums too


def test_full_size_trajectory_identical_with_and_without_the_launch_merges():
    """tools/fused_vs_unfused.py: 120 captured steps on the benchmark workload (full-size fast-mode graph, batch 512) with every round-4
    launch merge on -- fused optimiser launch + operand image, merged transform backward, grouped short products -- and with all of them
    off: losses and every parameter bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fused_vs_unfused.py'), '120'], cwd=root, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'losses identical True' in r.stdout and ' state tensors bit-identical' in r.stdout


def test_fallback_when_a_deferred_gradient_does_not_reach_its_parameter(monkeypatch, capsys):
    """A model for which the fused launches do not apply (here: forced -- GradSink.take loses one record, as if autograd had copied a
    gradient) must notice in the capture warm-up, BEFORE any update, switch every fused form off (incl. the operand images) and train
    exactly like the unfused step."""
    from kgwas_amd import ops
    from kgwas_amd.graph_step import GraphTrainStep
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.kgwas_data import KGWAS_Data
    from tests.helpers import params_by_name
    data = KGWAS_Data.from_synthetic(scale=0.21, seed=1, data_path='/tmp/kgwas_synth_medium')
    ids = np.asarray(data.train_input_nodes[1][:256 * 6])
    runs, steps = [], []
    real_take = ops.GradSink.take
    for broken in (True, False):
        if broken:
            for flag in ('_FUSED_ADAM', '_MERGED_TRANSFORM_BWD', '_DEFER_PRODUCTS', '_DUV_PIECES'):
                monkeypatch.setattr(ops, flag, True)           # (whatever the environment says)
            lost = []

            def take(self, grad):
                r = real_take(self, grad)
                if r is not None and r[0] is not None and not lost:
                    lost.append(1)
                    self.records[grad.data_ptr()] = r          # (put it back: nobody will claim it)
                    return None
                return r
            monkeypatch.setattr(ops.GradSink, 'take', take)
        else:
            monkeypatch.setattr(ops.GradSink, 'take', real_take)
            monkeypatch.setattr(ops, '_FUSED_ADAM', False)
        run = KGWAS(data, device=DEV, seed=11)
        run.initialize_model()
        if runs:
            run.model.load_state_dict(runs[0].model.state_dict(), strict=False)
        runs.append(run)
        gs = GraphTrainStep(run, ('SNP', ids), 256, lr=1e-3, weight_decay=5e-4)
        assert not gs.fused_adam and not gs._image_params and not gs.opt.packed_images
        steps.append(gs)
    assert 'not used' in capsys.readouterr().err
    for gs in steps:
        for i in range(4):
            gs.step(i)
    assert steps[0].check() == steps[1].check()
    pa, pb = params_by_name(runs[0].model), params_by_name(runs[1].model)
    for n in pa:
        assert torch.equal(pa[n], pb[n]), n


def test_image_owning_weight_updated_from_another_kind_of_gradient_is_refused():
    """``step_fused`` rewrites a weight's persistent kgw_gemm3 operand image only when the weight's gradient arrives as
    a KGW_GRAD_G3T record.  Any other kind of gradient (here: a complete tensor) would change the weight and leave the image the
    next forward reads stale, silently -- the launch must be refused (GradSinkMismatch, nothing launched: the trainer's warm-up
    then falls back to the unfused step, which packs in the forward)."""
    from kgwas_amd import ops
    from kgwas_amd.optim import FusedAdam
    w = torch.nn.Parameter(torch.randn(128, 256, device=DEV))
    b = torch.nn.Parameter(torch.randn(128, device=DEV))
    opt = FusedAdam([w, b], lr=1e-3)
    opt.packed_images[w] = torch.zeros(16, dtype=torch.uint8, device=DEV)
    w.grad, b.grad = torch.randn_like(w), torch.randn_like(b)
    w0, b0 = w.detach().clone(), b.detach().clone()
    with pytest.raises(ops.GradSinkMismatch, match='operand image'):
        opt.step_fused(ops.GradSink())
    torch.cuda.synchronize()
    assert torch.equal(w.detach(), w0) and torch.equal(b.detach(), b0) and int(opt.step_dev[0]) == 0
    # without an image the same gradients are an ordinary fused update
    opt.packed_images.clear()
    opt.step_fused(ops.GradSink())
    torch.cuda.synchronize()
    assert not torch.equal(w.detach(), w0) and int(opt.step_dev[0]) == 1
