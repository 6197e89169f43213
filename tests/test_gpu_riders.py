"""-m gpu: the step's parameter-only forward work as RIDER blocks of the first gene Linear's kgw_gemm3 launch (round 5,
kgwas_amd/csrc/kgw_riders.h, ops.ParamRiders): relation vectors of all layers + summed biases + the aggregates' zero fills
(kgw_relvec_fwd_multi) and the FC_output fold (kgw_fold_fwd), computed one wavefront per task on the compute units the product
leaves idle.  Same arithmetic as the stand-alone launches => every value downstream must be BIT-identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def wide_kg():
    from kgwas_amd.kgwas_data import KGWAS_Data
    # 4 607 genes x 1 024 features: the first gene Linear takes the resident kgw_gemm3 route (>= 4 096 rows, >= 512 wide)
    return KGWAS_Data.from_synthetic(scale=0.23, seed=2, feat_dims={'Gene': 1024}, data_path='/tmp/kgwas_synth_riders')


def _step(run, batch, ld_w, riders, monkeypatch):
    from kgwas_amd import ops
    monkeypatch.setattr(ops, '_G3_RIDERS', riders)
    m = run.model
    for p in m.parameters():
        p.grad = None
    loss, pred = m.forward_loss(batch.x_dict, batch.edge_index_dict, 256, batch.n_id('SNP'), batch.dg.y['SNP'], ld_w)
    loss.backward()
    torch.cuda.synchronize()
    return (loss.detach().clone(), pred.detach().clone(), {n: (p.grad.clone() if p.grad is not None else None) for n, p in m.named_parameters()},
            getattr(m, 'last_riders_taken', 0))


def test_riders_are_bit_identical_to_the_standalone_launches(wide_kg, monkeypatch):
    from kgwas_amd import _lib, ops
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.sampler import NeighborLoader
    run = KGWAS(wide_kg, device='cuda:0', seed=9)
    run.initialize_model()
    with torch.no_grad():                       # biases and attention vectors away from their symmetric initial values
        for pack in list(run.model.live_packs):
            pack.bias.normal_(0, 0.1)
        run.model.lin.bias.fill_(0.5)          # (keep the read-out's ReLU of model.py:86 alive: a dead one zeroes every gradient)
    ids = np.asarray(wide_kg.train_input_nodes[1][:256])
    batch = next(iter(NeighborLoader(wide_kg.data, [-1, -1], ('SNP', ids), batch_size=256, device='cuda:0')))
    assert 2 * batch.n_nodes['Gene'] > wide_kg.data['Gene'].x.shape[0], 'most genes must be in the batch for the resident route'
    ld_w = run._ld_weight_vector()
    run.model.train()
    g3 = ops.ROUTES.get('kgw_gemm3', 0)
    la, pa, ga, taken = _step(run, batch, ld_w, True, monkeypatch)
    assert ops.ROUTES.get('kgw_gemm3', 0) - g3 == 2, 'forward + weight gradient of the gene layer on kgw_gemm3'
    assert taken == 1, 'the parameter-only work must have ridden on the forward product'
    lb, pb, gb, taken_b = _step(run, batch, ld_w, False, monkeypatch)
    assert taken_b == 0
    assert torch.equal(la, lb) and torch.equal(pa, pb)
    assert float(pa.abs().max()) > 0
    for n in ga:
        assert (ga[n] is None) == (gb[n] is None), n
        if ga[n] is not None:
            assert torch.equal(ga[n], gb[n]), n


def test_riders_fall_back_when_no_product_takes_them(small_kg, monkeypatch):
    """Narrow gene features: no kgw_gemm3 launch in the forward -- the queue is flushed as the ordinary launches and the step is
    what it always was (the oracle tests run through this path on every small graph)."""
    from kgwas_amd import ops
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.sampler import NeighborLoader
    run = KGWAS(small_kg, device='cuda:0', seed=9)
    run.initialize_model()
    ids = np.asarray(small_kg.train_input_nodes[1][:64])
    batch = next(iter(NeighborLoader(small_kg.data, [-1, -1], ('SNP', ids), batch_size=64, device='cuda:0')))
    ld_w = run._ld_weight_vector()
    run.model.train()

    def go(riders):
        monkeypatch.setattr(ops, '_G3_RIDERS', riders)
        for p in run.model.parameters():
            p.grad = None
        loss, pred = run.model.forward_loss(batch.x_dict, batch.edge_index_dict, 64, batch.n_id('SNP'), batch.dg.y['SNP'], ld_w)
        loss.backward()
        return loss.detach().clone(), [p.grad.clone() for p in run.model.parameters() if p.grad is not None], run.model.last_riders_taken
    la, ga, ta = go(True)
    lb, gb, tb = go(False)
    assert ta == 0 and tb == 0
    assert torch.equal(la, lb) and len(ga) == len(gb) and all(torch.equal(a, b) for a, b in zip(ga, gb))


def test_captured_step_with_riders_equals_the_step_without(wide_kg, monkeypatch):
    """The captured single-GPU step with the riders on the gene layer's forward product: losses, parameters and running totals bit
    for bit those of the step that launches the relation vectors and the fold by themselves."""
    from kgwas_amd import ops
    from kgwas_amd.graph_step import GraphTrainStep
    from kgwas_amd.kgwas import KGWAS
    from tests.helpers import params_by_name
    ids = np.asarray(wide_kg.train_input_nodes[1][:256 * 6])
    outs = []
    sd0 = None
    for riders in (True, False):
        monkeypatch.setattr(ops, '_G3_RIDERS', riders)
        run = KGWAS(wide_kg, device='cuda:0', seed=13)
        run.initialize_model()
        if sd0 is None:
            with torch.no_grad():
                for pack in list(run.model.live_packs):
                    pack.bias.normal_(0, 0.1)
                run.model.lin.bias.fill_(0.5)
            import copy
            sd0 = copy.deepcopy(run.model.state_dict())
        else:
            run.model.load_state_dict(sd0)
        gs = GraphTrainStep(run, ('SNP', ids), 256, lr=1e-3, weight_decay=5e-4)
        assert gs.fused_adam
        run.model.train()
        losses = [float(gs.step(i)) for i in range(5)]
        totals = gs.check()
        assert gs.riders_taken == (1 if riders else 0), gs.riders_taken
        outs.append((losses, params_by_name(run.model), totals))
    assert outs[0][0] == outs[1][0] and outs[0][2] == outs[1][2]
    assert any(l > 0 for l in outs[0][0])
    for n in outs[0][1]:
        assert torch.equal(outs[0][1][n], outs[1][1][n]), n


def test_forward_rider_outputs_equal_the_standalone_kernels_tensor_by_tensor(wide_kg, monkeypatch):
    """u_r, v_r, summed biases of both layers and U', V', kappa, W', gamma of the fold: the rider blocks' values against
    kgw_relvec_fwd_multi / kgw_fold_fwd's, each tensor by name."""
    from kgwas_amd import ops
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.sampler import NeighborLoader
    run = KGWAS(wide_kg, device='cuda:0', seed=9)
    run.initialize_model()
    m = run.model
    with torch.no_grad():
        for pack in list(m.live_packs):
            pack.bias.normal_(0, 0.1)
        for mlp in (m.snp_feat_mlp, m.gene_feat_mlp, m.go_feat_mlp):
            mlp.FC_output.bias.normal_(0, 0.1)
    ids = np.asarray(wide_kg.train_input_nodes[1][:256])
    batch = next(iter(NeighborLoader(wide_kg.data, [-1, -1], ('SNP', ids), batch_size=256, device='cuda:0')))
    X = batch.dg.x['Gene']
    mlp = m.gene_feat_mlp
    names = ['tys', 'blocks', 'zws', 'U', 'V', 'bsum', 'Wv', 'kappa', 'Wp', 'gamma']

    def params(riders):
        q = ops.ParamRiders() if riders else None
        with torch.no_grad(), ops.param_riders_scope(q):
            prep = m._all_layer_params(batch, True)
            if riders:
                assert q.pending()
                ops.resident_first_linear(X, mlp.FC_hidden.weight, mlp.FC_hidden.bias)      # the launch that carries them
                assert not q.pending() and q.taken == 1
        torch.cuda.synchronize()
        return prep
    a, b = params(True), params(False)
    checked = 0
    for l, (pa, pb) in enumerate(zip(a, b)):
        for name, ta, tb in zip(names, pa, pb):
            if torch.is_tensor(ta):
                assert torch.equal(ta, tb), (l + 1, name, float((ta - tb).abs().max()))
                if name != 'zws':
                    assert float(ta.abs().max()) > 0 or name in ('kappa',), (l + 1, name)
                checked += 1
    assert checked >= 12


def test_operand_image_written_by_the_backward_kernel_equals_the_pack_launch(wide_kg, monkeypatch):
    """Round 5: the resident first layer's backward (kgw_mlp2_bwd_first_packed) writes d(pre-activation) as kgw_gemm3's B operand
    image itself -- same values, same three bf16 pieces, same sign periods as kgw_gemm3_pack over the fp32 rows: the weight
    gradient of FC_hidden (kgwas/model.py:13) and everything else must be bit-identical with and without the fused pack, eagerly
    and in the captured step's fused form (GradSink)."""
    from kgwas_amd import ops
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.optim import FusedAdam
    from kgwas_amd.sampler import NeighborLoader
    run = KGWAS(wide_kg, device='cuda:0', seed=9)
    run.initialize_model()
    m = run.model
    with torch.no_grad():
        for pack in list(m.live_packs):
            pack.bias.normal_(0, 0.1)
        m.lin.bias.fill_(0.5)
    ids = np.asarray(wide_kg.train_input_nodes[1][:256])
    batch = next(iter(NeighborLoader(wide_kg.data, [-1, -1], ('SNP', ids), batch_size=256, device='cuda:0')))
    ld_w = run._ld_weight_vector()
    m.train()

    def grads(fused_pack, with_sink):
        monkeypatch.setattr(ops, '_PACK_FUSED', fused_pack)
        for p in m.parameters():
            p.grad = None
        packs = ops.ROUTES.get('kgw_gemm3_pack', 0)
        loss, _ = m.forward_loss(batch.x_dict, batch.edge_index_dict, 256, batch.n_id('SNP'), batch.dg.y['SNP'], ld_w)
        sink = ops.GradSink() if with_sink else None
        with ops.grad_sink_scope(sink):
            loss.backward()
        if sink is not None:
            # (finish the deferred sums the way the captured step does, without moving the parameters: lr = 0)
            FusedAdam(m.parameters(), lr=0.0).step_fused(sink)
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    for with_sink in (False, True):
        ga, gb = grads(True, with_sink), grads(False, with_sink)
        assert ga.keys() == gb.keys()
        w1 = [n for n in ga if 'gene_feat_mlp.FC_hidden.weight' in n]
        assert w1 and float(ga[w1[0]].abs().max()) > 0
        for n in ga:
            assert torch.equal(ga[n], gb[n]), (with_sink, n, float((ga[n] - gb[n]).abs().max()))
