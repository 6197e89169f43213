"""-m gpu: FC_output folded into the layer-1 relation parameters (kgw_fold_fwd / kgw_fold_bwd, kgw_linear_splitk_ind,
kgw_ind_colsum, the logit constant of kgw_gat_aggregate_fwd) -- the HIP kernels against the same algebra written with
framework ops in float64 (tests.helpers.fold_fc_output_reference + autograd), and the folded model against the unfolded one.  The folded
model is what every other parity test (oracle, committed vectors) runs."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import assert_close, fold_fc_output_reference

pytestmark = pytest.mark.gpu


def test_fold_kernels_match_the_framework_formulation():
    from kgwas_amd import ops
    from kgwas_amd.model import RelationPack
    g = torch.Generator().manual_seed(0)
    NR, C = 29, 128
    edge_types = [(f's{r % 3}', f'r{r}', f'd{(r * 2) % 3}') for r in range(NR)]
    rel_ids = [3, 4, 5, 9, 10, 11, 12, 0, 1, 2, 20, 21, 28]
    pack = RelationPack(edge_types, rel_ids, C).cuda()
    n = len(rel_ids)
    sm = np.array([r % 3 for r in rel_ids], dtype=np.int32)
    dm = np.array([(r * 2) % 3 for r in rel_ids], dtype=np.int32)
    tab = (np.asarray(rel_ids, dtype=np.int32), sm, dm)
    fc = []
    for m in range(3):
        fc += [(torch.randn(C, C, generator=g) * 0.1).cuda().requires_grad_(True), (torch.randn(C, generator=g) * 0.1).cuda().requires_grad_(True)]
    U = torch.randn(NR, C, generator=g).cuda().requires_grad_(True)
    V = torch.randn(NR, C, generator=g).cuda().requires_grad_(True)
    outs = ops.fold_fc_output_hip(pack, U, V, fc, tab)
    # float64 twin with framework ops
    p64 = RelationPack(edge_types, rel_ids, C).cuda().double()
    p64.load_state_dict({k: v.double() for k, v in pack.state_dict().items()})
    U64, V64 = U.detach().double().requires_grad_(True), V.detach().double().requires_grad_(True)
    fc64 = [t.detach().double().requires_grad_(True) for t in fc]
    T3 = torch.stack([fc64[2 * m].t() for m in range(3)])
    c3 = torch.stack([fc64[2 * m + 1] for m in range(3)])
    ref = fold_fc_output_reference(p64, U64, V64, T3, c3, torch.from_numpy(sm).long().cuda(), torch.from_numpy(dm).long().cuda())
    names = ('U', 'V', 'kappa', 'W', 'gamma')
    for nm, a, b in zip(names, outs, ref):
        assert_close(a, b, 1e-5, 1e-6, f'fold fwd {nm}', rel_to_max=2e-6)
    ws = [torch.randn(o.shape, generator=g).cuda() for o in outs]
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    sum((o * w.double()).sum() for o, w in zip(ref, ws)).backward()
    assert_close(pack.w_src_t.grad, p64.w_src_t.grad, 1e-5, 1e-6, 'd w_src_t', rel_to_max=2e-6)
    assert_close(U.grad, U64.grad, 1e-5, 1e-6, 'dU', rel_to_max=2e-6)
    assert_close(V.grad, V64.grad, 1e-5, 1e-6, 'dV', rel_to_max=2e-6)
    for m in range(3):
        assert_close(fc[2 * m].grad, fc64[2 * m].grad, 1e-5, 1e-6, f'd FC_output.weight {m}', rel_to_max=2e-6)
        assert_close(fc[2 * m + 1].grad, fc64[2 * m + 1].grad, 1e-5, 1e-6, f'd FC_output.bias {m}', rel_to_max=2e-6)
    # rows of relations outside the pack are zero
    outside = [r for r in range(NR) if r not in rel_ids]
    assert float(outs[0][outside].abs().max()) == 0.0 and float(outs[2][outside].abs().max()) == 0.0


@pytest.mark.parametrize('rows,R,real', [(1171, 17, None), (512, 6, None), (40, 2, None)])
def test_transform_with_segment_constant_and_its_backward(rows, R, real):
    from kgwas_amd import _lib
    g = torch.Generator().manual_seed(rows)
    C = 128
    X = torch.randn(rows, R * C, generator=g)
    W = torch.randn(R * C, C, generator=g) * 0.1
    b = torch.randn(C, generator=g)
    gamma = torch.randn(R, C, generator=g)
    den = (torch.rand(rows * R, generator=g) > 0.3).float() * (1.0 + torch.rand(rows * R, generator=g))
    stat = torch.stack([torch.randn(rows * R, generator=g), den], 1).contiguous()
    L = _lib.lib()
    Xc, Wc, bc, gc, sc = X.cuda(), W.cuda(), b.cuda(), gamma.cuda(), stat.cuda()
    Y = torch.empty(rows, C).cuda()
    nws = int(L.kgw_linear_splitk_workspace_floats(rows, R * C, C))
    ws = torch.empty(nws).cuda()
    _lib.check(L.kgw_linear_splitk_ind(Xc.data_ptr(), R * C, Wc.data_ptr(), C, bc.data_ptr(), Y.data_ptr(), C, rows, R * C, 1,
                                       sc.data_ptr(), gc.data_ptr(), ws.data_ptr(), nws, None, None), 'splitk_ind')
    ind = (den > 0).double().view(rows, R)
    ref = (X.double() @ W.double() + b.double() + ind @ gamma.double()).relu()
    assert_close(Y, ref, 1e-5, 1e-5, 'transform + segment constant', rel_to_max=2e-6)
    dY = torch.randn(rows, C, generator=g)
    dg = torch.empty(R, C).cuda()
    _lib.check(L.kgw_ind_colsum(sc.data_ptr(), dY.cuda().data_ptr(), C, rows, R, dg.data_ptr(), None), 'ind_colsum')
    assert_close(dg, ind.t() @ dY.double(), 1e-5, 1e-5, 'd gamma', rel_to_max=2e-6)


def test_folded_model_equals_unfolded_model(small_kg):
    """Same weights, same batch: prediction, loss and every parameter gradient of the folded path equal the unfolded
    path's (KGW_FOLD_FC=0) to fp32 summation-order tolerance."""
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.sampler import NeighborLoader
    res = {}
    ids = np.asarray(small_kg.train_input_nodes[1][:64])
    for fold in ('1', '0'):
        os.environ['KGW_FOLD_FC'] = fold
        try:
            run = KGWAS(small_kg, device='cuda:0', seed=3)
            run.initialize_model()
        finally:
            os.environ.pop('KGW_FOLD_FC', None)
        assert run.model.fold_fc == (fold == '1')
        with torch.no_grad():
            for pk in list(run.model.live_packs) + list(run.model.dead_packs):
                pk.bias.copy_(torch.randn(pk.bias.shape, generator=torch.Generator().manual_seed(1)) * 0.1)
        batch = next(iter(NeighborLoader(small_kg.data, [-1, -1], ('SNP', ids), batch_size=64, device='cuda:0')))
        run.model.train()
        loss, pred = run.model.forward_loss(batch.x_dict, batch.edge_index_dict, 64, batch.n_id('SNP'), batch.dg.y['SNP'],
                                            run._ld_weight_vector())
        loss.backward()
        res[fold] = (float(loss), pred.detach().cpu(), {k: (None if v is None else v.detach().cpu())
                                                        for k, v in run.model.named_reference_tensors(grad=True).items()})
    assert abs(res['1'][0] - res['0'][0]) <= 1e-5 * abs(res['0'][0])
    assert_close(res['1'][1], res['0'][1], 1e-4, 1e-5, 'pred')
    for k, g0 in res['0'][2].items():
        g1 = res['1'][2][k]
        assert (g0 is None) == (g1 is None), k
        if g0 is not None:
            assert_close(g1, g0, 1e-4, max(1e-5, 1e-4 * float(g0.abs().max())), f'grad {k}')


def test_duv_pieces_added_by_their_consumers():
    """KGW_F_DUV_PIECES: d u_r / d v_r handed to kgw_fold_bwd (layer 1) and kgw_relvec_bwd_multi (layer 2) as the eight level-1
    pieces the aggregate's riders leave ([rows][8][128]) instead of complete [rows][128] tensors -- both kernels add the pieces in
    k_duv_fold's order as they read: every output bit for bit what the complete tensors give."""
    import ctypes as C
    from kgwas_amd import _lib, ops
    from kgwas_amd.model import RelationPack
    g = torch.Generator().manual_seed(5)
    NR, Cc = 29, 128
    edge_types = [(f's{r % 3}', f'r{r}', f'd{(r * 2) % 3}') for r in range(NR)]
    rel_ids = [3, 4, 5, 9, 10, 11, 12, 0, 1, 2, 20, 21, 28]
    pack = RelationPack(edge_types, rel_ids, Cc).cuda()
    n = len(rel_ids)
    pieces = torch.randn(2, NR, 8, Cc, generator=g).cuda()
    p = pieces
    summed = ((p[:, :, 0] + p[:, :, 1]) + (p[:, :, 2] + p[:, :, 3])) + ((p[:, :, 4] + p[:, :, 5]) + (p[:, :, 6] + p[:, :, 7]))
    dU, dV = summed[0].contiguous(), summed[1].contiguous()
    # ---- kgw_fold_bwd
    sm = np.array([r % 3 for r in rel_ids], dtype=np.int32)
    dm = np.array([(r * 2) % 3 for r in rel_ids], dtype=np.int32)
    rid = np.asarray(rel_ids, dtype=np.int32)
    fc = [(torch.randn(Cc, Cc, generator=g) * 0.1).cuda() if k % 2 == 0 else (torch.randn(Cc, generator=g) * 0.1).cuda() for k in range(6)]
    U, V = torch.randn(NR, Cc, generator=g).cuda(), torch.randn(NR, Cc, generator=g).cuda()
    dkappa, dWp, dgamma = torch.randn(NR, generator=g).cuda(), torch.randn(n, Cc, Cc, generator=g).cuda(), torch.randn(n, Cc, generator=g).cuda()
    w = pack.w_src_t.detach()

    def fold_bwd(use_pieces):
        outs = [torch.full_like(U, float('nan')), torch.full_like(V, float('nan')), torch.full_like(w, float('nan'))] + \
               [torch.full_like(t, float('nan')) for t in fc]
        a = _lib.KgwFoldArgs()
        a.n, a.n_rels, a.n_mlp = n, NR, 3
        a.rel_ids_host, a.src_mlp_host, a.dst_mlp_host = rid.ctypes.data, sm.ctypes.data, dm.ctypes.data
        a.w_src_t, a.U, a.V = w.data_ptr(), U.data_ptr(), V.data_ptr()
        for m in range(3):
            a.fc_weight[m], a.fc_bias[m] = fc[2 * m].data_ptr(), fc[2 * m + 1].data_ptr()
            a.d_fc_weight[m], a.d_fc_bias[m] = outs[3 + 2 * m].data_ptr(), outs[4 + 2 * m].data_ptr()
        a.dkappa, a.dWp, a.dgamma = dkappa.data_ptr(), dWp.data_ptr(), dgamma.data_ptr()
        if use_pieces:
            a.dUp, a.dVp, a.duv_pieces = pieces[0].data_ptr(), pieces[1].data_ptr(), 1
        else:
            a.dUp, a.dVp = dU.data_ptr(), dV.data_ptr()
        a.dU, a.dV, a.dws = outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr()
        _lib.check(_lib.lib().kgw_fold_bwd(C.byref(a), _lib.stream_ptr()), 'kgw_fold_bwd')
        torch.cuda.synchronize()
        return outs
    for x, y in zip(fold_bwd(True), fold_bwd(False)):
        assert torch.equal(x, y) and not bool(torch.isnan(x).any())
    # ---- kgw_relvec_bwd_multi (one job)
    def relvec_bwd(use_pieces):
        outs = [torch.full_like(pack.w_src_t, float('nan')), torch.full_like(pack.w_dst_t, float('nan')),
                torch.full_like(pack.att_src, float('nan')), torch.full_like(pack.att_dst, float('nan'))]
        j = (_lib.KgwRelvecJob * 1)()
        j[0].n_live, j[0].rel_ids, j[0].bip_pos = pack.att_src.shape[0], pack.rel_ids_i32.data_ptr(), pack.bip_pos_i32.data_ptr()
        j[0].w_src_t, j[0].w_dst_t = pack.w_src_t.data_ptr(), (pack.w_dst_t.data_ptr() if pack.w_dst_t.numel() else None)
        j[0].att_src, j[0].att_dst = pack.att_src.data_ptr(), pack.att_dst.data_ptr()
        if use_pieces:
            j[0].dU_full, j[0].dV, j[0].duv_pieces = pieces[0].data_ptr(), pieces[1].data_ptr(), 1
        else:
            j[0].dU_full, j[0].dV = dU.data_ptr(), dV.data_ptr()
        j[0].dw_src_t, j[0].dw_dst_t = outs[0].data_ptr(), (outs[1].data_ptr() if outs[1].numel() else None)
        j[0].datt_src, j[0].datt_dst = outs[2].data_ptr(), outs[3].data_ptr()
        _lib.check(_lib.lib().kgw_relvec_bwd_multi(1, j, _lib.stream_ptr()), 'kgw_relvec_bwd_multi')
        torch.cuda.synchronize()
        return outs
    for x, y in zip(relvec_bwd(True), relvec_bwd(False)):
        assert torch.equal(x, y)
