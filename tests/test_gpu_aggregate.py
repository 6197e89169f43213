"""-m gpu: fused attention-aggregate kernels (forward, dst-major + src-major backward) against the
op-for-op restatement of kgwas/conv.py:200-228 (oracle.pyg_semantics.segment_softmax + index_add).

Tolerance: fp32 kernels vs an fp64 oracle on the same inputs: rtol 1e-4, atol 1e-5 --
summation order over up-to-1500-edge rows differs from the oracle's sequential index_add."""
import numpy as np
import pytest
import torch

from oracle.pyg_semantics import segment_softmax
from tests.helpers import assert_close

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5


def _batch(data, L, n_seeds=48, seed=0, full=False):
    from kgwas_amd.sampler import NeighborLoader, sample_full_graph
    if full:
        return sample_full_graph(data, L, 'cuda:0')
    ids = np.random.default_rng(seed).choice(data['SNP'].x.shape[0], size=n_seeds, replace=False)
    return next(iter(NeighborLoader(data, [-1] * L, ('SNP', ids), batch_size=n_seeds, device='cuda:0')))


def _oracle_layer(batch, layer, H, V, U, slope=0.2, temp=1.0):
    """fp64 CPU: Z[zrow] = sum_j softmax_j(leaky_relu(<H_s[j],u_r> + <H_d[i],v_r>) / T) H_s[j]  (conv.py:144-228 with
    the attention projections re-associated: a_s = <x W_src^T, att_src> = <x, W_src^T att_src>)."""
    dg, m = batch.dg, batch.meta
    sc = dg.schema
    ei = {k: v.cpu() for k, v in batch.edge_index_dict.items()}
    z_rows = int(m.z_base[layer - 1][sc.NT])
    Z = torch.zeros(z_rows, 128, dtype=torch.float64)
    for r, et in enumerate(sc.edge_types):
        if not dg.kg.rel_live[layer - 1][r]:
            continue
        s, d = int(sc.src_type[r]), int(sc.dst_type[r])
        nr = int(m.n_rows[layer - 1][d])
        e = ei[et]
        keep = e[1] < nr                               # the layer only aggregates into hops <= L - l
        src, dst = e[0][keep], e[1][keep]
        if src.numel() == 0:
            continue
        Hs = H[int(m.src_base[layer - 1][s]):int(m.src_base[layer - 1][s]) + int(m.n_src[layer - 1][s])]
        zrow = int(m.z_base[layer - 1][d]) + dst * int(sc.R_dst[d]) + int(sc.slot_dst[r])
        a_s = Hs @ U[r]
        Hd = H[int(m.src_base[layer - 1][d]):int(m.src_base[layer - 1][d]) + nr]     # destination rows: head of their type's block
        a_d = Hd @ V[r]
        logit = torch.nn.functional.leaky_relu(a_s[src] + a_d[dst], slope)
        # softmax grouped by destination row of THIS relation
        alpha = segment_softmax(logit / temp, dst, nr)
        Z.index_add_(0, zrow, alpha.unsqueeze(-1) * Hs[src])
    return Z


@pytest.mark.parametrize('graph', ['small', 'edge', 'edge_full'])
@pytest.mark.parametrize('layer', [1, 2])
def test_aggregate_forward_backward(small_kg, edge_case_graph, graph, layer):
    from kgwas_amd import ops
    data = small_kg.data if graph == 'small' else edge_case_graph[0]
    batch = _batch(data, 2, full=(graph == 'edge_full'))
    dg, m = batch.dg, batch.meta
    sc = dg.schema
    g = torch.Generator().manual_seed(layer)
    n_src = int(m.src_base[layer - 1][sc.NT])
    z_rows = int(m.z_base[layer - 1][sc.NT])
    assert n_src > 0 and z_rows > 0
    H = torch.randn(n_src, 128, generator=g)
    V = torch.randn(sc.NR, 128, generator=g) * 0.2
    U = torch.randn(sc.NR, 128, generator=g) * 0.2
    G = torch.randn(z_rows, 128, generator=g)

    Hd, Vd, Ud = (t.cuda().requires_grad_(True) for t in (H, V, U))
    Z, stat, e_edge = ops.gat_aggregate(batch, layer, Hd, Ud, Vd)
    (Z * G.cuda()).sum().backward()

    Ho, Vo, Uo = (t.double().requires_grad_(True) for t in (H, V, U))
    Zo = _oracle_layer(batch, layer, Ho, Vo, Uo)
    (Zo * G.double()).sum().backward()

    assert_close(Z, Zo.detach(), RTOL, ATOL, 'Z')
    assert_close(Hd.grad, Ho.grad, RTOL, 2e-5, 'dH')
    live = [r for r in range(sc.NR) if dg.kg.rel_live[layer - 1][r]]
    assert_close(Ud.grad[live], Uo.grad[live], RTOL, 1e-4, 'dU')
    assert_close(Vd.grad[live], Vo.grad[live], RTOL, 1e-4, 'dV')
    dead = [r for r in range(sc.NR) if not dg.kg.rel_live[layer - 1][r]]
    assert float(Ud.grad[dead].abs().sum()) == 0.0 and float(Vd.grad[dead].abs().sum()) == 0.0
    # softmax rows sum to one wherever a row has edges
    alpha = ops.edge_alpha(batch, layer, stat, e_edge)
    assert torch.isfinite(alpha).all() and float(alpha.min()) >= 0.0


def test_temperature_and_slope(edge_case_graph):
    from kgwas_amd import ops
    data = edge_case_graph[0]
    batch = _batch(data, 2)
    m, sc = batch.meta, batch.dg.schema
    g = torch.Generator().manual_seed(3)
    H = torch.randn(int(m.src_base[0][sc.NT]), 128, generator=g)
    V = torch.randn(sc.NR, 128, generator=g) * 0.2
    U = torch.randn(sc.NR, 128, generator=g) * 0.2
    Z, _, _ = ops.gat_aggregate(batch, 1, H.cuda(), U.cuda(), V.cuda(), neg_slope=0.05, temperature=2.5)
    Zo = _oracle_layer(batch, 1, H.double(), V.double(), U.double(), slope=0.05, temp=2.5)
    assert_close(Z, Zo, RTOL, ATOL, 'Z (T=2.5, slope=0.05)')


def test_extreme_logits_do_not_overflow(edge_case_graph):
    """Rare branch (guide rule 26): a spike inside a long row forces the online-softmax rescale."""
    from kgwas_amd import ops
    data = edge_case_graph[0]
    batch = _batch(data, 2, full=True)
    m, sc = batch.meta, batch.dg.schema
    g = torch.Generator().manual_seed(9)
    n_src = int(m.src_base[0][sc.NT])
    H = torch.randn(n_src, 128, generator=g)
    H[777] *= 40.0                      # one SNP source row with a huge logit inside gene 0's 1500-edge hub row
    V = torch.randn(sc.NR, 128, generator=g) * 0.1
    U = torch.randn(sc.NR, 128, generator=g)
    Z, stat, e = ops.gat_aggregate(batch, 1, H.cuda(), U.cuda(), V.cuda())
    assert torch.isfinite(Z).all()
    Zo = _oracle_layer(batch, 1, H.double(), V.double(), U.double())
    assert_close(Z, Zo, 2e-4, 1e-4, 'Z with spike')


@pytest.mark.parametrize('relu_input', [False, True])
def test_backward_short_row_path_equals_general_path(small_kg, monkeypatch, relu_input):
    """The src-major backward's 8-rows-per-wavefront path (octets the sampler flags: short rows of a short-row type) and
    the general path compute the same dH / dU / dV: same batch, once with the octet path switched off.  They differ in
    summation order only (the a_src term is added per entry instead of per relation slot): rtol 1e-5."""
    from kgwas_amd import ops
    batch = _batch(small_kg.data, 2, n_seeds=64, seed=3)
    dg, m = batch.dg, batch.meta
    sc = dg.schema
    layer = 1
    n_src = int(m.src_base[layer - 1][sc.NT])
    z_rows = int(m.z_base[layer - 1][sc.NT])
    flags = batch.buf.t_cnt[layer - 1][:(n_src + 7) // 8].cpu().numpy()
    assert dg.short_type_mask & 1 and flags.sum() > 10, 'the SNP rows of this graph should take the octet path'
    g = torch.Generator().manual_seed(5)
    H = torch.randn(n_src, 128, generator=g)
    if relu_input:
        H = torch.relu(H)
    V = torch.randn(sc.NR, 128, generator=g) * 0.2
    U = torch.randn(sc.NR, 128, generator=g) * 0.2
    G = torch.randn(z_rows, 128, generator=g).cuda()
    out = []
    for short in (True, False):
        monkeypatch.setattr(ops, '_SHORT_ROWS', short)
        Hd, Vd, Ud = (t.cuda().requires_grad_(True) for t in (H, V, U))
        Z, _, _ = ops.gat_aggregate(batch, layer, Hd, Ud, Vd, relu_input=relu_input)
        (Z * G).sum().backward()
        out.append((Hd.grad.clone(), Ud.grad.clone(), Vd.grad.clone()))
    for a, b, name in zip(out[0], out[1], ('dH', 'dU', 'dV')):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-5 * scale + 1e-7, name
