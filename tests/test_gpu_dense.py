"""-m gpu: split-K MFMA TN GEMM (kgw_tn_gemm) and the fused MLP autograd nodes vs plain torch fp32/fp64.
fp32 MFMA is an exact fp32 FMA chain (guide 3); only the summation order differs from a library GEMM."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('rows,M,N', [(120001, 128, 128), (50000, 128, 20), (70000, 6, 128), (33333, 17, 128),
                                      (4097, 128, 128), (257, 128, 128), (1, 5, 7), (9000, 256, 128), (5000, 128, 300),
                                      (2176 * 3, 2176, 128)])
def test_tn_gemm_matches_fp64(rows, M, N):
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(rows)
    A = torch.randn(rows, M, generator=g)
    B = torch.randn(rows, N, generator=g)
    # asymmetric structure so a transposed / permuted output cannot pass
    A[:, 0] += 3.0
    B[:, -1] -= 2.0
    C, cs = ops.tn_gemm(A.cuda(), B.cuda(), colsum=True)
    ref = A.double().t() @ B.double()
    assert_close(C, ref, 1e-5, 1e-6, f'tn_gemm {rows}x{M}x{N}', rel_to_max=2e-6)
    assert_close(cs, A.double().sum(0), 1e-5, 1e-6, 'colsum', rel_to_max=2e-6)


@pytest.mark.parametrize('rows,M,N', [(122880, 128, 128), (20032, 128, 128), (7000, 128, 768), (1700, 128, 1408), (300, 64, 64)])
def test_tn_gemm_on_the_bf16_pipe_is_as_close_to_float64_as_on_the_fp32_pipe(rows, M, N):
    """Round 5: the 64 x 64-per-wavefront tiling (every weight-gradient product of the step but the narrow ones: dW = dY^T X of the
    MLPs' Linears, kgwas/model.py:13-21, and of the relation transforms, kgwas/conv.py:138) runs on the bf16 matrix pipe with each
    operand split exactly into three bf16 pieces (tn_rows_split3).  Against float64, operands spread over 2^-12 .. 2^12 so that
    every piece carries signal: the error relative to sum |a||b| must be no worse than 1.25 x the fp32 pipe's on the same data
    (kgw_tn_split(0)) and far below the K u bound; results are deterministic; the column sums are untouched."""
    from kgwas_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(rows + N)
    A = torch.randn(rows, M, device='cuda', generator=g) * torch.exp2(torch.randint(-12, 13, (rows, M), device='cuda', generator=g).float())
    B = torch.randn(rows, N, device='cuda', generator=g) * torch.exp2(torch.randint(-6, 7, (rows, N), device='cuda', generator=g).float())
    ref = A.double().t() @ B.double()
    sc = A.double().abs().t() @ B.double().abs()
    was = L.kgw_tn_split(1)
    try:
        C3, cs3 = ops.tn_gemm(A, B, colsum=True)
        C3b = ops.tn_gemm(A, B)
        L.kgw_tn_split(0)
        C32, cs32 = ops.tn_gemm(A, B, colsum=True)
    finally:
        L.kgw_tn_split(was)
    assert torch.equal(C3, C3b)
    e3 = ((C3.double() - ref).abs() / sc).max().item()
    e32 = ((C32.double() - ref).abs() / sc).max().item()
    print(f'[tn split] {rows}x{M}x{N}: max error / sum|a||b| bf16x3 {e3:.3e}, fp32 pipe {e32:.3e}')
    assert e3 <= max(1.25 * e32, 4 * 2.0 ** -24), (e3, e32)
    assert e3 <= 16 * 2.0 ** -24, e3
    assert float((C3 - C32).abs().max()) > 0 or rows < 64, 'the two pipes were not both exercised'
    assert_close(cs3, A.double().sum(0), 1e-5, 1e-6, 'colsum', rel_to_max=2e-6)


@pytest.mark.parametrize('rows', [122880, 20032])
def test_tn_gemm_on_the_bf16_pipe_has_no_one_sided_error(rows):
    """The bf16 MFMA's internal add truncates (a negative mean error, see tests/test_gpu_gemm3.py): a weight gradient with a
    one-sided error is what an optimiser integrates.  Odd wavefronts therefore accumulate the NEGATED product and are negated
    back (exact), so the means cancel inside every block.  Positive operands (every partial sum has one sign: the worst case), the
    step's tall shapes: the mean error must be within 2 x the fp32 pipe's own or a tenth of the mean absolute error, and the mean
    absolute error no worse than the fp32 pipe's."""
    from kgwas_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator(device='cuda').manual_seed(7)
    A = torch.rand(rows, 128, device='cuda', generator=g)
    B = torch.rand(rows, 128, device='cuda', generator=g) * (4.0 * 57.0 / rows)            # results ~57
    ref = A.double().t() @ B.double()
    was = L.kgw_tn_split(1)
    try:
        C3 = ops.tn_gemm(A, B)
        L.kgw_tn_split(0)
        C32 = ops.tn_gemm(A, B)
    finally:
        L.kgw_tn_split(was)
    e3, e32 = C3.double() - ref, C32.double() - ref
    m3, m32, a3, a32 = e3.mean().item(), e32.mean().item(), e3.abs().mean().item(), e32.abs().mean().item()
    print(f'[tn split bias] rows={rows}: mean error {m3:.3e} (fp32 pipe {m32:.3e}), mean |error| {a3:.3e} ({a32:.3e}), mean result {ref.mean().item():.1f}')
    assert abs(m3) <= max(2.0 * abs(m32), 0.1 * a3), (m3, m32)
    assert a3 <= 1.25 * a32, (a3, a32)


def test_tn_gemm_strided_inputs_and_determinism():
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(0)
    big = torch.randn(30000, 160, generator=g).cuda()
    A, B = big[:, :128], big[:, 32:160]           # row stride 160, not contiguous
    C1 = ops.tn_gemm(A, B)
    C2 = ops.tn_gemm(A, B)
    assert torch.equal(C1, C2)                    # fixed reduction order
    assert_close(C1, A.double().t() @ B.double(), 1e-5, 1e-6, 'strided', rel_to_max=2e-6)


def test_tn_gemm_transposed_output_and_repeated_colsum():
    """kgw_tn_gemm_ex: C^T written into a strided view, column sums replicated into q rows."""
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(4)
    for rows in (700, 40000):
        A = torch.randn(rows, 128, generator=g).cuda()
        B = torch.randn(rows, 384, generator=g).cuda()
        big = torch.full((5, 128, 128), 7.0).cuda()               # pack-like [n, k, c]; relations 1..3 are the target
        db = torch.full((5, 128), 7.0).cuda()
        ops.tn_gemm(A, B, out=big[1:4].view(384, 128), transpose_out=True, colsum_out=db[1:4])
        ref = B.double().t() @ A.double()
        assert_close(big[1:4].reshape(384, 128), ref, 1e-5, 1e-6, 'C^T', rel_to_max=2e-6)
        for q in (1, 2, 3):
            assert_close(db[q], A.double().sum(0), 1e-5, 1e-5, 'colsum copy', rel_to_max=2e-6)
        assert float(big[0].min()) == 7.0 and float(big[4].max()) == 7.0 and float(db[0].min()) == 7.0 and float(db[4].max()) == 7.0


def test_mlp_nodes_match_autograd():
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(1)
    n = 20000
    x = torch.randn(n, 20, generator=g).cuda()
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.1).cuda().requires_grad_(True)
    W1, b1, W2, b2, W3, b3 = mk(128, 20), mk(128), mk(128, 128), mk(128), mk(128, 128), mk(128)
    G = torch.randn(n, 128, generator=g).cuda()
    # rows with a pre-activation within fp32 rounding of zero have an ill-defined ReLU mask (fp32 kernel and fp64
    # reference may legitimately disagree on h > 0): give them no upstream gradient
    with torch.no_grad():
        z1 = x.double() @ W1.double().t() + b1.double()
        z2 = torch.relu(z1) @ W2.double().t() + b2.double()
        amb = (z1.abs() < 1e-5).any(1) | (z2.abs() < 1e-5).any(1)
        G[amb] = 0.0
    y = ops.mlp_tail(ops.linear_relu(x, W1, b1), W2, b2, W3, b3)
    (y * G).sum().backward()
    mine = [t.grad.clone() for t in (W1, b1, W2, b2, W3, b3)]
    ps = [t.detach().double().requires_grad_(True) for t in (W1, b1, W2, b2, W3, b3)]
    h = torch.relu(x.double() @ ps[0].t() + ps[1])
    h = torch.relu(h @ ps[2].t() + ps[3])
    yo = h @ ps[4].t() + ps[5]
    (yo * G.double()).sum().backward()
    assert_close(y, yo, 1e-5, 1e-6, 'mlp y')
    for a, b, n_ in zip(mine, ps, 'W1 b1 W2 b2 W3 b3'.split()):
        assert_close(a, b.grad, 1e-4, 1e-6, 'grad ' + n_)


@pytest.mark.parametrize('rows,K1,real', [(120003, 20, None), (16384, 20, None), (50001, 8, None), (40000, 20, 33333), (20000, 12, 0)])
def test_fused_two_layer_mlp_matches_fp64(rows, K1, real):
    """kgw_mlp2_fwd (ops.mlp2 on a narrow input): relu(relu(x W1^T + b1) W2^T + b2) in one launch, the hidden state kept for
    the backward; forward vs float64, gradients vs float64 autograd; whole / ragged last tiles; a static layout's padding
    rows (``rows_dev``) come out as zeros.  Tolerance: fp32 MFMA accumulation over K <= 128, rtol 1e-5 of the largest value."""
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(rows + K1)
    x = torch.randn(rows, K1, generator=g).cuda()
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.2).cuda().requires_grad_(True)
    W1, b1, W2, b2 = mk(128, K1), mk(128), mk(128, 128), mk(128)
    G = torch.randn(rows, 128, generator=g).cuda()
    rd = torch.tensor([real], dtype=torch.int32, device='cuda') if real is not None else None
    n = rows if real is None else real
    with torch.no_grad():
        z1 = x.double() @ W1.double().t() + b1.double()
        h1 = torch.relu(z1)
        z2 = h1 @ W2.double().t() + b2.double()
        G[(z1.abs() < 1e-5).any(1) | (z2.abs() < 1e-5).any(1)] = 0.0      # ill-defined ReLU masks: no upstream gradient
        G[n:] = 0.0
    h2 = ops.mlp2(x, W1, b1, W2, b2, rows_dev=rd)
    assert_close(h2[:n], torch.relu(z2)[:n], 1e-5, 1e-6, 'h2', rel_to_max=2e-6)
    assert float(h2.detach()[n:].abs().max() if n < rows else 0.0) == 0.0
    # the consumer hands back a gradient already multiplied by (h2 > 0) (see _MLP2)
    (h2 * G * (h2 > 0)).sum().backward()
    ps = [t.detach().double().requires_grad_(True) for t in (W1, b1, W2, b2)]
    ho = torch.relu(torch.relu(x.double()[:n] @ ps[0].t() + ps[1]) @ ps[2].t() + ps[3])
    (ho * G.double()[:n]).sum().backward()
    if n:
        for a, b, nm in zip((W1, b1, W2, b2), ps, 'W1 b1 W2 b2'.split()):
            assert_close(a.grad, b.grad, 1e-4, 1e-6, 'grad ' + nm, rel_to_max=1e-5)
    # the same with the row gather folded in: x = X[ids] of a larger resident matrix -- bit-identical to the sliced input
    Xres = torch.randn(rows + 1000, K1, generator=g).cuda()
    ids = torch.randint(0, rows + 1000, (rows,), generator=g, dtype=torch.int32).cuda()
    mine = [t.grad.clone() for t in (W1, b1, W2, b2)]
    for t in (W1, b1, W2, b2):
        t.grad = None
    ha = ops.mlp2(Xres[ids.long()].contiguous(), W1, b1, W2, b2, rows_dev=rd)
    (ha * G * (ha > 0)).sum().backward()
    ga = [t.grad.clone() for t in (W1, b1, W2, b2)]
    for t in (W1, b1, W2, b2):
        t.grad = None
    hb = ops.mlp2(Xres, W1, b1, W2, b2, rows_dev=rd, ids=ids)
    (hb * G * (hb > 0)).sum().backward()
    assert torch.equal(ha, hb)
    for t, ref in zip((W1, b1, W2, b2), ga):
        assert torch.equal(t.grad, ref)


@pytest.mark.parametrize('rows,K,N,kn', [(120003, 128, 128, False), (50001, 20, 128, False), (1000, 2176, 128, True),
                                         (777, 128, 128, True), (130, 768, 128, True), (5, 16, 128, False),
                                         (4097, 128, 20, True), (3000, 128, 260, False),
                                         # mid-size inputs: the 64-row-tile weight-resident variant
                                         (20033, 128, 128, False), (13001, 128, 128, True), (9000, 20, 128, False),
                                         (5000, 96, 64, False), (40001, 96, 100, True),
                                         # tall 128 x 128: the weights-in-registers variant (both weight layouts,
                                         # whole tiles only / a ragged last tile / fewer tiles than wavefronts)
                                         (120003, 128, 128, True), (32768, 128, 128, False), (40001, 128, 128, True),
                                         (262144 + 17, 128, 128, False),
                                         # up to 512 row tiles: one (tile, column half) per wavefront
                                         (16384, 128, 128, False), (4097, 128, 128, True), (16385, 128, 128, True)])
def test_linear_kernel_matches_fp64(rows, K, N, kn):
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(rows + K)
    X = torch.randn(rows, K, generator=g)
    W = torch.randn(*((K, N) if kn else (N, K)), generator=g) * 0.3
    b = torch.randn(N, generator=g)
    Mk = torch.randn(rows, N, generator=g)
    X[:, 0] += 2.0
    ref = X.double() @ (W.double() if kn else W.double().t()) + b.double()
    Y = ops.linear(X.cuda(), W.cuda(), b.cuda(), relu=False, w_kn=kn)
    assert_close(Y, ref, 1e-5, 1e-6, 'linear', rel_to_max=2e-6)
    Y2 = ops.linear(X.cuda(), W.cuda(), b.cuda(), relu=True, mask=Mk.cuda(), w_kn=kn)
    assert_close(Y2, torch.relu(ref) * (Mk.double() > 0), 1e-5, 1e-6, 'linear relu+mask', rel_to_max=2e-6)


def test_linear_act_node_matches_autograd():
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6000, 768, generator=g).cuda().requires_grad_(True)
    Wt = (torch.randn(768, 128, generator=g) * 0.05).cuda().requires_grad_(True)
    b = torch.randn(128, generator=g).cuda().requires_grad_(True)
    G = torch.randn(6000, 128, generator=g).cuda()
    y = ops.linear_act(x, Wt, b, True)
    (y * G).sum().backward()
    xo, Wo, bo = (t.detach().double().requires_grad_(True) for t in (x, Wt, b))
    yo = torch.relu(xo @ Wo + bo)
    (yo * G.double()).sum().backward()
    assert_close(y, yo, 1e-5, 1e-6, 'y')
    assert_close(x.grad, xo.grad, 1e-4, 1e-6, 'dx')
    assert_close(Wt.grad, Wo.grad, 1e-4, 1e-6, 'dWt')
    assert_close(b.grad, bo.grad, 1e-4, 1e-6, 'db')


def test_fused_adam_matches_torch_adam():
    from kgwas_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(2)
    shapes = [(23, 128, 128), (128,), (128, 20), (1, 128), (6, 128), (3, 5, 7)]
    pa = [torch.randn(*s, generator=g).cuda().requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    dead_a = torch.randn(4, 4).cuda().requires_grad_(True)          # never gets a gradient -> never touched
    dead_b = dead_a.detach().clone().requires_grad_(True)
    oa = FusedAdam(pa + [dead_a], lr=1e-3, weight_decay=5e-4)
    ob = torch.optim.Adam(pb + [dead_b], lr=1e-3, weight_decay=5e-4)
    for it in range(6):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 if it == 2 else 1.0)
            a.grad = gr.clone(); b.grad = gr.clone()
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        assert_close(a, b, 2e-6, 1e-7, 'adam param', rel_to_max=0)
    assert torch.equal(dead_a, dead_b) and int(oa.step_dev) == 6


def test_weighted_mse_matches_reference_expression():
    """kgw_wmse_fwd / _bwd vs the reference's expression, kgwas/kgwas.py:139-145 (float32 residual, float64 weights)."""
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(8)
    N, n = 5000, 512
    y_all = torch.rand(N, generator=g)
    w_all = torch.rand(N, generator=g, dtype=torch.float64) * 3
    n_id = torch.randperm(N, generator=g)[:700].to(torch.int32)          # longer than the batch: only the head is used
    pred = torch.rand(n, generator=g).cuda().requires_grad_(True)
    loss = ops.weighted_mse(pred, n_id.cuda(), y_all.cuda(), w_all.cuda())
    (loss * 1.7).backward()
    po = pred.detach().cpu().requires_grad_(True)
    ids = n_id[:n].long()
    lo = torch.mean(w_all[ids] * (po - y_all[ids]) ** 2)
    (lo * 1.7).backward()
    assert loss.dtype == torch.float64
    assert abs(float(loss) - float(lo)) <= 1e-12 + 1e-9 * abs(float(lo))
    assert_close(pred.grad, po.grad, 1e-6, 1e-9, 'd pred')


@pytest.mark.parametrize('rows,real', [(137000, 120017), (20000, 13001), (6000, 0), (3000, 2999)])
def test_rows_dev_skips_padding_rows(rows, real):
    """Static-capacity inputs: kgw_linear / kgw_tn_gemm_ex read the real row count from the device, compute only
    those rows, and kgw_linear writes zeros into the padding rows (garbage there must not leak anywhere)."""
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(rows)
    X = torch.randn(rows, 128, generator=g).cuda()
    X[real:] = float('nan')                                   # padding rows hold garbage
    W = (torch.randn(128, 128, generator=g) * 0.2).cuda(); b = torch.randn(128, generator=g).cuda()
    Mk = torch.randn(rows, 128, generator=g).cuda()
    cnt = torch.tensor([real], dtype=torch.int32).cuda()
    for kn in (False, True):
        Y = torch.full((rows, 128), 7.0).cuda()
        ops.linear(X, W, b, relu=True, mask=Mk, w_kn=kn, out=Y, rows_dev=cnt)
        ref = torch.relu(X[:real].double() @ (W.double() if kn else W.double().t()) + b.double()) * (Mk[:real].double() > 0)
        assert_close(Y[:real], ref, 1e-5, 1e-6, 'linear rows_dev', rel_to_max=2e-6)
        assert float(Y[real:].abs().sum()) == 0.0
    B = torch.randn(rows, 20, generator=g).cuda()
    Cm, cs = ops.tn_gemm(X, B, colsum=True, rows_dev=cnt)
    assert_close(Cm, X[:real].double().t() @ B[:real].double(), 1e-5, 1e-5, 'tn rows_dev', rel_to_max=2e-6)
    assert_close(cs, X[:real].double().sum(0), 1e-5, 1e-5, 'colsum rows_dev', rel_to_max=2e-6)


@pytest.mark.parametrize('n,rows,relu,h_is_relu', [(5, 9, True, False), (512, 512, True, True), (33, 40, False, True), (1, 1, True, False)])
def test_readout_weighted_mse_matches_autograd(n, rows, relu, h_is_relu):
    """kgw_readout_wmse_fwd / _bwd (read-out Linear(128->1) [+ReLU] + weighted MSE, kgwas/model.py:86 +
    kgwas/kgwas.py:139-145) vs the same expression in fp64 autograd; ragged sizes; optional folded ReLU mask on dH."""
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(n * 7 + rows)
    N = 300 + rows
    H = torch.randn(rows, 128, generator=g)
    if h_is_relu:
        H = torch.relu(H)
    wl = torch.randn(1, 128, generator=g) * 0.2; bl = torch.randn(1, generator=g)
    y_all = torch.rand(N, generator=g); w_all = torch.rand(N, generator=g, dtype=torch.float64) + 0.1
    n_id = torch.randperm(N, generator=g)[:rows].to(torch.int32)
    Hd, wd, bd = (t.cuda().requires_grad_(True) for t in (H, wl, bl))
    loss, pred = ops.readout_weighted_mse(Hd, wd, bd, n_id.cuda(), y_all.cuda(), w_all.cuda(), n, relu=relu, h_is_relu=h_is_relu)
    (loss * 0.7).backward()
    Ho, wo, bo = (t.double().requires_grad_(True) for t in (H, wl, bl))
    p = (Ho[:n] @ wo.t() + bo).reshape(-1)
    if relu:
        p = torch.relu(p)
    ids = n_id[:n].long()
    lo = torch.mean(w_all[ids] * (p - y_all[ids].double()) ** 2)
    (lo * 0.7).backward()
    assert_close(pred, p.detach(), 1e-5, 1e-6, 'pred')
    assert abs(float(loss) - float(lo)) <= 1e-9 + 2e-6 * abs(float(lo))
    ref_dH = Ho.grad * (H.double() > 0) if h_is_relu else Ho.grad          # the folded mask of the producing ReLU
    assert_close(Hd.grad, ref_dH, 1e-4, 1e-7, 'dH', rel_to_max=1e-5)
    assert float(Hd.grad[n:].abs().sum()) == 0.0
    assert_close(wd.grad, wo.grad, 1e-4, 1e-7, 'd lin.weight', rel_to_max=1e-5)
    assert_close(bd.grad, bo.grad, 1e-4, 1e-7, 'd lin.bias', rel_to_max=1e-5)


@pytest.mark.parametrize('rows,M,N', [(1, 128, 128), (63, 128, 20), (3, 6, 128), (257, 64, 128), (1000, 128, 2176)])
def test_tn_gemm_small_and_ragged_row_counts(rows, M, N):
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(rows + M + N)
    A = torch.randn(rows, M, generator=g).cuda(); B = torch.randn(rows, N, generator=g).cuda()
    C_, cs = ops.tn_gemm(A, B, colsum=True)
    assert_close(C_, A.double().t() @ B.double(), 1e-5, 1e-6, 'C', rel_to_max=2e-6)
    assert_close(cs, A.double().sum(0), 1e-5, 1e-6, 'colsum', rel_to_max=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('n_all,n_batch', [(20032, 13000), (1000, 1000), (37, 0), (9, 5)])
def test_scatter_relu_rows_matches_index_add(n_all, n_batch):
    """kgw_scatter_relu_rows == zeros.index_add(ids, g) * (h > 0) and its column sums (the backward of the resident
    first gene layer), bit for bit on the scatter and to fp32 rounding on the sums."""
    from kgwas_amd import _lib
    gen = torch.Generator().manual_seed(n_all)
    ids = torch.randperm(n_all, generator=gen)[:n_batch].to(torch.int32)
    g2l = torch.full((n_all,), -1, dtype=torch.int32)
    g2l[ids.long()] = torch.arange(n_batch, dtype=torch.int32)
    g = torch.randn(max(n_batch, 1), 128, generator=gen)[:n_batch].contiguous()
    h = torch.randn(n_all, 128, generator=gen)
    ref = torch.zeros(n_all, 128).index_add_(0, ids.long(), g) * (h > 0)
    gd, hd, ld = g.cuda(), h.cuda(), g2l.cuda()
    dz = torch.full((n_all, 128), 3.0).cuda(); cs = torch.empty(128).cuda()
    ws = torch.empty(int(_lib.lib().kgw_scatter_relu_rows_workspace_floats(n_all))).cuda()
    _lib.check(_lib.lib().kgw_scatter_relu_rows(gd.data_ptr() if n_batch else 0, ld.data_ptr(), hd.data_ptr(), n_all, dz.data_ptr(),
                                                cs.data_ptr(), ws.data_ptr(), _lib.stream_ptr()), 'scatter')
    assert torch.equal(dz.cpu(), ref)
    assert_close(cs, ref.double().sum(0), 1e-5, 1e-5, 'colsum', rel_to_max=2e-6)


@pytest.mark.gpu
def test_gather_rows_multi_and_narrow_rows():
    """kgw_gather_rows (flattened indexing: 20-float rows fill the lanes; odd widths take the scalar path) and the
    multi-job entry point equal index_select, including empty jobs."""
    from kgwas_amd.sampler import gather_rows, gather_rows_multi
    gen = torch.Generator().manual_seed(5)
    for w, N, n in [(20, 5000, 12345), (128, 300, 1000), (7, 50, 64), (5120, 40, 9)]:
        src = torch.randn(N, w, generator=gen).cuda()
        ids = torch.randint(0, N, (n,), generator=gen).to(torch.int32).cuda()
        assert torch.equal(gather_rows(src, ids), src.index_select(0, ids.long()))
    srcs = [torch.randn(N, 128, generator=gen).cuda() for N in (100, 7, 3000)]
    idss = [torch.randint(0, s.shape[0], (n,), generator=gen).to(torch.int32).cuda() for s, n in zip(srcs, (250, 0, 1111))]
    out = torch.full((250 + 0 + 1111, 128), -1.0).cuda()
    gather_rows_multi([(srcs[0], idss[0], out[:250]), (srcs[1], idss[1], out[250:250]), (srcs[2], idss[2], out[250:])])
    assert torch.equal(out, torch.cat([s.index_select(0, i.long()) for s, i in zip(srcs, idss)], 0))


@pytest.mark.gpu
@pytest.mark.parametrize('rows,real', [(50001, None), (20000, 13001), (3000, 2999)])
def test_grouped_weight_gradients_match_fp64(rows, real):
    """kgw_tn_gemm_multi: the three weight / bias gradients of an MLP (128x128, 128x128, 128x20) in one launch pair
    == the fp64 products, with and without a device-side row count; bitwise equal to the one-product entry point."""
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(rows)
    dYs = [torch.randn(rows, 128, generator=g).cuda() for _ in range(3)]
    Xs = [torch.randn(rows, 128, generator=g).cuda(), torch.randn(rows, 128, generator=g).cuda(), torch.randn(rows, 20, generator=g).cuda()]
    cnt = torch.tensor([real], dtype=torch.int32).cuda() if real is not None else None
    n = real if real is not None else rows
    if real is not None:
        for t in dYs + Xs:
            t[real:] = float('nan')
    outs = ops.weight_grads(list(zip(dYs, Xs)), rows_dev=cnt)
    for (dW, db), dY, X in zip(outs, dYs, Xs):
        assert_close(dW, dY[:n].double().t() @ X[:n].double(), 1e-5, 1e-5, 'grouped dW', rel_to_max=2e-6)
        assert_close(db, dY[:n].double().sum(0), 1e-5, 1e-5, 'grouped db', rel_to_max=2e-6)
    one = ops.tn_gemm(dYs[0], Xs[0], colsum=True, rows_dev=cnt)
    assert torch.equal(one[0], outs[0][0]) and torch.equal(one[1], outs[0][1])


@pytest.mark.parametrize('rows,K,N,kn,real', [(1171, 2176, 128, True, None), (1171, 1536, 128, True, None), (512, 768, 128, True, None),
                                              (1171, 128, 2176, False, None), (1171, 128, 1536, False, None), (512, 128, 768, False, None),
                                              (1, 128, 128, True, None), (33, 256, 128, False, None), (1216, 2176, 128, True, 1171),
                                              (1216, 128, 2176, False, 1171), (8000, 768, 128, True, None), (64, 128, 128, False, 0)])
def test_splitk_transform_kernel_matches_fp64(rows, K, N, kn, real):
    """kgw_linear_splitk at the benchmark's transform shapes ([~1.2 k, R*128] x [R*128, 128] forward, its dZ twin) and
    the corner cases: one row, ragged last tile, padding rows of a static layout (rows_dev), a zero-row batch."""
    from kgwas_amd import _lib, ops
    g = torch.Generator().manual_seed(rows + K)
    X = torch.randn(rows, K, generator=g)
    W = torch.randn(K, N, generator=g) if kn else torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    X[:, 0] += 2.0
    rd = None if real is None else torch.tensor([real], dtype=torch.int32).cuda()
    for relu, bias in ((True, b), (False, None)):
        Y = ops.linear(X.cuda(), W.cuda(), bias.cuda() if bias is not None else None, relu=relu, w_kn=kn, rows_dev=rd)
        ref = X.double() @ (W.double() if kn else W.double().t())
        if bias is not None:
            ref = ref + bias.double()
        if relu:
            ref = ref.relu()
        if real is not None:
            ref[real:] = 0.0
        assert_close(Y, ref, 1e-5, 1e-5, f'splitk {rows}x{K}x{N} kn={kn}', rel_to_max=2e-6)
    # deterministic (fixed slab order), writes into a strided output block in place
    out = torch.full((rows, N + 128), 7.0).cuda()
    Y1 = ops.linear(X.cuda(), W.cuda(), b.cuda(), relu=True, w_kn=kn, out=out[:, :N], rows_dev=rd)
    Y2 = ops.linear(X.cuda(), W.cuda(), b.cuda(), relu=True, w_kn=kn, rows_dev=rd)
    assert Y1.data_ptr() == out.data_ptr() and torch.equal(Y1, Y2) and bool((out[:, N:] == 7.0).all())
    # the C ABI refuses what the kernel does not take
    L = _lib.lib()
    assert L.kgw_linear_splitk(X.cuda().data_ptr(), K, W.cuda().data_ptr(), N if kn else K, None, Y2.data_ptr(), N, rows, K + 4, N,
                               0, 1 if kn else 0, None, 0, None, None) == -3


def test_fused_wide_gathered_mlp_matches_unfused():
    """kgw_mlp2w_fwd (ops.mlp2_gathered: three resident 128-wide matrices, rows by id, both hidden layers in one launch) vs
    gather + ops.mlp2 on the concatenated rows: forward within fp32 summation order (rtol 1e-5), gradients likewise; ragged
    job sizes, a last partial tile."""
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(9)
    Xs = [torch.randn(n, 128, generator=g).cuda() for n in (1192, 17248, 4512)]
    ids = [torch.randint(0, x.shape[0], (m,), generator=g, dtype=torch.int32).cuda() for x, m in zip(Xs, (377, 5003, 1201))]
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.1).cuda().requires_grad_(True)
    W1, b1, W2, b2 = mk(128, 128), mk(128), mk(128, 128), mk(128)
    jobs = list(zip(Xs, ids))
    assert ops.mlp2_gathered_ok(jobs, W1, W2)
    rows = sum(int(i.numel()) for i in ids)
    G = torch.randn(rows, 128, generator=g).cuda()
    ya = ops.mlp2_gathered(jobs, W1, b1, W2, b2)
    (ya * G * (ya > 0)).sum().backward()
    ga = [t.grad.clone() for t in (W1, b1, W2, b2)]
    for t in (W1, b1, W2, b2):
        t.grad = None
    x = torch.cat([X[i.long()] for X, i in jobs], 0)
    yo = torch.relu(torch.relu(x.double() @ W1.double().t() + b1.double()) @ W2.double().t() + b2.double())
    assert_close(ya, yo, 1e-5, 1e-6, 'h2', rel_to_max=2e-6)
    yb = ops.mlp2(x, W1, b1, W2, b2)
    (yb * G * (yb > 0)).sum().backward()
    for a, t, nm in zip(ga, (W1, b1, W2, b2), 'W1 b1 W2 b2'.split()):
        assert_close(a, t.grad, 1e-4, 1e-6, 'grad ' + nm, rel_to_max=1e-5)


def test_layer_transforms_of_all_destination_types_in_one_launch():
    """kgw_linear_splitk_multi / kgw_ind_colsum_multi (round 4): the forward transforms of a layer's destination types (genes
    [1171, 17 x 128], seed SNPs [512, 6 x 128] at the benchmark's shapes; a third, ragged one), their dZ twins and their d gamma
    sums, each group in ONE launch -- bit for bit what the per-type launches give (the per-element arithmetic is the same code),
    and the whole autograd node (ops.layer_transform) equal with the switch on and off."""
    import ctypes as C
    from kgwas_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(4)
    shapes = [(1171, 17), (512, 6), (77, 3)]
    Xs = [torch.randn(r, R * 128, generator=g).cuda() for r, R in shapes]
    Ws = [(torch.randn(R * 128, 128, generator=g) * 0.05).cuda() for r, R in shapes]
    bs = [torch.randn(128, generator=g).cuda() for _ in shapes]
    gam = [torch.randn(R, 128, generator=g).cuda() for r, R in shapes]
    stat = [torch.stack([torch.randn(r * R, generator=g), (torch.rand(r * R, generator=g) > 0.3).float()], 1).cuda() for r, R in shapes]
    st = _lib.stream_ptr()
    # forward with the per-segment constant: multi vs kgw_linear_splitk_ind per job
    Ym = [torch.empty(r, 128, device='cuda') for r, R in shapes]
    jobs = (_lib.KgwSplitKJob * 3)()
    for j, X, W, b, y, gm, s_, (r, R) in zip(jobs, Xs, Ws, bs, Ym, gam, stat, shapes):
        j.X, j.ldx, j.W, j.ldw, j.bias, j.Y, j.ldy, j.rows = X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0), b.data_ptr(), y.data_ptr(), 128, r
        j.K, j.N, j.relu, j.w_is_kn, j.seg_stat, j.gamma = R * 128, 128, 1, 1, s_.data_ptr(), gm.data_ptr()
    _lib.check(L.kgw_linear_splitk_multi(3, jobs, st), 'multi')
    for X, W, b, y, gm, s_, (r, R) in zip(Xs, Ws, bs, Ym, gam, stat, shapes):
        y1 = torch.empty(r, 128, device='cuda')
        _lib.check(L.kgw_linear_splitk_ind(X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0), b.data_ptr(), y1.data_ptr(), 128, r, R * 128, 1,
                                           s_.data_ptr(), gm.data_ptr(), None, 0, None, st), 'single')
        assert torch.equal(y, y1)
        ref = (X.double() @ W.double() + b.double() + ((s_[:, 1] > 0).double().view(r, R, 1) * gm.double().view(1, R, 128)).sum(1)).relu()
        assert_close(y, ref, 1e-5, 1e-5, 'multi transform vs fp64', rel_to_max=2e-6)
    # dZ twins: [rows, 128] x [128, R * 128] with W as [N, K]
    dys = [torch.randn(r, 128, generator=g).cuda() for r, R in shapes]
    dZm = [torch.empty(r, R * 128, device='cuda') for r, R in shapes]
    jobs = (_lib.KgwSplitKJob * 3)()
    for j, dy, W, dz, (r, R) in zip(jobs, dys, Ws, dZm, shapes):
        j.X, j.ldx, j.W, j.ldw, j.bias, j.Y, j.ldy, j.rows = dy.data_ptr(), 128, W.data_ptr(), W.stride(0), None, dz.data_ptr(), dz.stride(0), r
        j.K, j.N, j.relu, j.w_is_kn = 128, R * 128, 0, 0
    _lib.check(L.kgw_linear_splitk_multi(3, jobs, st), 'multi twin')
    for dy, W, dz in zip(dys, Ws, dZm):
        assert torch.equal(dz, ops.linear(dy, W))
    # d gamma sums
    dgm = [torch.empty(R, 128, device='cuda') for r, R in shapes]
    jobs = (_lib.KgwSplitKJob * 3)()
    for j, dy, s_, dg, (r, R) in zip(jobs, dys, stat, dgm, shapes):
        j.seg_stat, j.Y, j.ldy, j.rows, j.K, j.dgamma = s_.data_ptr(), dy.data_ptr(), 128, r, R * 128, dg.data_ptr()
    _lib.check(L.kgw_ind_colsum_multi(3, jobs, st), 'multi colsum')
    for dy, s_, dg, (r, R) in zip(dys, stat, dgm, shapes):
        d1 = torch.empty(R, 128, device='cuda')
        _lib.check(L.kgw_ind_colsum(s_.data_ptr(), dy.data_ptr(), 128, r, R, d1.data_ptr(), st), 'single colsum')
        assert torch.equal(dg, d1)
    # mixed kinds, too many jobs: refused before any launch
    jobs = (_lib.KgwSplitKJob * 2)()
    for j, X, W, y, (r, R) in zip(jobs, Xs, Ws, Ym, shapes):
        j.X, j.ldx, j.W, j.ldw, j.Y, j.ldy, j.rows, j.K, j.N, j.w_is_kn = X.data_ptr(), X.stride(0), W.data_ptr(), W.stride(0), y.data_ptr(), 128, r, R * 128, 128, 1
    jobs[1].K, jobs[1].N, jobs[1].w_is_kn = 128, 768, 0
    assert L.kgw_linear_splitk_multi(2, jobs, st) == -3
    assert L.kgw_linear_splitk_multi(5, jobs, st) == -2


@pytest.mark.parametrize('shapes', [[(1171, 17), (512, 6), (77, 3)], [(512, 6)], [(4000, 2), (33, 1)]])
@pytest.mark.parametrize('with_dz,with_gamma', [(True, True), (True, False), (False, True)])
def test_transform_backward_in_one_launch(shapes, with_dz, with_gamma):
    """kgw_transform_bwd: the weight / bias gradients (Z^T dY, column sums), the dZ twins (dY W^T) and the d gamma sums of a layer's
    destination types as blocks of ONE launch -- bit for bit what kgw_tn_gemm_multi + kgw_linear_splitk_multi + kgw_ind_colsum_multi
    give one after the other (the same code per block), and close to float64."""
    from kgwas_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(len(shapes) * 7 + shapes[0][0])
    st = _lib.stream_ptr()
    n = len(shapes)
    Zs = [torch.randn(r, R * 128, generator=g).cuda() for r, R in shapes]
    Ws = [(torch.randn(R * 128, 128, generator=g) * 0.05).cuda() for r, R in shapes]
    dys = [torch.randn(r, 128, generator=g).cuda() for r, R in shapes]
    stat = [torch.stack([torch.randn(r * R, generator=g), (torch.rand(r * R, generator=g) > 0.3).float()], 1).cuda() for r, R in shapes]

    def run(merged):
        dW = [torch.full((R * 128, 128), float('nan'), device='cuda') for r, R in shapes]
        db = [torch.full((R, 128), float('nan'), device='cuda') for r, R in shapes]
        dZ = [torch.full((r, R * 128), float('nan'), device='cuda') for r, R in shapes]
        dg = [torch.full((R, 128), float('nan'), device='cuda') for r, R in shapes]
        tn = (_lib.KgwTnJob * n)(); sk = (_lib.KgwSplitKJob * n)(); cs = (_lib.KgwSplitKJob * n)()
        keep = []
        for q, (r, R) in enumerate(shapes):
            nws = int(L.kgw_tn_gemm_workspace_floats(r, 128, R * 128))
            ws = torch.empty(nws, device='cuda'); keep.append(ws)
            j = tn[q]
            j.A, j.lda, j.B, j.ldb, j.rows = dys[q].data_ptr(), 128, Zs[q].data_ptr(), Zs[q].stride(0), r
            j.C, j.ldc, j.colsum_a, j.colsum_ld = dW[q].data_ptr(), 128, db[q].data_ptr(), 128
            j.workspace, j.workspace_floats, j.rows_dev = ws.data_ptr(), nws, None
            j.M, j.N, j.c_transposed, j.colsum_repeat = 128, R * 128, 1, R
            k = sk[q]
            k.X, k.ldx, k.W, k.ldw, k.bias, k.Y, k.ldy, k.rows = dys[q].data_ptr(), 128, Ws[q].data_ptr(), 128, None, dZ[q].data_ptr(), R * 128, r
            k.K, k.N, k.relu, k.w_is_kn = 128, R * 128, 0, 0
            c = cs[q]
            c.seg_stat, c.Y, c.ldy, c.rows, c.K, c.dgamma = stat[q].data_ptr(), dys[q].data_ptr(), 128, r, R * 128, dg[q].data_ptr()
        if merged:
            _lib.check(L.kgw_transform_bwd(n, tn, n if with_dz else 0, sk, n if with_gamma else 0, cs, st), 'kgw_transform_bwd')
        else:
            if n > 1:
                _lib.check(L.kgw_tn_gemm_multi(n, tn, st), 'tn multi')
            else:
                j = tn[0]
                _lib.check(L.kgw_tn_gemm_ex(j.A, j.lda, j.M, j.B, j.ldb, j.N, j.rows, j.C, j.ldc, 1, j.colsum_a, j.colsum_repeat, j.colsum_ld,
                                            j.workspace, j.workspace_floats, None, st), 'tn single')
            if with_dz:
                _lib.check(L.kgw_linear_splitk_multi(n, sk, st), 'twin multi')
            if with_gamma:
                _lib.check(L.kgw_ind_colsum_multi(n, cs, st), 'colsum multi')
        torch.cuda.synchronize()
        return dW, db, dZ, dg

    a, b = run(True), run(False)
    for q, (r, R) in enumerate(shapes):
        assert torch.equal(a[0][q], b[0][q]) and torch.equal(a[1][q], b[1][q])
        assert_close(a[0][q], Zs[q].double().t() @ dys[q].double(), 1e-4, 1e-5, 'dW^T vs fp64', rel_to_max=1e-5)
        assert_close(a[1][q], dys[q].double().sum(0).expand(R, 128), 1e-4, 1e-5, 'db vs fp64', rel_to_max=1e-5)
        if with_dz:
            assert torch.equal(a[2][q], b[2][q])
            assert_close(a[2][q], dys[q].double() @ Ws[q].double().t(), 1e-4, 1e-5, 'dZ vs fp64', rel_to_max=1e-5)
        else:
            assert bool(torch.isnan(a[2][q]).all())
        if with_gamma:
            assert torch.equal(a[3][q], b[3][q])
            ref = ((stat[q][:, 1] > 0).double().view(r, R, 1) * dys[q].double().view(r, 1, 128)).sum(0)
            assert_close(a[3][q], ref, 1e-4, 1e-5, 'd gamma vs fp64', rel_to_max=1e-5)
        else:
            assert bool(torch.isnan(a[3][q]).all())
    # argument checks before any launch
    assert L.kgw_transform_bwd(5, None, 0, None, 0, None, st) == -2
    assert L.kgw_transform_bwd(1, None, 0, None, 0, None, st) == -1
    assert L.kgw_transform_bwd(0, None, 0, None, 0, None, st) == 0


@pytest.mark.parametrize('deferred', [False, True])
@pytest.mark.parametrize('n,rows', [(512, 512), (500, 576), (1, 1), (5000, 5120)])
def test_readout_loss_training_node_matches_autograd(n, rows, deferred):
    """kgw_readout_wmse_train (unit loss gradient: forward + backward of the read-out + LD-weighted MSE node together, the
    blocks' partials + a fold launch) against fp64 autograd, ragged and large row counts.  ``deferred``: the fold launch left to
    the backward pass (ops.readout_fold_deferred, what a captured step does); otherwise the loss is complete after the forward."""
    import contextlib
    from kgwas_amd import ops
    g = torch.Generator().manual_seed(n + rows)
    N = 300 + rows
    H = torch.relu(torch.randn(rows, 128, generator=g))
    wl = torch.randn(1, 128, generator=g) * 0.2; bl = torch.randn(1, generator=g) + 0.5
    y_all = torch.rand(N, generator=g); w_all = torch.rand(N, generator=g, dtype=torch.float64) + 0.1
    n_id = torch.randperm(N, generator=g)[:rows].to(torch.int32)
    Hd, wd, bd = (t.cuda().requires_grad_(True) for t in (H, wl, bl))
    with (ops.readout_fold_deferred() if deferred else contextlib.nullcontext()):
        loss, pred = ops.readout_weighted_mse(Hd, wd, bd, n_id.cuda(), y_all.cuda(), w_all.cuda(), n, relu=True, h_is_relu=True, unit_grad=True)
    Ho, wo, bo = (t.double().requires_grad_(True) for t in (H, wl, bl))
    p = torch.relu((Ho[:n] @ wo.t() + bo).reshape(-1))
    ids = n_id[:n].long()
    lo = torch.mean(w_all[ids] * (p - y_all[ids].double()) ** 2)
    if not deferred:            # (an eager caller may read the loss before -- or without -- calling backward)
        assert abs(float(loss) - float(lo)) <= 1e-9 + 2e-6 * abs(float(lo))
    loss.backward(gradient=ops.unit_gradient(torch.device('cuda:0')))
    lo.backward()
    assert_close(pred, p.detach(), 1e-5, 1e-6, 'pred')
    assert abs(float(loss) - float(lo)) <= 1e-9 + 2e-6 * abs(float(lo))
    assert_close(Hd.grad, Ho.grad * (H.double() > 0), 1e-4, 1e-7, 'dH', rel_to_max=1e-5)
    assert float(Hd.grad[n:].abs().sum()) == 0.0
    assert_close(wd.grad, wo.grad, 1e-4, 1e-7, 'd lin.weight', rel_to_max=1e-5)
    assert_close(bd.grad, bo.grad, 1e-4, 1e-7, 'd lin.bias', rel_to_max=1e-5)
