"""-m gpu: the cross-lane reductions every aggregate kernel relies on (DPP butterflies +
v_permlane16_swap / v_permlane32_swap), checked lane by lane."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_half_and_wave_allsum():
    from kgwas_amd import _lib
    L = _lib.lib()
    x = torch.arange(64, dtype=torch.float32) ** 2 + 1.0          # asymmetric, exactly representable
    xd = x.cuda()
    oh, ow, st = torch.zeros(64).cuda(), torch.zeros(64).cuda(), torch.zeros(512).cuda()
    _lib.check(L.kgw_debug_reduce(xd.data_ptr(), oh.data_ptr(), ow.data_ptr(), st.data_ptr(), _lib.stream_ptr()), 'dbg')
    torch.cuda.synchronize()
    xn = x.numpy().astype(np.float64)
    st = st.cpu().numpy().reshape(8, 64)
    for k, width in enumerate((2, 4, 8, 16)):
        ref = xn.reshape(-1, width).sum(1).repeat(width)
        assert np.array_equal(st[k], ref), f'DPP stage {k} (groups of {width}): {st[k][:16]} vs {ref[:16]}'
    ref_half = np.concatenate([np.full(32, xn[:32].sum()), np.full(32, xn[32:].sum())])
    assert np.array_equal(oh.cpu().numpy(), ref_half), (oh.cpu().numpy(), ref_half)
    assert np.array_equal(ow.cpu().numpy(), np.full(64, xn.sum()))
    # raw semantics of the swap instructions the reductions are built on (operands a = lane, b = 100 + lane)
    lane = np.arange(64)
    a, b = lane, 100 + lane
    row = lane // 16
    assert np.array_equal(st[4], np.where(row % 2 == 0, a, b - 16))       # vdst': odd rows <- even rows of src
    assert np.array_equal(st[5], np.where(row % 2 == 0, a + 16, b))       # src' : even rows <- odd rows of vdst
    assert np.array_equal(st[6], np.where(lane < 32, a, b - 32))
    assert np.array_equal(st[7], np.where(lane < 32, a + 32, b))


def test_transposed_eight_way_reduction_primitives():
    """kgw_half_reduce8 & friends (kgw_common.h): eight reductions over a 32-lane half at once -- lane (hl & 7) == p ends
    with the half-wide total of value p; 8-lane-group max / sum / broadcast; the bank-masked DPP xor-4 / xor-8 moves."""
    from kgwas_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(12)
    x = torch.randn(64, 8, generator=g)
    xd = x.cuda().contiguous()
    out = torch.zeros(6, 64).cuda()
    _lib.check(L.kgw_debug_reduce8(xd.data_ptr(), out.data_ptr(), _lib.stream_ptr()), 'dbg8')
    out = out.cpu()
    lanes = torch.arange(64)
    half_sum = torch.stack([x[:32].double().sum(0), x[32:].double().sum(0)])            # [2, 8]
    want = half_sum[lanes // 32, lanes % 8]
    assert torch.allclose(out[0].double(), want, rtol=1e-5, atol=1e-5)
    v0 = x[:, 0]
    grp = v0.view(8, 8)
    assert torch.equal(out[1], grp.max(1).values.repeat_interleave(8))
    assert torch.allclose(out[2].double(), grp.double().sum(1).repeat_interleave(8), rtol=1e-5, atol=1e-6)
    assert torch.equal(out[3], grp[:, 3].repeat_interleave(8))
    assert torch.equal(out[4], v0[lanes ^ 4])
    assert torch.equal(out[5], v0[lanes ^ 8])
