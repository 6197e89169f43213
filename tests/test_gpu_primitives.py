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
