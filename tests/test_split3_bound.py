"""The arithmetic kgw_gemm3 / k_mlp2_fwd3 / k_mlp2_bwd_first3 rest on (kgwas_amd/csrc/kgw_common.h: kgw_split3x8), restated in numpy
and checked on the CPU: every fp32 value is the EXACT sum of three bf16 pieces, and the six piece products the kernels keep differ
from the exact product by at most 3 * 2^-25 |a b| -- below the rounding of one fp32 multiply-add (2^-24 |a b|)."""
import numpy as np


def bf16_rne(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 does for finite inputs)."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x: np.ndarray):
    x = x.astype(np.float32)
    p1 = bf16_rne(x)
    r1 = (x - p1).astype(np.float32)
    p2 = bf16_rne(r1)
    r2 = (r1 - p2).astype(np.float32)
    p3 = bf16_rne(r2)
    return p1, p2, p3, r1, r2


def _samples(n, seed):
    g = np.random.default_rng(seed)
    mant = g.standard_normal(n).astype(np.float32)
    expo = g.integers(-60, 61, n)
    x = (mant * np.exp2(expo).astype(np.float32)).astype(np.float32)
    # plus values that stress the rounding: all-ones significands, ties, powers of two
    extra = np.array([1.0, -1.0, 1.9999999, 1.0039062, 1.0039063, 0.33333334, 3.4e38 / 4, 1e-30, 65504.0, 1.00390625 + 2 ** -16],
                     dtype=np.float32)
    return np.concatenate([x, extra])


def test_three_bf16_pieces_sum_to_the_fp32_value_exactly():
    x = _samples(200000, 1)
    p1, p2, p3, r1, r2 = split3(x)
    # the residuals the kernel forms in fp32 are exact (checked in float64), and the third piece loses nothing
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - p1.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - p2.astype(np.float64))
    assert np.array_equal(p3, r2)
    assert np.array_equal(p1.astype(np.float64) + p2.astype(np.float64) + p3.astype(np.float64), x.astype(np.float64))
    # piece magnitudes: |p2| <= 2^-8 |x|, |p3| <= 2^-16 |x| (round to nearest halves each)
    ax = np.abs(x.astype(np.float64))
    assert np.all(np.abs(p2.astype(np.float64)) <= ax * 2.0 ** -8)
    assert np.all(np.abs(p3.astype(np.float64)) <= ax * 2.0 ** -16)


def test_six_piece_products_are_within_the_bound_of_the_exact_product():
    a = _samples(100000, 2)[:100000]
    b = _samples(100000, 3)[:100000]
    pa = [p.astype(np.float64) for p in split3(a)[:3]]
    pb = [p.astype(np.float64) for p in split3(b)[:3]]
    # every bf16 x bf16 product has 16 significant bits: exact in fp32 (and in the float64 used here)
    kept = pa[0] * pb[0] + (pa[0] * pb[1] + pa[1] * pb[0]) + (pa[1] * pb[1] + pa[0] * pb[2] + pa[2] * pb[0])
    exact = a.astype(np.float64) * b.astype(np.float64)
    dropped = pa[1] * pb[2] + pa[2] * pb[1] + pa[2] * pb[2]
    scale = np.abs(exact)
    ok = scale > 0
    assert np.all(np.abs(exact - kept - dropped)[ok] <= scale[ok] * 2.0 ** -50)         # nothing else is lost
    assert np.all(np.abs(dropped)[ok] <= 3 * 2.0 ** -25 * scale[ok])                    # the kernels' truncation
    # and a product of two bf16 pieces really fits fp32
    q = (pa[0] * pb[0])
    assert np.array_equal(q.astype(np.float32).astype(np.float64)[np.abs(q) < 3e38], q[np.abs(q) < 3e38])
