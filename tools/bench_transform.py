#!/usr/bin/env python
"""kgw_linear_splitk vs the library GEMM at the transform shapes of a 512-seed batch (GPU box): us per call."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kgwas_amd import ops  # noqa: E402


def t(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for rows, K, N, kn in [(1171, 2176, 128, True), (512, 768, 128, True), (1171, 128, 2176, False), (512, 128, 768, False),
                       (1216, 2176, 128, True), (2400, 2176, 128, True)]:
    X = torch.randn(rows, K, device='cuda')
    W = torch.randn(K, N, device='cuda') if kn else torch.randn(N, K, device='cuda')
    b = torch.randn(N, device='cuda')
    Wop = W if kn else W.t()
    g = torch.cuda.CUDAGraph()
    ops.linear(X, W, b, relu=True, w_kn=kn)
    with torch.cuda.graph(g):
        for _ in range(10):
            ops.linear(X, W, b, relu=True, w_kn=kn)
    own = t(lambda: ops.linear(X, W, b, relu=True, w_kn=kn))
    own_g = t(g.replay, 50) / 10
    lib = t(lambda: torch._addmm_activation(b, X, Wop))
    print(f'{rows:6d} x {K:5d} x {N:5d} kn={kn!s:5}: splitk {own:6.1f} us eager, {own_g:6.1f} us/launch-pair in a graph; library {lib:6.1f} us eager')
