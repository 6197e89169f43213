#!/usr/bin/env python
"""Does a side queue cost the captured step through RESIDENCY alone?  The step's heavy MFMA kernels (k_g3_gemm, k_mlp2_fwd3,
k_mlp2_bwd_first3, k_linear_wreg) run workgroups that need a WHOLE compute unit (two 256-register wavefronts per SIMD, or one of
512): such a workgroup cannot start on a CU while any other wavefront is resident there.  This tool replays the step with side
graphs of kernels that only stay resident (tools/micro/spin.hip: s_sleep, no memory traffic) -- spread over all CUs (256 blocks of
256 threads, the side sampler's shape) or confined to 16 (16 blocks of 1024 threads) -- next to the real sampler and to nothing.
usage: python tools/side_queue_blocking.py [steps]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kgwas_amd.graph_step import GraphTrainStep
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data

spin = C.CDLL(os.path.join(ROOT, 'tools', 'micro', 'libspin.so'))
spin.spin_launch.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_void_p]

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_synth_full_sq')
ids = np.asarray(data.train_input_nodes[1])
run = KGWAS(data, device='cuda:0', seed=1)
run.initialize_model()
gs = GraphTrainStep(run, ('SNP', ids), 512, lr=1e-4, weight_decay=5e-4)


def timed():
    for i in range(20):
        gs.step(i % gs.n_batches)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        gs.step((20 + i) % gs.n_batches)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


class Spin:
    """A side graph of ``launches`` kernels of ``blocks`` x ``threads`` that each stay resident for ``us`` microseconds."""
    def __init__(self, launches, blocks, threads, us):
        self.g = torch.cuda.CUDAGraph()
        self.args = (launches, blocks, threads, int(us * 100))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.body(s)
            torch.cuda.synchronize()
            with torch.cuda.graph(self.g, stream=s):
                self.body(s)
        torch.cuda.current_stream().wait_stream(s)

    def body(self, s):
        l, b, t, ticks = self.args
        for _ in range(l):
            rc = spin.spin_launch(b, t, ticks, C.c_void_p(s.cuda_stream))
            assert rc == 0, rc

    def replay(self):
        self.g.replay()


res = {}
res['real sampler'] = timed()
real = gs.sample_graphs
gs._skip_resample = True
res['nothing beside the step'] = timed()
gs._skip_resample = False
cases = {
    '30 x (256 blocks x 256 thr) resident 10 us': (30, 256, 256, 10),
    ' 8 x (256 blocks x 256 thr) resident 40 us': (8, 256, 256, 40),
    ' 8 x (1024 blocks x 256 thr) resident 40 us': (8, 1024, 256, 40),
    '30 x (16 blocks x 1024 thr) resident 10 us': (30, 16, 1024, 10),
    ' 8 x (16 blocks x 1024 thr) resident 40 us': (8, 16, 1024, 40),
    ' 8 x (32 blocks x 1024 thr) resident 40 us': (8, 32, 1024, 40),
    ' 8 x (64 blocks x 256 thr) resident 40 us': (8, 64, 256, 40),
}
for name, a in cases.items():
    d = Spin(*a)
    gs.sample_graphs = [d, d]
    res[name] = timed()
gs.sample_graphs = real
base = res['nothing beside the step']
for k, v in res.items():
    print('%-48s %.4f ms / step  (%+5.1f us)' % (k, v, (v - base) * 1e3))
