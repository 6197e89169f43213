"""The whole training step as ONE HIP graph.

A step of kgwas/kgwas.py:129-151 is ~300 short launches; issued one by one the host, not the GPU, sets the
pace (5.1 ms of enqueue for 3.7 ms of kernels on MI355X).  With a *static layout* -- every per-batch row block
padded to a capacity measured over the loader's fixed batch order, actual counts kept on the device
(KgwBatchMeta) -- all buffer addresses and launch geometries are batch independent, so
sampling -> feature MLPs -> fused attention layers -> loss -> backward -> Adam is captured once and replayed
with a single launch per batch.  Padding rows carry exactly zero gradient (k_agg_bwd_src zero-fills them), so
parameter gradients equal the eager path's (tests/test_gpu_graph.py).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .sampler import BatchBuffers, NeighborLoader, SampledBatch, sample_into


class GraphTrainStep:
    def __init__(self, run, input_nodes, batch_size: int, lr: float = 1e-4, weight_decay: float = 5e-4,
                 margin: float = 1.03, capture_optimizer: bool = True):
        self.run = run
        self.model = run.model
        self.batch_size = int(batch_size)
        dev = torch.device(run.device)
        self.input_type, ids = input_nodes
        L = run.gnn_num_layers
        probe = NeighborLoader(run.data.data, [-1] * L, (self.input_type, ids), batch_size=self.batch_size,
                               drop_last=True, device=dev, prefetch=False)
        self.n_batches = len(probe)
        if self.n_batches == 0:
            raise ValueError('no full batch in input_nodes')
        self.caps = probe.measure_caps(margin)
        self.dg = probe.dg.with_static_caps(self.caps)
        self.seed_type = probe.seed_type
        self.ids = probe.ids
        self.buf = BatchBuffers(self.dg)
        self.meta = self.dg.static_meta()
        self.seeds = torch.zeros(self.batch_size, dtype=torch.int64, device=dev)
        self.ld_w = run._ld_weight_vector()
        self.capture_optimizer = capture_optimizer
        self.world = 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.world = dist.get_world_size()
        except Exception:
            pass
        if self.world > 1:
            capture_optimizer = False          # gradients are all-reduced between backward and Adam
            self.capture_optimizer = False
        from .optim import FusedAdam
        self.opt = FusedAdam(self.model.parameters(), lr=lr, weight_decay=weight_decay)   # one launch, capturable
        # device-side statistics accumulated inside the graph: [edges layer 1..L, sampled edges, error mask]
        nbytes = C.sizeof(_lib.KgwBatchMeta)
        self._meta_i32 = self.buf.meta.view(torch.int32)
        base = _lib.KgwBatchMeta
        self._idx = torch.tensor([base.n_edges.offset // 4 + l for l in range(L)] +
                                 [base.edge_end.offset // 4 + self.dg.n_hops - 1, base.error.offset // 4],
                                 dtype=torch.long, device=dev)
        self.stats = torch.zeros(L + 2, dtype=torch.int64, device=dev)
        self.loss = None
        self.graph = None
        self._capture()

    # one step on the current stream, static shapes only
    def _step_body(self):
        bs = self.batch_size
        sample_into(self.dg, self.buf, self.seeds, self.seed_type, record=False)
        batch = SampledBatch(self.dg, self.buf, self.meta, self.input_type, bs, static=True)
        self.opt.zero_grad(set_to_none=True)
        out = self.model(batch.x_dict, batch.edge_index_dict, bs)
        n_id = batch.n_id(self.input_type)[:bs].long()
        y = self.dg.y[self.input_type][n_id]
        w = self.ld_w[n_id]
        loss = torch.mean(w * (out.reshape(-1) - y) ** 2)            # float64, kgwas.py:145
        loss.backward()
        if self.capture_optimizer:
            self.opt.step()
        vals = self._meta_i32[self._idx].long()
        self.stats[:-1] += vals[:-1]
        self.stats[-1] |= vals[-1]
        return loss

    def _capture(self):
        # warm-up on a side stream (allocator pools, one-time kernel attributes, Adam state), then undo its effect
        params = [p for p in self.model.parameters()]
        snap = [p.detach().clone() for p in params]
        self.seeds.copy_(self.ids[:self.batch_size])
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                self._step_body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        err = int(self.buf.read_meta().error)
        if err:
            raise _lib.KgwasHipError(f'static layout does not fit the sampler buffers (error mask {err})')
        with torch.no_grad():
            for p, q in zip(params, snap):
                p.copy_(q)
            for st in self.opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
            self.opt.step_dev.zero_()
        self.stats.zero_()
        self.opt.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._step_body()

    def step(self, i: int):
        """Train on batch ``i`` of the loader's fixed order; returns the (device, float64) loss tensor."""
        b = self.batch_size
        self.seeds.copy_(self.ids[i * b:(i + 1) * b])
        self.graph.replay()
        if not self.capture_optimizer:
            if self.world > 1:
                from . import dist as kdist
                kdist.allreduce_grads(self.model, self.world)
            self.opt.step()
        return self.loss

    def grads_ready(self):
        return [p.grad for p in self.model.parameters() if p.grad is not None]

    def check(self):
        """Synchronise and verify that no batch overflowed the static capacities."""
        torch.cuda.synchronize()
        err = int(self.stats[-1])
        if err:
            raise _lib.KgwasHipError(f'a batch exceeded the static capacities (error mask {err}); raise `margin`')
        return [int(v) for v in self.stats[:-1]]
