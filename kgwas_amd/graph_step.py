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
import os
import sys

import numpy as np
import torch

from . import _lib, ops
from .sampler import BatchBuffers, NeighborLoader, SampledBatch, sample_into


# Blocks per launch of a sampler that runs beside a step graph.  Round 6: 1 024 (256 until then).  What the side sampler costs the
# step is not its length but how long each of ITS kernels stays resident: the step's heavy MFMA kernels (k_g3_gemm, k_mlp2_fwd3,
# k_mlp2_bwd_first3, k_linear_wreg) run workgroups that need a WHOLE compute unit's registers, so one of them cannot start on a CU
# while any sampler wavefront sits there, and waits for the sampler kernel in flight to drain (tools/side_queue_blocking.py:
# side kernels that only stay resident -- no memory traffic -- cost the step +1 us at 30 x 10 us, +24 us at 8 x 40 us over 256 blocks,
# +43 us over 1 024).  More blocks = shorter sampler kernels: 1.060 / 1.045 / 1.041 / 1.051 ms per step at 256 / 512 / 1 024 / 2 048
# blocks on one box, 1.068 / 1.135 / 1.48 at 128 / 64 / 32 (profiles/r6/r6_sampler_geometry.txt).
SIDE_SAMPLER_GRID = int(os.environ.get('KGW_SIDE_SAMPLER_GRID', '1024'))       # (the knob: re-measured whenever the sampler changes)
def side_stream(device):
    """The stream the next batch's sampler graph is captured on and replayed on: an ordinary stream.  (Round 4 measured a CU-masked
    stream -- the two graphs then ran strictly one after the other --, stream priorities and a timed offset at the head of the
    sampler's graph: none helped, profiles/r4/r4_sampler_interference_experiments.txt; the knobs are gone.)"""
    return torch.cuda.Stream(device=device)


class BatchCache:
    """The sampled batches of epoch 1, kept in HBM and put back in later epochs instead of sampling them again.

    The reference's training loader has a fixed batch order (``NeighborLoader`` without shuffle, kgwas/kgwas.py:93-101), so every
    epoch asks the sampler for exactly the structures of the first -- ~0.26 ms of sampler kernels per step that cost the step
    ~50-75 us beside it.  After a batch is sampled (epoch 1) ONE launch on the sampler's stream copies the arrays the step reads --
    node lists, relabelled columns, chunk records, the src-major structures, the batch's counts: ~20 MB at the benchmark's shape,
    sized by the trainer's static capacities -- into the batch's slot of a resident cache (kgw_segments_copy); from epoch 2 on one
    launch copies the slot back into the idle batch buffer where the sampler's ~25 launches ran.  Same bytes in the same
    buffers => the step's kernels produce the same bits as after live sampling (tests/test_gpu_graph.py).  19 GB for the
    956 batches of the benchmark; a cache that would take more than ``max_fraction`` of the free HBM is not made."""

    def __init__(self, dg, bufs, n_batches: int, max_fraction: float = 0.5):
        self.n_batches = int(n_batches)
        dev = dg.device
        sc, L, H, caps, g = dg.schema, dg.num_layers, dg.n_hops, dg.caps, dg.kg
        segs = []                                  # (attribute, layer or None, first byte, bytes)

        def add(name, layer, first_elem, n_elem, width=4):
            segs.append((name, layer, int(first_elem) * width, int(n_elem) * width))
        add('meta', None, 0, C.sizeof(_lib.KgwBatchMeta), 1)
        n_through = lambda t, h: int(caps.node_off[t][h])          # nodes of type t through hop h - 1
        for t in range(sc.NT):
            n = n_through(t, H + 1)
            if n:
                add('n_id', None, dg.node_base[t], n)
            if n and 2 * n > dg.n_nodes[t]:                        # (a type the forward may address by global id: model._features)
                add('g2l', None, dg.node_base[t], dg.n_nodes[t])
        n_seg = sum(n_through(int(g.rel_dst[r]), H) for r in range(sc.NR))
        add('seg_ptr', None, 0, n_seg + 2)
        add('seg_chptr', None, 0, n_seg + 2)
        add('col_local', None, 0, int(caps.edges[0]) + 1)
        add('chunks', None, 0, (int(caps.chunks[0]) + 1) * 8)
        for h in range(H):
            n_multi = min(int(dg.multi_cap), sum(n_through(int(g.rel_dst[r]), h + 1) for r in range(sc.NR)))
            add('multi', None, h * int(dg.multi_cap) * 4, n_multi * 4)
        for l in range(1, L + 1):
            rows = sum(int(g.cap_src[l - 1][t]) for t in range(sc.NT))
            trows = sum(int(g.cap_src[l - 1][t]) * int(g.R_src[t]) for t in range(sc.NT))
            e = int(caps.edges[l - 1]) + 1
            add('t_ptr', l, 0, trows + 2)
            add('t_cnt', l, 0, (rows + 7) // 8 + 1)                # (the octet flags of the backward's short-row path)
            add('t_edge', l, 0, e)
            add('t_zrow', l, 0, e)
            add('t_rel', l, 0, e, 1)
        if len(segs) > _lib.KGW_SEGCOPY_MAX:
            raise ValueError(f'{len(segs)} arrays per batch: more than kgw_segments_copy takes')
        off, self.plans, self.slot_off = 0, [], []
        units = []
        for name, layer, first, nbytes in segs:
            t0 = getattr(bufs[0], name)
            t0 = t0[layer - 1] if layer is not None else t0
            room = t0.numel() * t0.element_size() - first
            u = min(-(-nbytes // 16), room // 16)
            if first % 16 or u * 16 < nbytes:
                raise ValueError(f'batch array {name} is not laid out in whole 16-byte units')
            units.append(u)
            self.slot_off.append(off)
            off += u * 16
        self.slot_bytes = off
        total = self.slot_bytes * self.n_batches
        free = torch.cuda.mem_get_info(dev)[0]
        if total > max_fraction * free:
            raise MemoryError(f'batch cache of {total / 2**30:.1f} GiB against {free / 2**30:.1f} GiB free')
        self.slots = torch.empty(total, dtype=torch.uint8, device=dev)
        self.index = torch.zeros(1, dtype=torch.int64, device=dev)         # the slot of the launch being replayed
        self._all = torch.arange(self.n_batches, dtype=torch.int64, device=dev)
        self.filled = [False] * self.n_batches
        self.grid = bufs[0].c.grid_blocks
        for buf in bufs:
            pair = []
            for to_slot in (1, 0):
                p = _lib.KgwSegCopy()
                p.n, p.to_slot = len(segs), to_slot
                for j, ((name, layer, first, _), u) in enumerate(zip(segs, units)):
                    t = getattr(buf, name)
                    t = t[layer - 1] if layer is not None else t
                    p.ptr[j], p.units[j], p.slot_off[j] = t.data_ptr() + first, u, self.slot_off[j]
                p.slots, p.slot_stride, p.slot_index = self.slots.data_ptr(), self.slot_bytes, self.index.data_ptr()
                pair.append(p)
            self.plans.append(pair)
        self.saved = self.restored = 0

    def _launch(self, which: int, to_slot: bool, i: int):
        self.index.copy_(self._all[i:i + 1], non_blocking=True)
        _lib.check(_lib.lib().kgw_segments_copy(C.byref(self.plans[which][0 if to_slot else 1]), self.grid, _lib.stream_ptr()),
                   'kgw_segments_copy')

    def save(self, which: int, i: int):
        """Buffer ``which`` holds batch ``i``, freshly sampled on the CURRENT stream: keep it."""
        self._launch(which, True, i)
        self.filled[i] = True
        self.saved += 1

    def restore(self, which: int, i: int):
        """Batch ``i`` back into buffer ``which`` (on the current stream)."""
        assert self.filled[i]
        self._launch(which, False, i)
        self.restored += 1


class GraphTrainStep:
    def __init__(self, run, input_nodes, batch_size: int, lr: float = 1e-4, weight_decay: float = 5e-4,
                 margin: float = 1.03, capture_optimizer: bool = True, overlap_sampling: bool = None,
                 shard_gene_layer: bool = None, cache_batches: bool = False):
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
        # two batch buffers: while the graph trains on one, its side branch samples the NEXT batch into the other
        self.bufs = [BatchBuffers(self.dg, SIDE_SAMPLER_GRID), BatchBuffers(self.dg, SIDE_SAMPLER_GRID)]
        self.buf = self.bufs[0]
        self.meta = self.dg.static_meta()
        self.seeds = torch.zeros(self.batch_size, dtype=torch.int64, device=dev)        # seeds sampled by the side branch
        self.ld_w = run._ld_weight_vector()
        self.capture_optimizer = capture_optimizer
        self.world = 1
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                self.world = dist.get_world_size()
        except Exception:
            pass
        # (KGW_FORCE_MULTIRANK_PATH=1: a single rank takes the multi-rank path -- split backward, RCCL all-reduce of one rank on
        # the side stream, eager Adam -- so that a 1-GPU box exercises exactly what an 8-GPU node runs)
        self._multi = self.world > 1 or (os.environ.get('KGW_FORCE_MULTIRANK_PATH') == '1' and torch.distributed.is_initialized())
        if self._multi:
            self.capture_optimizer = False       # gradients are all-reduced between backward and Adam
        # multi-rank: the backward pass is captured in TWO graphs, cut at the feature MLPs' outputs.  The gradients of the
        # first half (every relation pack, the read-out, the folded FC_output: ~3.9 MB) are all-reduced on a side stream
        # while the second half (the MLPs' backward: ~45 % of the backward's time, incl. the 5120-wide gene dW product)
        # runs; only the MLPs' own bucket (~2.7 MB) is reduced in the open.
        # single GPU, optimiser inside the graph: the partial-sum folds of the weight gradients, Adam and the statistics as one launch
        self.fused_adam = (ops._FUSED_ADAM and self.capture_optimizer and not self._multi and
                           sum(1 for p in self.model.parameters() if p.requires_grad) <= _lib.ADAM_FUSED_MAX)
        self.deferred_gradients = 0
        self.finish_fused = False          # (set below: multi-rank split backward)
        self._images, self._image_params, self._image_version, self._pack_probe = {}, [], {}, None
        self.split_backward = self._multi
        self.finish_fused = ops._FUSED_ADAM and self.split_backward
        self.duv_pieces = ops._DUV_PIECES and self.fused_adam           # (single GPU, whole backward in one autograd pass)
        # multi-rank, optional: the first gene Linear over the resident feature matrix (forward + weight gradient: 0.27 of the
        # step's 1.2 ms, identical work on every rank) split by gene rows over the ranks -- ops.GeneLayerShard in its STAGED form:
        # partial product (captured) | all-gather | the step's graphs | reduce-scatter of dz | partial weight gradient (captured).
        # Pays when (world - 1) / world of 0.27 ms (x 11 for the 57 742-wide features) exceeds two 10 MB collectives + three extra
        # graph boundaries -- under weak scaling as much as under strong: the layer's output does not depend on the batch (round 4:
        # on by default from 4 ranks at width 5 120, from 2 at 57 742).  KGW_SHARD_GENE_LAYER=1/0 overrides.
        env = os.environ.get('KGW_SHARD_GENE_LAYER')
        if shard_gene_layer is None:                                   # default: where it pays (ops.gene_layer_split_pays)
            shard_gene_layer = ops.gene_layer_split_pays(self.world, int(getattr(run.data, 'gene_init_dim_size', 0) or 0))
        want = (shard_gene_layer if env is None else env == '1') and self.split_backward
        # (whether the model takes the resident route for the gene features depends on the rank's own batches: probed by a
        #  warm-up step WITHOUT the shard and agreed over the ranks before any collective depends on it -- _capture)
        self.gene_shard = None
        self._want_gene_shard = bool(want)
        self._probe = None
        self._g_partial = self._g_dw = None
        self._comm = torch.cuda.Stream(device=dev) if self._multi else None
        self._ev_a, self._ev_ca = torch.cuda.Event(), torch.cuda.Event()
        self._flat_a = self._flat_b = None
        self._cut = [None, None]
        self.graphs_b = [None, None]
        from .optim import FusedAdam
        self.opt = FusedAdam(self.model.parameters(), lr=lr, weight_decay=weight_decay)   # one launch, capturable
        # device-side statistics accumulated inside the graph: [edges layer 1..L, sampled edges, error mask]
        self.stats = torch.zeros(L + 2, dtype=torch.int64, device=dev)
        self.loss = [None, None]
        self._unit = ops.unit_gradient(dev)
        self._flat = None                         # multi-rank: gradient bucket + {param: view}
        self._flat_grads = None
        self.graphs = [None, None]
        if overlap_sampling is None:
            overlap_sampling = '2'
        # '2' (default): the next batch is sampled by a graph of its own, replayed on a side stream while the step's
        # graph runs (the sampler is ~50 short dependent launches: they hide under the step's GEMMs; measured 2.21 vs
        # 2.31 ms/step); '0' / False: at the tail of the step's graph; '1' / True: on a forked branch of the same graph
        # (no gain: 2.31)
        self.overlap = overlap_sampling in (True, '1', 1)
        self.twin = overlap_sampling in ('2', 2)
        self.sample_graphs = [None, None]
        self._sampled = [torch.cuda.Event(), torch.cuda.Event()]
        self._side = side_stream(dev)
        self._have = [-1, -1]                      # batch index currently sampled into each buffer
        # (timing experiments only: train on stale batches to see what the concurrent sampler costs the step)
        self._skip_resample = False
        self._twin_pending = [False, False]        # the side stream holds an unfinished sample of this buffer
        # ``cache_batches`` (KGWAS.train with more than one epoch; off for a bench line -- the headline samples live): the batches of
        # the first pass over the loader are kept and put back in later passes instead of being sampled again (BatchCache)
        self.cache = None
        self._want_cache = bool(cache_batches) and os.environ.get('KGW_EPOCH_CACHE', '1') != '0'
        # (parameter-only kernels on a parallel branch of the captured step: measured again in round 5 -- 1.397 ms against 1.081, and
        #  the branch's queue displaces the side sampler's, overlap ratio -0.14 -- HIP-graph branches are not a tool here; removed)
        self._capture()

    # train on the batch held by bufs[cur]; concurrently sample ``self.seeds`` into bufs[1 - cur]
    def _step_body(self, cur: int):
        # (the shard is active for the forward passes THIS trainer issues and for nothing else in the process; so are the operand
        #  images its optimiser launch keeps current)
        with ops.gene_shard_scope(self.gene_shard, self._probe), ops.packed_scope(self._images, self._pack_probe):
            return self._step_body_inner(cur)

    # ---- operand images of packed weights (the first gene Linear on kgw_gemm3) ---------------------------------------------------
    # With the fused optimiser launch the image of the UPDATED weight is written by that launch, so the next step's forward does not
    # pack (one launch less).  The trainer owns the image: it packs it once itself, and again whenever the weight was changed by
    # anything but its own optimiser launch (load_state_dict, a copy_ between steps: the tensor's version counter moves -- the HIP
    # launches do not move it).
    def _adopt_images(self, seen):
        byptr = {p.data_ptr(): p for p in self.model.parameters()}
        for W in seen:
            p = byptr.get(W.data_ptr())
            if p is None or p.data_ptr() in self._images or not p.requires_grad or p.shape[1] % 32:
                continue
            img = torch.empty(int(_lib.lib().kgw_gemm3_packed_bytes(p.shape[1])), dtype=torch.uint8, device=p.device)
            self._images[p.data_ptr()] = img
            self.opt.packed_images[p] = img
            self._image_params.append(p)
        self._refresh_images(force=True)

    def _refresh_images(self, force=False):
        for p in self._image_params:
            if force or self._image_version.get(p) != p._version:
                ops.gemm3_pack(p.detach(), p.shape[1], False, out=self._images[p.data_ptr()])
                self._image_version[p] = p._version

    def _step_body_inner(self, cur: int):
        bs = self.batch_size
        main = torch.cuda.current_stream()
        if self.overlap:
            self._side.wait_stream(main)                               # fork
            with torch.cuda.stream(self._side):
                sample_into(self.dg, self.bufs[1 - cur], self.seeds, self.seed_type, record=False)
        buf = self.bufs[cur]
        batch = SampledBatch(self.dg, buf, self.meta, self.input_type, bs, static=True)
        self.opt.zero_grad(set_to_none=True)
        if self.split_backward:
            mlp_out = []
            with ops.readout_fold_deferred():
                loss, _ = self.model.forward_loss(batch.x_dict, batch.edge_index_dict, bs, batch.n_id(self.input_type),
                                                  self.dg.y[self.input_type], self.ld_w, mlp_out=mlp_out, unit_grad=True)
            hs = [h for h in mlp_out if h.requires_grad]
            late = self._late_params()
            early = [p for p in self.model.parameters() if p.requires_grad and id(p) not in late]
            g = torch.autograd.grad(loss, hs + early, grad_outputs=self._unit, allow_unused=True)
            live = [(p, gi) for p, gi in zip(early, g[len(hs):]) if gi is not None]
            if self._flat_a is None:
                self._flat_a = torch.empty(sum(p.numel() for p, _ in live), device=self.seeds.device)
                self._flat_grads, off = {}, 0
                for p, _ in live:
                    self._flat_grads[p] = self._flat_a[off:off + p.numel()].view_as(p)
                    off += p.numel()
            torch.cat([gi.reshape(-1) for _, gi in live], out=self._flat_a)
            self._cut[cur] = (hs, list(g[:len(hs)]))
        else:
            with ops.readout_fold_deferred():                          # (the backward pass below always runs)
                loss, _ = self.model.forward_loss(batch.x_dict, batch.edge_index_dict, bs, batch.n_id(self.input_type),
                                                  self.dg.y[self.input_type], self.ld_w, unit_grad=True)      # kgwas.py:137-145
            # fused optimiser launch: the weight-gradient products that feed only Adam stop after their first launch, their last
            # sums, the update, the step counter and the running totals are ONE launch (ops.GradSink, kgw_adam_fused)
            sink = ops.GradSink() if self.fused_adam else None
            pieces = {} if self.duv_pieces else None                   # (d u_r / d v_r stay eight pieces: their consumers add them)
            with ops.grad_sink_scope(sink), ops.duv_pieces_scope(pieces):
                loss.backward(gradient=self._unit)                     # (a resident 1.0: no ones_like fill per step)
            if pieces:
                raise ops.GradSinkMismatch(f'{len(pieces)} d u_r / d v_r tensor(s) left in pieces reached no consumer that adds them')
            if sink is not None:
                self.deferred_gradients = sum(1 for r in sink.records.values() if r[0] is not None) + 2 * len(sink.products)   # (gradients whose last sums the optimiser launch takes)
                self.opt.step_fused(sink, buf.meta.data_ptr(), self.run.gnn_num_layers, self.dg.n_hops, self.stats)
                # (launches of the step that carried its parameter-only forward work as rider blocks: the gene layer's kgw_gemm3)
                self.riders_taken = getattr(self.model, 'last_riders_taken', 0)
                self.reduces_ridden = sink.reduces_ridden    # (second launches of product groups that rode in a later launch)
                self.folds_ridden = sink.folds_ridden
                self.tail_taken = sink.tail_taken          # (kgw_param_tail: the backward's parameter-only end inside the deferred products' launch)
                if self.overlap:
                    main.wait_stream(self._side)                       # join
                elif not self.twin:
                    sample_into(self.dg, self.bufs[1 - cur], self.seeds, self.seed_type, record=False)
                return loss
        ticked = False
        if self.capture_optimizer:
            self.opt.step(tick=False)                                  # (the counter advances in the statistics launch below)
            ticked = True
        elif self.split_backward:
            pass                                                       # (second half: _step_body_b)
        elif self._multi:
            # gradients of all live tensors into ONE persistent bucket (the all-reduce and Adam run on it after the replay)
            live = [p for p in self.model.parameters() if p.grad is not None]
            if self._flat is None:
                self._flat = torch.empty(sum(p.numel() for p in live), device=self.seeds.device)
                self._flat_grads, off = {}, 0
                for p in live:
                    self._flat_grads[p] = self._flat[off:off + p.numel()].view_as(p)
                    off += p.numel()
            torch.cat([p.grad.reshape(-1) for p in live], out=self._flat)
        _lib.check(_lib.lib().kgw_accumulate_stats_tick(buf.meta.data_ptr(), self.run.gnn_num_layers, self.dg.n_hops,
                                                        self.stats.data_ptr(), self.opt.step_dev.data_ptr() if ticked else None,
                                                        _lib.stream_ptr()), 'kgw_accumulate_stats_tick')
        if self.overlap:
            main.wait_stream(self._side)                               # join
        elif not self.twin:
            sample_into(self.dg, self.bufs[1 - cur], self.seeds, self.seed_type, record=False)
        return loss

    def _late_params(self):
        """ids of the parameters whose gradients come out of the SECOND half of the split backward: the feature MLPs'
        Linears that sit before the cut (FC_output too unless it is folded into layer 1)."""
        m = self.model
        late = set()
        for mlp in (m.snp_feat_mlp, m.gene_feat_mlp, m.go_feat_mlp):
            mods = [mlp.FC_hidden, mlp.FC_hidden2] + ([] if getattr(m, 'fold_fc', False) else [mlp.FC_output])
            for mod in mods:
                late.update(id(p) for p in mod.parameters())
        return late

    def _step_body_b(self, cur: int):
        """Second half of the split backward: from the feature MLPs' outputs down to their weights."""
        hs, dhs = self._cut[cur]
        keep = [(h, d) for h, d in zip(hs, dhs) if d is not None]
        # (finish_fused: the MLPs' weight-gradient products stop after their first launch and ONE launch writes every finished
        #  gradient of this half into its slot of the bucket -- kgw_grad_finish: the fused optimiser launch's work units without the
        #  update, which has to wait for the all-reduce here)
        sink = ops.GradSink() if self.finish_fused else None
        with ops.grad_sink_scope(sink):
            torch.autograd.backward([h for h, _ in keep], grad_tensors=[d for _, d in keep])
        late = self._late_params()
        live = [p for p in self.model.parameters() if id(p) in late and p.grad is not None]
        gs = self.gene_shard
        staged = gs is not None and not gs.inline and gs.last is not None
        w1 = gs.last[1] if staged else None            # (its gradient is NOT produced by the backward: the partial product after the
        n_live = sum(p.numel() for p in live)          #  reduce-scatter writes it straight into the bucket, _stage_dw)
        if self._flat_b is None:
            self._flat_b = torch.empty(n_live + (w1.numel() if staged else 0), device=self.seeds.device)
            off = 0
            for p in live:
                self._flat_grads[p] = self._flat_b[off:off + p.numel()].view_as(p)
                off += p.numel()
            if staged:
                self._flat_grads[w1] = self._flat_b[off:off + w1.numel()].view_as(w1)
        if sink is not None:
            self.deferred_gradients = sum(1 for r in sink.records.values() if r[0] is not None) + 2 * len(sink.products)
            self.opt.finish_into(sink, [(p, self._flat_grads[p]) for p in live])
        else:
            torch.cat([p.grad.reshape(-1) for p in live], out=self._flat_b[:n_live])

    def _warm_step(self, k: int):
        """One eager step of the capture warm-up; a fused form that does not apply to this model (ops.GradSinkMismatch, raised before
        the launch that would have used it) is switched off and the step issued again."""
        try:
            self._step_body(k)
            if self.split_backward:
                self._step_body_b(k)
        except ops.GradSinkMismatch as e:
            print(f'kgwas_amd: fused gradient-finishing launch not used ({e})', file=sys.stderr)
            self.fused_adam = self.finish_fused = self.duv_pieces = False
            # (nobody keeps an operand image current without the fused optimiser launch: the forward packs again)
            self._images.clear(); self._image_params.clear(); self._image_version.clear(); self.opt.packed_images.clear()
            self.opt.zero_grad(set_to_none=True)
            self._step_body(k)
            if self.split_backward:
                self._step_body_b(k)

    def _sample_now(self, which: int, i: int):
        b = self.batch_size
        if self.cache is not None:
            torch.cuda.current_stream().wait_stream(self._side)    # (the cache's slot index is shared with the side stream's launches)
        if self.cache is not None and self.cache.filled[i]:
            self.cache.restore(which, i)
        else:
            self.seeds.copy_(self.ids[i * b:(i + 1) * b])
            sample_into(self.dg, self.bufs[which], self.seeds, self.seed_type, record=False)
            if self.cache is not None:
                self.cache.save(which, i)
        self._have[which] = i

    def _capture(self):
        # warm-up on a side stream (allocator pools, one-time kernel attributes, Adam state), then undo its effect
        params = [p for p in self.model.parameters()]
        snap = [p.detach().clone() for p in params]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._sample_now(0, 0)
            if self._want_gene_shard:
                # probe: one step with the layer computed locally (no collective) tells whether -- and on which (X, W, b) -- this
                # rank's model takes the resident route; the shard is switched on only if EVERY rank does (MIN over the ranks),
                # and then directly in its staged form: no collective is ever issued from inside an autograd node here
                import torch.distributed as dist
                self._probe = []
                self._warm_step(0)
                seen, self._probe = self._probe, None
                ok = torch.tensor([1 if len(seen) == 1 else 0], dtype=torch.int32, device=self.seeds.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                gs0 = ops.GeneLayerShard(dist.get_rank(), dist.get_world_size(), None, inline=False) if int(ok[0]) else None
                if gs0 is not None and not gs0.selftest(self.seeds.device):      # (collective: every rank, same verdict)
                    print('kgwas_amd: the gene-layer split\'s collectives failed their self-test on this backend: layer kept replicated',
                          file=sys.stderr)
                    gs0 = None
                if gs0 is not None:
                    gs0._setup(seen[0][0])
                    gs0.last = seen[0]
                    self.gene_shard = gs0
                    self._flat_b = None                        # (rebuilt with a slot for the first layer's weight)
            gs = self.gene_shard
            for k in range(4):
                if gs is not None:
                    gs.forward_partial(*gs.last)
                    gs.gather()
                if k == 0 and self.fused_adam and os.environ.get('KGW_ADAM_PACKS', '1') != '0':
                    self._pack_probe = []                      # which weights does the forward pack for kgw_gemm3?
                self._warm_step(k % 2)                         # (a fused form that does not apply is switched off here)
                if self._pack_probe is not None:
                    seen, self._pack_probe = self._pack_probe, None
                    if self.fused_adam:
                        self._adopt_images(seen)
                if gs is not None:
                    gs.scatter()
                    gs.weight_grad_partial(gs.last[0], out=self._flat_grads[gs.last[1]])
                if self.twin:
                    self._sample_now(1 - k % 2, 0)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for b in self.bufs:
            err = int(b.read_meta().error)
            if err:
                raise _lib.KgwasHipError(f'static layout does not fit the sampler buffers (error mask {err})')
        if self.world > 1:
            # the gradient buckets hold the parameters that received a gradient in the warm-up steps: the same set on every rank
            # unless a node type is missing from one rank's batches -- agree before the first collective is sized by it
            from . import dist as kdist
            for name, b in (('first', self._flat_a), ('second', self._flat_b), ('whole', self._flat)):
                kdist.check_same_on_all_ranks(0 if b is None else b.numel(), f'size of the {name} gradient bucket')
        with torch.no_grad():
            for p, q in zip(params, snap):
                p.copy_(q)
            for st in self.opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
            self.opt.step_dev.zero_()
        self._refresh_images(force=True)                       # (the warm-up's updates are undone: so are their images)
        self.stats.zero_()
        self.opt.zero_grad(set_to_none=True)
        for cur in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.loss[cur] = self._step_body(cur)
            self.graphs[cur] = g
            if self.split_backward:
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb, pool=g.pool()):
                    self._step_body_b(cur)
                self.graphs_b[cur] = gb
                gs = self.gene_shard
                if gs is not None and self._g_partial is None:
                    # the two products of the sharded first gene layer: batch independent, captured once
                    self._g_partial, self._g_dw = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._g_partial, pool=g.pool()):
                        gs.forward_partial(*gs.last)
                    with torch.cuda.graph(self._g_dw, pool=g.pool()):
                        gs.weight_grad_partial(gs.last[0], out=self._flat_grads[gs.last[1]])
            if self.twin:
                gs = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gs, stream=self._side):
                    sample_into(self.dg, self.bufs[cur], self.seeds, self.seed_type, record=False)
                self.sample_graphs[cur] = gs
        self._have = [-1, -1]
        if self._want_cache and self.twin:
            try:
                self.cache = BatchCache(self.dg, self.bufs, self.n_batches)
            except (MemoryError, ValueError) as e:
                print(f'kgwas_amd: sampled batches are not kept for later epochs ({e})', file=sys.stderr)

    def step(self, i: int):
        """Train on batch ``i`` of the loader's fixed order (and pre-sample batch i+1); returns the (device,
        float64) loss tensor."""
        cur = i % 2
        if self._image_params:
            self._refresh_images()               # (a weight changed by anything but this trainer's own optimiser launch: re-pack)
        if self._have[cur] != i:                 # first call / non-sequential access: sample it now
            self._sample_now(cur, i)
        nxt = (i + 1) % self.n_batches
        b = self.batch_size
        if self.gene_shard is not None:
            self._g_partial.replay()               # this rank's rows of the first gene layer, then everybody's
            self.gene_shard.gather()
        if self.twin:
            main = torch.cuda.current_stream()
            # (round 6, measured: the step's graph handed to its queue BEFORE the sampler's changes nothing -- 0.9825 / 0.9812 against
            #  0.9794 / 0.9806 ms)
            self._side.wait_stream(main)          # the previous step (reader of bufs[1 - cur], writer of nothing here) is done
            with torch.cuda.stream(self._side):
                cache = self.cache if not self._skip_resample else None
                if cache is not None and cache.filled[nxt]:
                    cache.restore(1 - cur, nxt)                     # (a later epoch: one copy launch instead of the sampler's ~25)
                else:
                    self.seeds.copy_(self.ids[nxt * b:(nxt + 1) * b])
                    if not self._skip_resample:
                        self.sample_graphs[1 - cur].replay()
                        if cache is not None:
                            cache.save(1 - cur, nxt)
                self._sampled[1 - cur].record(self._side)
            if self._twin_pending[cur]:
                main.wait_event(self._sampled[cur])
            self.graphs[cur].replay()
            self._twin_pending[1 - cur] = True
            self._twin_pending[cur] = False
        else:
            self.seeds.copy_(self.ids[nxt * b:(nxt + 1) * b])
            self.graphs[cur].replay()
        self._have[1 - cur] = nxt
        self._have[cur] = -1
        if not self.capture_optimizer:
            if self.split_backward:
                from . import dist as kdist
                main = torch.cuda.current_stream()
                self._ev_a.record(main)
                with torch.cuda.stream(self._comm):                   # first bucket: reduced while the MLPs' backward runs
                    self._comm.wait_event(self._ev_a)
                    kdist.allreduce_flat(self._flat_a, self.world, 'gradients of the relation packs + read-out, under the MLPs\' backward')
                    self._ev_ca.record(self._comm)
                self.graphs_b[cur].replay()
                if self.gene_shard is not None:
                    self.gene_shard.scatter()      # dz summed over the ranks' batches, each rank its own gene rows
                    self._g_dw.replay()            # partial weight gradient of the first gene layer -> its slot of the bucket
                kdist.allreduce_flat(self._flat_b, self.world, 'gradients of the feature MLPs')        # second bucket: the MLPs' own gradients
                main.wait_event(self._ev_ca)
                self.opt.step(self._flat_grads)
            elif self._multi:
                from . import dist as kdist
                kdist.allreduce_flat(self._flat, self.world)          # one RCCL collective over the bucket
                self.opt.step(self._flat_grads)
            else:
                self.opt.step()
        return self.loss[cur]

    def measure_overlap(self, n: int = 20) -> dict:
        """Did the next batch's sampler really run BESIDE the step?  HIP maps streams onto a few hardware queues and a replayed
        graph runs on the queue of the stream it was captured on: two streams that share a queue serialise whatever the program
        says.  Times ``n`` steps with the side sampler, ``n`` without it (stale batches) and ``n`` sampler replays alone; overlap =
        the share of the sampler's own time that did NOT show up in the step.  Every rank calls it (multi-rank steps contain
        collectives).  Leaves the buffers unsampled (the next step() samples its batch itself) and the model, the optimiser state
        and the running totals as it found them."""
        import time
        if not self.twin:
            return {'overlapped': False, 'note': 'the sampler is not run beside the step in this configuration'}
        # the measurement runs real training steps (half of them on stale batches): parameters, optimiser state, step counter and
        # the running totals are put back afterwards -- a measurement must not move the trajectory
        params = [p for p in self.model.parameters()]
        snap = [p.detach().clone() for p in params]
        ostate = {p: {k: v.clone() for k, v in st.items() if torch.is_tensor(v)} for p, st in self.opt.state.items()}
        step0, stats0 = self.opt.step_dev.clone(), self.stats.clone()

        def timed(fn):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(n):
                fn(i)
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n * 1e3

        def samp(i):
            with torch.cuda.stream(self._side):
                self.sample_graphs[i % 2].replay()
        k = [0]

        def step(_):
            self.step(k[0] % self.n_batches)                  # (consecutive batch indices: no step samples in the open)
            k[0] += 1
        keep = self._skip_resample
        for i in range(2):
            step(i)
        both = timed(step)
        self._skip_resample = True
        step(0)
        alone = timed(step)
        self._skip_resample = keep
        sampler = timed(samp)
        torch.cuda.current_stream().wait_stream(self._side)
        torch.cuda.synchronize()
        self._have, self._twin_pending = [-1, -1], [False, False]
        with torch.no_grad():
            for p, q in zip(params, snap):
                p.copy_(q)
            for p, st in self.opt.state.items():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        v.copy_(ostate[p][k]) if p in ostate and k in ostate[p] else v.zero_()
            self.opt.step_dev.copy_(step0)
            self.stats.copy_(stats0)
        self._refresh_images(force=True)                        # (the operand images follow the restored weights)
        ratio = (alone + sampler - both) / max(sampler, 1e-9)
        return {'step_with_side_sampler_ms': both, 'step_alone_ms': alone, 'sampler_alone_ms': sampler, 'overlap_ratio': ratio,
                'overlapped': bool(ratio > 0.5), 'steps': n}

    def describe(self) -> str:
        if self.split_backward:
            return ('HIP graphs: forward + first half of the backward | second half (feature MLPs' +
                    (', its gradients finished and written into their bucket by one launch' if self.finish_fused else '') +
                    '); the first half\'s gradient bucket '
                    'is all-reduced (RCCL) on a side stream under the second graph, the MLPs\' bucket after it, then one Adam launch' +
                    ('; next batch sampled by a third graph on a side stream' if self.twin else ''))
        adam = ' + Adam)'
        if self.fused_adam:
            adam = (f' + one optimiser launch that also finishes the partial sums of {self.deferred_gradients} weight gradients' +
                    (', writes the first gene Linear\'s bf16 operand image' if self._image_params else '') + ' and keeps the step counters)')
        return ('HIP graphs: step graph (fwd + bwd' + (adam if self.capture_optimizer else '), RCCL all-reduce + Adam eager') +
                (' with the next batch sampled by a second graph on a side stream' if self.twin else ', sampling inside it'))

    def grads_ready(self):
        return [p.grad for p in self.model.parameters() if p.grad is not None]

    def poll(self):
        """Non-blocking form of ``check`` for the middle of an epoch: looks at the sticky error bit as it was when the PREVIOUS
        poll asked for it (an asynchronous copy into pinned memory) and asks again -- no stream sync, so the queue of replayed
        steps never drains (KGWAS.train polled with a blocking check every 128 steps: ~3 % of an epoch).  An overflow is reported
        one poll late; ``check`` at the end of the epoch is exact."""
        pend = getattr(self, '_poll', None)
        if pend is not None and pend[1].query():
            err = int(pend[0][0])
            self._poll = pend = None
            if err:
                raise _lib.KgwasHipError(f'a batch exceeded the static capacities (error mask {err}); raise `margin`')
        if pend is None:
            host = getattr(self, '_poll_host', None)
            if host is None:
                host = self._poll_host = torch.zeros(1, dtype=torch.int64).pin_memory()
            host.copy_(self.stats[-1:], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._poll = (host, ev)

    def check(self):
        """Synchronise and verify that no batch overflowed the static capacities."""
        torch.cuda.synchronize()
        err = int(self.stats[-1])
        if err:
            raise _lib.KgwasHipError(f'a batch exceeded the static capacities (error mask {err}); raise `margin`')
        return [int(v) for v in self.stats[:-1]]


# seeds per captured evaluation batch.  A seed's prediction does not depend on its batch mates (tests/test_gpu_fullsize.py), so the
# evaluation loops need not keep the loaders' 512 (kgwas/kgwas.py:104-113): every 512-seed batch of the benchmark graph pulls in all
# 20 032 genes and recomputes the whole gene side of layer 1 (the 26 GFLOP first Linear, ~0.7 M gene-to-gene edges) -- 1 061 times
# over for the whole-genome inference pass.  ONE batch of all 542 758 labelled SNPs aggregates ~21 M edges instead of 1.1 G.
EVAL_BATCH = int(os.environ.get('KGW_EVAL_BATCH', str(1 << 20)))


class GraphEvalStep:
    """Forward-only twin of GraphTrainStep for the evaluation / inference loops (kgwas/utils.py:20-39 driven by the
    val / test / infer loaders of kgwas/kgwas.py:104-113): one captured forward graph per buffer parity, the next
    batch sampled by its own graph on a side stream.  The nodes are evaluated in batches of ``eval_batch`` seeds (default
    EVAL_BATCH, never below the loader's own batch size): up to that many nodes are ONE batch; beyond, the id list is padded to a
    whole number of batches with ids from its head (distinct from the tail's) and the padded outputs are dropped."""

    def __init__(self, model, graph, num_layers: int, input_nodes, batch_size: int, device, margin: float = 1.03,
                 eval_batch: int = None):
        self.model = model
        dev = torch.device(device)
        self.input_type, ids = input_nodes
        ids = np.asarray(ids.cpu() if torch.is_tensor(ids) else ids, dtype=np.int64).reshape(-1)
        self.n = len(ids)
        if self.n < 1:
            raise ValueError('no evaluation nodes')
        big = max(int(batch_size), int(EVAL_BATCH if eval_batch is None else eval_batch))
        self.batch_size = bs = self.n if self.n <= big else big
        self.n_batches = (self.n + bs - 1) // bs
        pad = self.n_batches * bs - self.n
        ids_p = np.concatenate([ids, ids[:pad]])
        probe = NeighborLoader(graph, [-1] * num_layers, (self.input_type, ids_p), batch_size=bs, drop_last=True,
                               device=dev, prefetch=False)
        self.caps = probe.measure_caps(margin)
        self.dg = probe.dg.with_static_caps(self.caps)
        self.seed_type = probe.seed_type
        self.ids = probe.ids
        # (a single batch has nothing to run beside: its sampler takes the whole GPU, and one buffer is enough)
        grid = SIDE_SAMPLER_GRID if self.n_batches > 1 else 0
        self.bufs = [BatchBuffers(self.dg, grid) for _ in range(2 if self.n_batches > 1 else 1)]
        self.meta = self.dg.static_meta()
        self.seeds = torch.zeros(bs, dtype=torch.int64, device=dev)
        self.out = torch.zeros(self.n_batches * bs, device=dev)
        self.pred = [None, None]
        self.graphs = [None, None]
        self.sample_graphs = [None, None]
        self._side = torch.cuda.Stream(device=dev)
        self._sampled = [torch.cuda.Event(), torch.cuda.Event()]
        self._capture()

    def _forward(self, cur: int):
        batch = SampledBatch(self.dg, self.bufs[cur], self.meta, self.input_type, self.batch_size, static=True)
        with torch.no_grad():
            return self.model(batch.x_dict, batch.edge_index_dict, self.batch_size).reshape(-1)

    def _sample_now(self, which: int, i: int):
        b = self.batch_size
        self.seeds.copy_(self.ids[i * b:(i + 1) * b])
        sample_into(self.dg, self.bufs[which], self.seeds, self.seed_type, record=False)

    def _capture(self):
        nb = len(self.bufs)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for k in range(3 if nb > 1 else 2):
                self._sample_now(k % nb, 0)
                self._forward(k % nb)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for b in self.bufs:
            err = int(b.read_meta().error)
            if err:
                raise _lib.KgwasHipError(f'static layout does not fit the sampler buffers (error mask {err})')
        for cur in range(nb):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.pred[cur] = self._forward(cur)
            self.graphs[cur] = g
            gs = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gs, stream=self._side):
                sample_into(self.dg, self.bufs[cur], self.seeds, self.seed_type, record=False)
            self.sample_graphs[cur] = gs

    def run(self) -> torch.Tensor:
        """Predictions of every input node, in input order (device tensor [n])."""
        b, main = self.batch_size, torch.cuda.current_stream()
        self._sample_now(0, 0)
        err = torch.zeros(1, dtype=torch.int32, device=self.out.device)
        meta_err = [bf.meta.view(torch.int32)[_lib.KgwBatchMeta.error.offset // 4:][:1] for bf in self.bufs]
        pending = [False, False]
        for i in range(self.n_batches):
            cur = i % len(self.bufs)
            if i + 1 < self.n_batches:
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    self.seeds.copy_(self.ids[(i + 1) * b:(i + 2) * b])
                    self.sample_graphs[1 - cur].replay()
                    self._sampled[1 - cur].record(self._side)
                pending[1 - cur] = True
            if pending[cur]:
                main.wait_event(self._sampled[cur])
                pending[cur] = False
            self.graphs[cur].replay()
            self.out[i * b:(i + 1) * b].copy_(self.pred[cur])
            err |= meta_err[cur]
        torch.cuda.synchronize()
        if int(err):
            raise _lib.KgwasHipError(f'a batch exceeded the static capacities (error mask {int(err)}); raise `margin`')
        return self.out[:self.n]
