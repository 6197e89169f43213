"""SNP-sharded multi-GPU mode (BASELINE.json north_star / configs[3]).

The reference trains on one device (kgwas/kgwas.py:38-39) and has nothing to mirror here; the contract is the
north_star's: *the SNP nodes shard across the GPUs of one node with Gene / GO replicated, boundary SNP->Gene messages and
gradients moved via RCCL*.  One process per GPU; rank p owns the SNPs of one contiguous id range (genome order =>
locality) -- their features, labels, LD weights, the CSR rows of every relation INTO them and the edges of every relation
OUT of them; genes, GO terms, their relations and all weights are replicated.

All ranks work on the SAME 512-seed batch of the reference's batch order:
  1. sampling: a rank expands the seeds it owns; after the first hop the ranks merge their gene frontiers (one all-reduce
     MIN over the ~44 k-entry replicated part of the global->local table, kgw_sample_batch_parts) so that every rank
     expands the same hop-1 genes in the same local order;
  2. layer 1 on the hop-1 genes: Gene<-Gene and Gene<-GO relations are computed by every rank (replicated); for a
     Gene<-SNP relation a rank aggregates only the edges whose SNP source it owns and leaves a PARTIAL online-softmax
     state (m, s, sum exp(e-m) h) per (gene, relation) (KgwLayerArgs.partial_rels).  The ranks all-gather those states
     (~3.7 MB per rank: 1.2 k genes x 6 relations x 132 floats) and each merges them in rank order
     (kgw_softmax_merge) -- the softmax of kgwas/conv.py:223 over ALL in-edges, bit-identical on every rank;
  3. everything downstream of the merged Z (transform, layer 2 on the rank's own seeds, read-out, loss over the rank's
     seeds / 512) is local;
  4. backward: gradients are linear in their upstream gradient, so replicated computations simply run on each rank's
     PARTIAL upstream gradient (their sum over ranks is the true gradient); the one place that needs the COMPLETE
     upstream gradient is the sharded aggregation itself -- dZ of the exchanged (gene, relation) segments is all-reduced
     (SUM, same 3.7 MB) before a rank differentiates its own Gene<-SNP edges against the merged softmax statistics;
  5. one flat all-reduce (SUM) of the parameter gradients, Adam on every rank (identical updates).
With P = 1 every collective is the identity and the step is the single-GPU step.

What is and is not saved: the 120 k-row SNP MLP, the SNP-side aggregation and the SNP feature / CSR memory divide by P;
the gene MLP, the Gene<-Gene / Gene<-GO aggregation (70 % of a batch's edges on the benchmark graph) and the transforms are
replicated work.  Seed-data-parallel training (kgwas_amd/dist.py) therefore scales better on graphs that fit one GPU;
this mode is for SNP sets whose features and edges do not (north_star: ~10 M SNPs), and it is the mode the contract names.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import KGW_C, PART_STRIDE

KGW_PENDING = -2            # include/kgwas_hip.h: a node flagged for expansion in the global->local table (unsampled = -1)
from .graph import HeteroGraph
from .sampler import BatchBuffers, DeviceGraph, SampledBatch, _ptr


def shard_range(n: int, rank: int, world: int):
    return n * rank // world, n * (rank + 1) // world


def shard_graph(data: HeteroGraph, rank: int, world: int, sharded_type: str = 'SNP', n_pad: int = 0):
    """Rank-local graph: nodes of ``sharded_type`` restricted to this rank's id range [lo, hi) and renumbered from 0,
    every other type whole; a relation keeps the edges whose ``sharded_type`` endpoint the rank owns.  ``n_pad`` extra nodes of
    the sharded type follow the owned ones (local ids hi - lo ...): zero features, no edges -- seeds of zero loss weight that pad
    a rank's share of a batch to a fixed count (the static layout of a captured step).  Returns (local HeteroGraph, lo, hi)."""
    n = int(data[sharded_type].num_nodes)
    lo, hi = shard_range(n, rank, world)
    g = HeteroGraph()
    for t in data.node_types:
        st = data[t]
        for k, v in st.items():
            if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == st.num_nodes:
                if t == sharded_type:
                    v = v[lo:hi]
                    if n_pad:
                        v = torch.cat([v, v.new_zeros((n_pad,) + tuple(v.shape[1:]))])
                g[t][k] = v
            elif k == 'num_nodes_':
                g[t][k] = (hi - lo + n_pad) if t == sharded_type else v
    for et in data.edge_types:
        s, _, d = et
        ei = data[et].edge_index
        ei = ei if torch.is_tensor(ei) else torch.as_tensor(ei)
        if s == sharded_type and d == sharded_type:
            if ei.shape[1]:
                raise NotImplementedError(f'relation {et}: both ends are the sharded type (needs a halo exchange)')
        elif s == sharded_type:
            keep = (ei[0] >= lo) & (ei[0] < hi)
            ei = ei[:, keep].clone()
            ei[0] -= lo
        elif d == sharded_type:
            keep = (ei[1] >= lo) & (ei[1] < hi)
            ei = ei[:, keep].clone()
            ei[1] -= lo
        g[et].edge_index = ei
    return g, lo, hi


def _streams_on_own_queues(device, group, multi: bool, n: int = 2, tries: int = 16):
    """``n`` streams that really run BESIDE each other AND beside the collective backend's stream.  HIP maps streams onto a few
    hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in creation order; two streams on one queue execute one after the other
    whatever the program says, and a REPLAYED GRAPH RUNS ON THE QUEUE OF THE STREAM IT WAS CAPTURED ON, not of the stream it is
    launched into.  Measured on the captured sharded step (one RCCL rank): 1.72 ms with the sampler's graphs on the step's queue,
    1.37 ms beside it; raising the queue count globally is no cure (16 queues: the seed-parallel step 1.25 -> 2.5 ms).  Which pool
    stream lands where depends on how many streams the process made before, so the choice is MEASURED: a busy stretch -- a spin
    kernel on an already chosen stream, a 64 MB all-reduce on the backend's stream -- and a tiny kernel on the candidate that
    must finish long before it.  Every rank runs the same sequence of collectives."""
    dev = torch.device(device)
    main = torch.cuda.current_stream(dev)
    x = torch.zeros(64, device=dev)
    big = torch.zeros(16 << 20, device=dev) if multi else None

    def beside(busy, cand):
        torch.cuda.synchronize(dev)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        if busy is None:                                   # the collective backend's stream
            e0.record(main)
            dist.all_reduce(big, group=group)              # (the main stream waits for it)
            e1.record(main)
        else:
            with torch.cuda.stream(busy):
                e0.record(busy)
                torch.cuda._sleep(200_000)
                e1.record(busy)
        with torch.cuda.stream(cand):
            cand.wait_event(e0)
            x.add_(1.0)
            e2.record(cand)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e2) < 0.5 * e0.elapsed_time(e1)

    chosen, spare = [], []
    for _ in range(tries):
        if len(chosen) == n:
            break
        cand = torch.cuda.Stream(device=dev)
        spare.append(cand)
        ok = beside(None, cand) if multi else True         # (every test on every candidate: the ranks stay in step)
        for c in chosen:
            ok = beside(c, cand) and ok
        if multi:
            # the verdict is a TIMING measurement, it can differ between ranks: a candidate is accepted only if every rank accepts
            # it (MIN), so that all ranks leave the loop after the same number of all-reduces -- a rank that went on probing alone
            # would pair its next probe with its peers' next real collective
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            ok = bool(int(t[0]))
        if ok:
            chosen.append(cand)
    while len(chosen) < n:                                 # (no free queue found: correctness does not depend on it)
        chosen.append(spare[len(chosen) % len(spare)])
    return chosen


class SegmentedCapture:
    """A step whose kernels are captured as HIP graphs with EAGER pieces (collectives) in between: ``cut(fn)`` ends the graph
    being captured, registers ``fn`` to run between the graphs at replay, and starts the next graph (one shared memory pool, so
    tensors allocated in one segment are the operands of the next and of the collectives).  While capturing, ``fn`` is NOT run
    -- every rank skips it alike."""

    def __init__(self, device, stream=None):
        self.items, self.pool, self.g = [], None, None
        self.stream = stream if stream is not None else torch.cuda.Stream(device=device)
        self.capturing = False

    def begin(self):
        self.g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        self.g.capture_begin(pool=self.pool)
        self.capturing = True

    def cut(self, fn):
        self.g.capture_end()
        self.items += [self.g, fn]
        self.begin()

    def end(self):
        self.g.capture_end()
        self.items.append(self.g)
        self.g, self.capturing = None, False

    def replay(self):
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()


class ShardExchange:
    """The collectives of the sharded mode for one rank-local DeviceGraph (see the module docstring)."""

    def __init__(self, dg: DeviceGraph, sharded_type: str = 'SNP', group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # (KGW_FORCE_MULTIRANK_PATH=1: a single rank still issues every collective -- a 1-GPU box then exercises them over RCCL)
        self.multi = self.world > 1 or (os.environ.get('KGW_FORCE_MULTIRANK_PATH') == '1' and dist.is_initialized())
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # the flat all-gather exists on RCCL ("nccl"); gloo (the CPU / one-GPU tests) takes the list form.  Decided ONCE, from
        # the backend: a collective that fails at run time must surface as an error on every rank, not send one rank down a
        # different collective sequence from its peers.
        backend = dist.get_backend(group) if self.multi else None
        # ("fake": bench.py --as-rank R/P -- one GPU doing rank R's work, collectives that move nothing.  The stand-ins below keep
        #  the data finite and the merged frontier the size it has in the real job: see emulate_peers)
        self.fake = backend == 'fake'
        self.flat_gather = self.multi and backend in ('nccl', 'fake')
        self.peer_frontier, self.emu_batch = None, 0
        self.collectives = {}                     # name -> [calls, bytes this rank handed to the collective]
        # STAGED form (a step captured in segments, ShardedTrainer(use_graph=True)): the backward's exchange is not issued from
        # inside the autograd node (the engine runs it on another thread: no place to end a capture) -- ops.gat_aggregate hands the
        # downstream graph a detached leaf of the merged Z and lists (layer, Z, leaf) in ``cuts``; the trainer differentiates down
        # to the leaves, exchanges their gradients itself (``backward``) and continues from Z.
        self.staged = False
        self.seg = None                           # SegmentedCapture while a step is being captured
        self.cuts = []
        sc = dg.schema
        self.sharded = sc.type_id[sharded_type]
        self.dev = dg.device
        # replicated node types as contiguous runs of the global->local table
        runs, t = [], 0
        while t < sc.NT:
            if t == self.sharded:
                t += 1
                continue
            u = t
            while u + 1 < sc.NT and u + 1 != self.sharded:
                u += 1
            runs.append((dg.node_base[t], dg.node_base[u + 1]))
            t = u + 1
        self.rep_runs = runs
        # per layer: relations whose source is sharded and whose destination is replicated ("exchange relations")
        self.mask, self.slots = {}, {}
        for l in range(1, dg.num_layers + 1):
            m, by_type = 0, {}
            for r in dg.live_rel[l]:
                s, d = int(sc.src_type[r]), int(sc.dst_type[r])
                if s == self.sharded and d == self.sharded:
                    raise NotImplementedError('a relation inside the sharded type needs a halo exchange')
                if s == self.sharded and d != self.sharded:
                    m |= 1 << r
                    by_type.setdefault(d, []).append(int(sc.slot_dst[r]))
            self.mask[l] = m
            self.slots[l] = {d: torch.tensor(sorted(v), dtype=torch.int32, device=self.dev) for d, v in by_type.items()}
        self.bytes_moved = 0

    def _collective(self, fn):
        """Run ``fn`` (a closure issuing collectives on persistent tensors) now -- or, while a step is being captured, end the
        current graph segment, register ``fn`` to run between the segments at replay, and open the next one."""
        if self.seg is not None and self.seg.capturing:
            self.seg.cut(fn)
        else:
            fn()

    def _count(self, name: str, nbytes: int):
        c = self.collectives.setdefault(name, [0, 0])
        c[0] += 1
        c[1] += int(nbytes)
        self.bytes_moved += int(nbytes)

    # -- sampling ---------------------------------------------------------------------------------------------------
    def merge_frontier(self, buf: BatchBuffers):
        """Union of the ranks' PENDING flags on the replicated node types (KGW_PENDING = -2 < -1 = unsampled)."""
        if not self.multi:
            return

        def fn():
            for lo, hi in self.rep_runs:
                dist.all_reduce(buf.g2l[lo:hi], op=dist.ReduceOp.MIN, group=self.group)
                self._count('all_reduce_min(frontier flags)', (hi - lo) * 4)
            if self.peer_frontier is not None:             # emulation: what the peers' flags would have added
                idx = self.peer_frontier[self.emu_batch % len(self.peer_frontier)]
                if idx.numel():
                    buf.g2l.index_fill_(0, idx, KGW_PENDING)
        self._collective(fn)

    def emulate_peers(self, dg: DeviceGraph, full: HeteroGraph, batches, lo: int, hi: int, sharded_type: str = 'SNP'):
        """bench.py --as-rank only (the "fake" backend): the frontier merge is a no-op there, so this rank would expand only the
        hop-1 neighbours of ITS seeds and the replicated work (70 % of a batch's edges) would come out too small.  Precompute, from
        the full host graph, the replicated nodes the OTHER ranks' seeds of every batch reach in one hop -- exactly the flags their
        all-reduce would deliver -- and set them in place of the collective.  ``batches``: global seed ids per batch."""
        assert self.fake, 'peer emulation is for the fake backend only'
        assert dg.n_hops == 2, 'one frontier merge (2 hops): later merges would meet nodes that already have local ids'
        dg_base = {name: int(dg.node_base[dg.schema.type_id[name]]) for name in full.node_types}
        adj = []
        n_sh = int(full[sharded_type].num_nodes)
        for et in full.edge_types:
            s_, _, d_ = et
            if d_ != sharded_type or s_ == sharded_type:
                continue
            ei = full[et].edge_index
            ei = (ei if torch.is_tensor(ei) else torch.as_tensor(ei)).cpu().numpy()
            order = np.argsort(ei[1], kind='stable')
            ptr = np.zeros(n_sh + 1, dtype=np.int64)
            np.add.at(ptr, ei[1] + 1, 1)
            adj.append((np.cumsum(ptr), ei[0][order], dg_base[s_]))
        out = []
        for b in batches:
            b = np.asarray(b, dtype=np.int64)
            other = b[(b < lo) | (b >= hi)]
            found = []
            for ptr, src, base in adj:
                a, e = ptr[other], ptr[other + 1]
                if len(other) and int((e - a).sum()):
                    found.append(np.concatenate([src[x:y] for x, y in zip(a, e)]) + base)
            idx = np.unique(np.concatenate(found)) if found else np.zeros(0, dtype=np.int64)
            out.append(torch.from_numpy(idx.astype(np.int64)).to(self.dev))
        self.peer_frontier = out

    # -- layer exchange ---------------------------------------------------------------------------------------------
    def seg_rows(self, batch: SampledBatch, layer: int) -> Optional[torch.Tensor]:
        """int32 Z rows of the exchanged (destination row, relation) segments of a layer, destination-type major; the
        destination rows of a replicated type are the same nodes in the same order on every rank."""
        key = ('xseg', layer)
        cache = batch.__dict__.setdefault('_xchg_cache', {})
        if key not in cache:
            m, sc = batch.meta, batch.dg.schema
            parts = []
            for d, slots in sorted(self.slots[layer].items()):
                rows = int(m.n_rows[layer - 1][d])
                if rows == 0:
                    continue
                R = int(sc.R_dst[d])
                base = torch.arange(rows, dtype=torch.int32, device=self.dev) * R + int(m.z_base[layer - 1][d])
                parts.append((base[:, None] + slots[None, :]).reshape(-1))
            cache[key] = torch.cat(parts) if parts else None
        return cache[key]

    def forward(self, batch: SampledBatch, layer: int, Z: torch.Tensor, stat: torch.Tensor):
        """Z / stat hold this rank's partial states on the exchanged segments: replace them, in place, by the merged
        softmax result over all ranks."""
        seg = self.seg_rows(batch, layer)
        if seg is None:
            return
        n = int(seg.numel())
        L = _lib.lib()
        mine = torch.empty(n * PART_STRIDE, device=self.dev)
        _lib.check(L.kgw_softmax_pack(_ptr(Z), _ptr(stat), _ptr(seg), n, _ptr(mine), _lib.stream_ptr()), 'kgw_softmax_pack')
        if self.multi:
            allp = torch.empty(self.world * n * PART_STRIDE, device=self.dev)

            def fn():
                if self.fake:                                # (stand-in: every peer's partial state = this rank's; finite)
                    allp.view(self.world, -1).copy_(mine.view(1, -1).expand(self.world, -1))
                elif self.flat_gather:
                    dist.all_gather_into_tensor(allp, mine, group=self.group)
                else:
                    dist.all_gather(list(allp.view(self.world, -1).unbind(0)), mine, group=self.group)
                self._count('all_gather(partial softmax states)', allp.numel() * 4)
            self._collective(fn)
        else:
            allp = mine
        _lib.check(L.kgw_softmax_merge(_ptr(allp), self.world, _ptr(seg), n, _ptr(Z), _ptr(stat), _lib.stream_ptr()),
                   'kgw_softmax_merge')

    def backward(self, batch: SampledBatch, layer: int, dZ: torch.Tensor) -> torch.Tensor:
        """dZ holds this rank's PARTIAL upstream gradient; the exchanged segments need the complete one: sum their rows
        over the ranks, in place."""
        seg = self.seg_rows(batch, layer)
        if seg is None or not self.multi:
            return dZ
        n = int(seg.numel())
        L = _lib.lib()
        rows = torch.empty(n, KGW_C, device=self.dev)
        _lib.check(L.kgw_gather_rows(_ptr(dZ), _ptr(seg), n, KGW_C, _ptr(rows), _lib.stream_ptr()), 'kgw_gather_rows')

        def fn():
            dist.all_reduce(rows, op=dist.ReduceOp.SUM, group=self.group)
            self._count('all_reduce_sum(dZ of exchanged segments)', rows.numel() * 4)
        self._collective(fn)
        _lib.check(L.kgw_scatter_rows(_ptr(rows), _ptr(seg), n, KGW_C, _ptr(dZ), _lib.stream_ptr()), 'kgw_scatter_rows')
        return dZ


def sample_sharded(dg: DeviceGraph, buf: BatchBuffers, seeds: torch.Tensor, seed_type: int, xchg: ShardExchange, record: bool = True):
    """kgw_sample_batch with the frontier merge between the two halves of every hop but the last."""
    st = torch.cuda.current_stream()
    L = _lib.lib()
    n = int(seeds.numel())
    last = 2 * dg.n_hops
    begin = 0
    for h in range(dg.n_hops - 1):
        _lib.check(L.kgw_sample_batch_parts(C.byref(dg.kg), C.byref(buf.c), _ptr(seeds), n, seed_type, 0, begin, 2 * h,
                                            C.c_void_p(st.cuda_stream)), 'kgw_sample_batch_parts')
        xchg.merge_frontier(buf)
        begin = 2 * h + 1
    _lib.check(L.kgw_sample_batch_parts(C.byref(dg.kg), C.byref(buf.c), _ptr(seeds), n, seed_type, 0, begin, last,
                                        C.c_void_p(st.cuda_stream)), 'kgw_sample_batch_parts')
    if record:
        buf.ready.record(st)


class ShardedTrainer:
    """Training steps of kgwas/kgwas.py:129-151 in the sharded mode.  ``input_nodes`` = (type, GLOBAL ids in training
    order): every rank is handed the same list and works on the seeds of each ``batch_size`` batch that it owns."""

    def __init__(self, run, input_nodes, batch_size: int, lr: float = 1e-4, weight_decay: float = 5e-4,
                 use_graph: bool = False, sharded_type: str = 'SNP', group=None):
        self.run, self.model = run, run.model
        self.batch_size = int(batch_size)
        self.input_type, ids = input_nodes
        if self.input_type != sharded_type:
            raise NotImplementedError('seeds must be of the sharded node type')
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        dev = torch.device(run.device)
        self.dev = dev
        full = run.data.data
        # ``use_graph``: the step runs from captured HIP graphs, its collectives between them (SegmentedCapture).  That needs a
        # STATIC layout: a rank's share of a batch is padded to a fixed seed count with zero-weight, edge-less pad nodes appended
        # to its SNP range (they contribute nothing to loss, gradients or the exchange)
        self.use_graph = bool(use_graph)
        self.n_pad = self.batch_size if self.use_graph else 0
        self.local, self.lo, self.hi = shard_graph(full, self.rank, self.world, sharded_type, n_pad=self.n_pad)
        L = run.gnn_num_layers
        self.dg = DeviceGraph(self.local, L, dev)
        self.xchg = ShardExchange(self.dg, sharded_type, group)
        # the first gene Linear over the resident (replicated) gene features is the same 2 x 26 GFLOP on every rank: split by gene
        # rows over the ranks, inline collectives inside the autograd node (ops.GeneLayerShard).  Default from 4 ranks on (two 10 MB
        # collectives against (world - 1) / world of 0.27 ms); KGW_SHARD_GENE_LAYER=1/0 overrides.
        from . import ops
        env = os.environ.get('KGW_SHARD_GENE_LAYER')
        on = self.xchg.multi and (ops.gene_layer_split_pays(self.world, int(getattr(run.data, 'gene_init_dim_size', 0) or 0))
                                  if env is None else env == '1')
        # (captured form: the STAGED variant -- partial product and all-gather ahead of the forward, reduce-scatter and the partial
        #  weight gradient after the backward, all at the trainer's level where a capture can be cut: _static_body)
        # (the replicated gene type expands the MERGED frontier: its row count -- hence the resident-route decision -- is the same
        #  on every rank by construction; the shard is active only inside this trainer's own forward passes, ops.gene_shard_scope)
        self.gene_shard = ops.GeneLayerShard(self.rank, self.world, group, inline=True) if on else None
        self.seed_type = self.dg.schema.type_id[self.input_type]
        self.buf = BatchBuffers(self.dg)
        ids = np.asarray(ids.cpu() if torch.is_tensor(ids) else ids, dtype=np.int64).reshape(-1)
        self.n_batches = len(ids) // self.batch_size
        if self.hi <= self.lo:
            raise ValueError(f'rank {self.rank} of {self.world} owns no {sharded_type} node ({int(full[sharded_type].num_nodes)} nodes)')
        # a rank that owns no seed of a batch still takes part in every collective of the step: it expands one node it owns
        # and its loss share is multiplied by zero (the extra hop-1 genes this adds to the merged frontier change no real
        # seed's prediction: a seed's output depends on ITS neighbourhood only)
        self.local_seeds, self.loss_scale = [], []
        for i in range(self.n_batches):
            b = ids[i * self.batch_size:(i + 1) * self.batch_size]
            mine = b[(b >= self.lo) & (b < self.hi)] - self.lo
            self.loss_scale.append(len(mine) / self.batch_size)
            if len(mine) == 0:
                mine = np.zeros(1, dtype=np.int64)
            self.local_seeds.append(torch.from_numpy(mine).to(dev))
        if self.xchg.fake:                 # bench.py --as-rank: stand-in for the peers' share of every frontier merge
            self.xchg.emulate_peers(self.dg, full, [ids[i * self.batch_size:(i + 1) * self.batch_size] for i in range(self.n_batches)],
                                    self.lo, self.hi, sharded_type)
        self.ld_w = run._ld_weight_vector()[self.lo:self.hi].contiguous()
        if self.n_pad:
            self.ld_w = torch.cat([self.ld_w, self.ld_w.new_zeros(self.n_pad)])       # pad seeds: loss weight 0
        self.y = self.dg.y[self.input_type]
        from .optim import FusedAdam
        self.opt = FusedAdam(self.model.parameters(), lr=lr, weight_decay=weight_decay)
        self._flat = None
        self._live = None
        self.last_loss = None
        self.seg = None
        if self.use_graph:
            self._setup_static()

    # ---- captured form ------------------------------------------------------------------------------------------------------
    def _setup_static(self):
        """Static layout + capture: (1) every batch's seed list padded to this rank's largest share; (2) a dry pass over all
        batches (with the frontier merges: every rank takes part) measures the capacities; (3) warm-up steps in the staged eager
        form (real collectives: allocator, Adam state, parameter liveness agreed), parameters restored; (4) ONE step captured
        as graph segments with the collectives between them."""
        from . import dist as kdist
        dev, sc, L = self.dev, self.dg.schema, self.dg.num_layers
        n_own = self.hi - self.lo
        self.s_cap = max(1, max(int(t.numel()) if sc_ > 0 else 0 for t, sc_ in zip(self.local_seeds, self.loss_scale)))
        seeds = []
        for t, sc_ in zip(self.local_seeds, self.loss_scale):
            real = t if sc_ > 0 else t[:0]
            pad = torch.arange(n_own, n_own + self.s_cap - int(real.numel()), dtype=torch.int64, device=dev)
            seeds.append(torch.cat([real, pad]))
        self.seed_table = torch.stack(seeds)                       # [n_batches, s_cap] local ids, real seeds first
        self.seeds_dev = torch.zeros(self.s_cap, dtype=torch.int64, device=dev)
        # (2) capacities
        node_off = np.zeros((sc.NT, L + 2), dtype=np.int64)
        edges, chunks = np.zeros(L, dtype=np.int64), np.zeros(L, dtype=np.int64)
        for i in range(self.n_batches):
            self.xchg.emu_batch = i
            sample_sharded(self.dg, self.buf, self.seed_table[i], self.seed_type, self.xchg)
            self.buf.ready.synchronize()
            m = self.buf.read_meta()
            if m.error:
                raise _lib.KgwasHipError(f'sampler capacity exceeded (error mask {m.error})')
            for t in range(sc.NT):
                for k in range(L + 2):
                    node_off[t, k] = max(node_off[t, k], int(m.node_off[t][min(k, self.dg.n_hops + 1)]))
            for l in range(L):
                edges[l] = max(edges[l], int(m.n_edges[l])); chunks[l] = max(chunks[l], int(m.n_chunks[l]))

        def up(v, hi=None):
            w = int(-(-int(v * 1.03 + 1) // 64) * 64)
            return min(w, hi) if hi is not None else w
        for t in range(sc.NT):
            for k in range(L + 2):
                if k == 0:
                    node_off[t, k] = 0
                elif t == self.seed_type and k == 1:
                    node_off[t, k] = self.s_cap
                else:
                    node_off[t, k] = up(node_off[t, k], hi=self.dg.n_nodes[t])
            node_off[t] = np.maximum.accumulate(node_off[t])
        from .sampler import BatchCaps
        caps = BatchCaps(node_off.tolist(), [up(e) for e in edges], [up(c) for c in chunks])
        # replicated types expand the same merged frontier on every rank: their capacities -- hence the exchanged row counts --
        # must agree (checked, not assumed: a mismatch would size the all-gather differently per rank)
        for t in range(sc.NT):
            if t != self.xchg.sharded:
                for k in range(L + 2):
                    kdist.check_same_on_all_ranks(int(node_off[t, k]), f'capacity of replicated node type {sc.node_types[t]}, hop {k}')
        self.dg_eager, self.buf_eager = self.dg, self.buf
        self.dg = self.dg_eager.with_static_caps(caps)
        # two batch buffers: while the step computes on one, the NEXT batch is sampled into the other on a side stream (its
        # frontier merge is a collective of that stream's segment list)
        self.overlap = True
        from .graph_step import SIDE_SAMPLER_GRID               # (a sampler beside a step keeps its launches small)
        grid = SIDE_SAMPLER_GRID if self.overlap else 0
        self.bufs = [BatchBuffers(self.dg, grid), BatchBuffers(self.dg, grid)]
        self.buf = self.bufs[0]
        self.seeds2 = [self.seeds_dev, torch.zeros_like(self.seeds_dev)]
        self.meta = self.dg.static_meta()
        self.xchg.staged = True
        self.loss_const = self.s_cap / self.batch_size              # mean over s_cap seeds (pads weigh 0) -> share of the batch mean
        self.loss_dev = [None, None]
        # (the stream the step's graphs are captured on, and the one the sampler's are captured on AND replayed on)
        self._cap_stream, self._side = _streams_on_own_queues(dev, self.xchg.group, self.xchg.multi)
        self._sampled = [torch.cuda.Event(), torch.cuda.Event()]
        self._pending = [False, False]
        self._have = [-1, -1]
        self.stats = torch.zeros(L + 2, dtype=torch.int64, device=dev)
        # (3) warm-up, eager + staged
        params = [p for p in self.model.parameters()]
        snap = [p.detach().clone() for p in params]
        self._skip_resample = False                                 # (measure_overlap: the step without a sampler beside it)
        for k in range(3):
            self.xchg.emu_batch = 0
            self.seeds2[k % 2].copy_(self.seed_table[0])
            self._sample_body(k % 2)
            self._compute_body(k % 2)
            gs = self.gene_shard
            if k == 0 and gs is not None:
                # the first warm-up step ran the gene-layer shard INLINE (fine outside a capture) -- which also tells whether the
                # model takes the resident route for the gene features at all; from here on the stages run at this level
                if gs.last is None:
                    self.gene_shard = None
                else:
                    gs.inline = False
        torch.cuda.synchronize()
        for b in self.bufs:
            if int(b.read_meta().error):
                raise _lib.KgwasHipError('static layout of the sharded step does not fit its buffers')
        with torch.no_grad():
            for p, q in zip(params, snap):
                p.copy_(q)
            for st in self.opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
            self.opt.step_dev.zero_()
        self.stats.zero_()
        # (4) capture
        self.samp_seg, self.comp_seg = [], []
        torch.cuda.synchronize()
        for b in (0, 1):
            for body, lst in ((self._sample_body, self.samp_seg), (self._compute_body, self.comp_seg)):
                # (a replayed graph runs on the hardware queue of the stream it was CAPTURED on: the sampler's segments are
                #  captured on the side stream they are replayed on, the step's on one stream of their own)
                seg = SegmentedCapture(dev, self._side if body == self._sample_body else self._cap_stream)
                self.xchg.seg = seg
                with torch.cuda.stream(seg.stream):
                    seg.begin()
                    body(b)
                    seg.end()
                torch.cuda.current_stream().wait_stream(seg.stream)
                lst.append(seg)
        self.seg = self.comp_seg[0]
        self.xchg.seg = None
        self.xchg.collectives, self.xchg.bytes_moved = {}, 0       # (count what the training steps move, not the set-up passes)

    def _sample_body(self, b: int):
        """Sampling of the seeds in ``seeds2[b]`` into ``bufs[b]`` (static layout; the frontier merge goes through
        ShardExchange._collective like every collective: run now, or a cut between two graph segments while capturing)."""
        sample_sharded(self.dg, self.bufs[b], self.seeds2[b], self.seed_type, self.xchg, record=False)

    def _compute_body(self, b: int):
        """One training step on the batch held by ``bufs[b]``: static layout, staged exchange."""
        xchg = self.xchg
        gs = self.gene_shard
        staged = gs is not None and not gs.inline
        if staged:                       # this rank's rows of the first gene Linear, then everybody's (batch independent)
            gs.forward_partial(*gs.last)
            xchg._collective(gs.gather)
        batch = SampledBatch(self.dg, self.bufs[b], self.meta, self.input_type, self.s_cap, static=True)
        batch.exchange = xchg
        xchg.cuts = []
        for p in self.model.parameters():
            p.grad = None
        from . import ops
        with ops.gene_shard_scope(gs):
            loss, _ = self.model.forward_loss(batch.x_dict, batch.edge_index_dict, self.s_cap, batch.n_id(self.input_type), self.y, self.ld_w)
        part = loss * self.loss_const
        # down to the leaves ops.gat_aggregate cut at the exchanged Z.  (retain_graph: parameter-side nodes -- the attention
        # vectors, the FC_output fold -- feed both sides of a cut and are differentiated once per side, each time with the part
        # of their output gradient that side produces: the parameter gradients accumulate to the same sums.)
        cuts = [c for c in reversed(xchg.cuts)]
        part.backward(retain_graph=bool(cuts))
        for k, (layer, Z, Zx) in enumerate(cuts):
            if Zx.grad is None:
                continue
            dZ = xchg.backward(batch, layer, Zx.grad)              # partial upstream gradient -> summed over the ranks
            Z.backward(dZ, retain_graph=k + 1 < len(cuts))         # ... and on through this rank's own edges
        xchg.cuts = []
        if staged:                       # dz summed over the ranks for the rows this rank owns -> its partial of the weight gradient
            xchg._collective(gs.scatter)
            gs.last[1].grad = gs.weight_grad_partial(gs.last[0])
        self.allreduce_grads()
        self.opt.step()
        _lib.check(_lib.lib().kgw_accumulate_stats(self.bufs[b].meta.data_ptr(), self.dg.num_layers, self.dg.n_hops,
                                                   self.stats.data_ptr(), _lib.stream_ptr()), 'kgw_accumulate_stats')
        self.loss_dev[b] = part.detach()

    def check(self):
        """Synchronise; raise if a batch overflowed the static capacities (captured form)."""
        torch.cuda.synchronize()
        if self.use_graph and int(self.stats[-1]):
            raise _lib.KgwasHipError(f'a batch exceeded the static capacities of the sharded step (error mask {int(self.stats[-1])})')

    def measure_overlap(self, n: int = 20) -> dict:
        """Did the next batch's sampler really run BESIDE the step?  (A replayed graph runs on the hardware queue of the stream
        it was captured on: two streams mapped to one queue serialise whatever the program says.)  Times ``n`` steps with the side
        sampler, ``n`` without it (stale batches) and ``n`` sampler replays alone; overlap = the share of the sampler's own time
        that did NOT show up in the step.  Every rank calls it (the steps contain collectives).  Leaves the buffers unsampled."""
        import time
        if not (self.use_graph and self.overlap):
            return {'overlapped': False, 'note': 'the sampler is not run beside the step in this configuration'}

        def timed(fn):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(n):
                fn(i)
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n * 1e3

        def samp(i):
            with torch.cuda.stream(self._side):
                self.xchg.emu_batch = i % self.n_batches
                self.samp_seg[i % 2].replay()
        k = [0]

        def step(_):
            self.step(k[0])                                    # (consecutive batch indices: no step samples in the open)
            k[0] += 1
        for i in range(2):
            step(i)
        both = timed(step)
        self._skip_resample = True
        step(0)
        alone = timed(step)
        self._skip_resample = False
        sampler = timed(samp)
        self._have, self._pending = [-1, -1], [False, False]
        ratio = (alone + sampler - both) / max(sampler, 1e-9)
        return {'step_with_side_sampler_ms': both, 'step_alone_ms': alone, 'sampler_alone_ms': sampler, 'overlap_ratio': ratio,
                'overlapped': bool(ratio > 0.5), 'steps': n}

    def describe(self) -> str:
        return (('HIP-graph segments with the collectives between them' if self.use_graph else 'eager launches') + '; per step: 1 frontier all-reduce(MIN), 1 all-gather of partial softmax states + 1 all-reduce of '
                'their dZ per exchanged layer, 1 flat gradient all-reduce (SUM)' +
                ('; first gene Linear split by gene rows over the ranks (all-gather of its output, reduce-scatter of its dz)'
                 if self.gene_shard is not None else ''))

    def collectives(self) -> dict:
        """{name: [calls, bytes handed to the collective by this rank]} of everything this trainer moved so far."""
        out = dict(self.xchg.collectives)
        if self.gene_shard is not None:
            out.update({k: v for k, v in self.gene_shard.bytes.items() if v[0]})
        return out

    def _eager(self):
        """(graph, buffers) of the exact-size eager passes: inference, edge counting, the uncaptured training step."""
        return (self.dg_eager, self.buf_eager) if self.use_graph else (self.dg, self.buf)

    def sample(self, i: int) -> SampledBatch:
        seeds = self.local_seeds[i % self.n_batches]
        dg, buf = self._eager()
        self.xchg.emu_batch = i % self.n_batches
        sample_sharded(dg, buf, seeds, self.seed_type, self.xchg)
        torch.cuda.current_stream().wait_event(buf.ready)
        buf.ready.synchronize()
        meta = buf.read_meta()
        if meta.error:
            raise _lib.KgwasHipError(f'sampler capacity exceeded (error mask {meta.error})')
        batch = SampledBatch(dg, buf, meta, self.input_type, int(seeds.numel()))
        batch.exchange = self.xchg
        return batch

    def count_edges(self, i: int):
        """(edges the step's kernels aggregate, edges the reference would touch) for batch i, every edge of the GLOBAL batch
        counted once across the ranks -- an exact-size sampling pass outside any timing (collective: every rank calls it)."""
        return self._count_edges(self.sample(i))

    def forward_backward(self, i: int):
        """Loss contribution of this rank's seeds (their weighted squared errors / batch_size) and its backward."""
        if self.use_graph:
            raise NotImplementedError('forward_backward is the uncaptured form: ShardedTrainer(use_graph=False)')
        batch = self.sample(i)
        n = batch.batch_size
        for p in self.model.parameters():
            p.grad = None
        from . import ops
        with ops.gene_shard_scope(self.gene_shard):
            loss, pred = self.model.forward_loss(batch.x_dict, batch.edge_index_dict, n, batch.n_id(self.input_type), self.y, self.ld_w)
        # mean over the rank's seeds -> its share of the batch mean (0 for a rank that owns none of the batch's seeds)
        part = loss * self.loss_scale[i % self.n_batches]
        part.backward()
        return batch, part.detach(), pred

    def allreduce_grads(self):
        """One flat SUM all-reduce over a FIXED parameter list -- every parameter that can receive a gradient (requires_grad:
        the structurally dead relation packs do not), zero-filled where this rank's backward produced none (a relation or
        node type that is empty on this shard), so the collective has the same size on every rank whatever its batch held.
        Parameters whose gradient is None on EVERY rank of every step (an MLP no live relation touches) cost a few zeros."""
        params = [p for p in self.model.parameters() if p.requires_grad]
        if self._flat is None:
            self._flat = torch.empty(sum(p.numel() for p in params), device=self.dev)
        flat = self._flat
        had = [p.grad is not None for p in params]
        # one launch for the whole bucket (zeros for the parameters without a gradient on this rank)
        torch.cat([p.grad.reshape(-1) if p.grad is not None else p.new_zeros(p.numel()) for p in params], out=flat)
        if self.xchg.multi:
            # which parameters are live is structural (the same relations / MLPs reach the read-out on every rank) except for
            # a node type or relation that happens to be empty on one shard: live = live on ANY rank, agreed once (first step)
            if self._live is None:
                live = torch.tensor(had, dtype=torch.int32, device=self.dev)
                dist.all_reduce(live, op=dist.ReduceOp.MAX, group=self.xchg.group)
                self._live = [bool(v) for v in live.cpu().tolist()]
            def fn():
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.xchg.group)
                self.xchg._count('all_reduce_sum(parameter gradients)', flat.numel() * 4)
            self.xchg._collective(fn)
            had = self._live
        off = 0
        for p, h in zip(params, had):
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p) if h else None      # (None: Adam skips it, like the reference's unused parameters)
            off += n

    def _count_edges(self, batch: SampledBatch):
        """(edges this step's kernels aggregated, edges the reference would touch) counting every edge of the GLOBAL batch
        once across the ranks: an edge with a sharded endpoint exists on exactly one rank, an edge between replicated
        types on all of them (counted by rank 0 only)."""
        m, dg = batch.meta, batch.dg
        sc, L = dg.schema, dg.num_layers
        sp = batch.buf.seg_ptr
        idx, wk, wr = [], [], []
        for h in range(dg.n_hops):
            for r in range(sc.NR):
                a, b = int(m.seg_off[h][r]), int(m.seg_off[h][r + 1])
                if b <= a:
                    continue
                own = 1 if (self.rank == 0 or self.xchg.sharded in (int(sc.src_type[r]), int(sc.dst_type[r]))) else 0
                layers = sum(1 for l in range(1, L + 1) if h <= L - l and r in dg.live_rel[l])
                idx += [a, b]; wk.append(own * layers); wr.append(own * L)
        if not idx:
            return 0, 0
        e = sp[torch.tensor(idx, device=self.dev)].view(-1, 2)
        cnt = (e[:, 1] - e[:, 0]).cpu().numpy().astype(np.int64)
        return int((cnt * np.asarray(wk)).sum()), int((cnt * np.asarray(wr)).sum())

    def step(self, i: int):
        if self.use_graph:
            cur, nb = i % 2, self.n_batches
            main = torch.cuda.current_stream()
            if self._have[cur] != i % nb:                          # first call / non-sequential access: sample it now
                self.seeds2[cur].copy_(self.seed_table[i % nb])
                self.xchg.emu_batch = i % nb
                self.samp_seg[cur].replay()
                self._have[cur], self._pending[cur] = i % nb, False
            if self._skip_resample:
                self._have[1 - cur] = (i + 1) % nb                 # (timing only: the next step trains on a stale batch)
            elif self.overlap:
                # the NEXT batch, on the side stream beside this step (the previous step, reader of that buffer, is enqueued
                # on the main stream already).  Its frontier all-reduce is issued before this step's collectives on every rank.
                nxt = (i + 1) % nb
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    self.seeds2[1 - cur].copy_(self.seed_table[nxt])
                    self.xchg.emu_batch = nxt
                    self.samp_seg[1 - cur].replay()
                    self._sampled[1 - cur].record(self._side)
                self._have[1 - cur], self._pending[1 - cur] = nxt, True
            if self._pending[cur]:
                main.wait_event(self._sampled[cur])
                self._pending[cur] = False
            self.comp_seg[cur].replay()
            self._have[cur] = -1
            self.last_loss = self.loss_dev[cur]
            return None
        batch, part, _ = self.forward_backward(i)
        self.allreduce_grads()
        self.opt.step()
        self.last_loss = part
        return self._count_edges(batch)

    @torch.no_grad()
    def predict(self, ids, model=None) -> torch.Tensor:
        """Predictions of GLOBAL SNP ids ``ids`` (any order, any ownership) in input order, on every rank: batches of
        ``batch_size``, each rank scores the seeds it owns (sharded forward), the results are summed across ranks.
        ``model``: score with this (replicated) model instead of the one being trained (the best-so-far copy)."""
        ids = np.asarray(ids.cpu() if torch.is_tensor(ids) else ids, dtype=np.int64).reshape(-1)
        out = torch.zeros(len(ids), device=self.dev)
        trained, self.model = self.model, (model if model is not None else self.model)
        try:
            return self._predict(ids, out)
        finally:
            self.model = trained

    def _predict(self, ids, out):
        was_training = self.model.training
        self.model.eval()
        bs = self.batch_size
        for a in range(0, len(ids), bs):
            b = ids[a:a + bs]
            sel = np.nonzero((b >= self.lo) & (b < self.hi))[0]
            # (a rank without a seed in the batch still takes part in the exchange: it samples an arbitrary owned node)
            mine = b[sel] - self.lo if len(sel) else np.zeros(1, dtype=np.int64)
            seeds = torch.from_numpy(np.ascontiguousarray(mine)).to(self.dev)
            dg, buf = self._eager()
            sample_sharded(dg, buf, seeds, self.seed_type, self.xchg)
            torch.cuda.current_stream().wait_event(buf.ready)
            buf.ready.synchronize()
            meta = buf.read_meta()
            if meta.error:
                raise _lib.KgwasHipError(f'sampler capacity exceeded (error mask {meta.error})')
            batch = SampledBatch(dg, buf, meta, self.input_type, int(seeds.numel()))
            batch.exchange = self.xchg
            p = self.model(batch.x_dict, batch.edge_index_dict, int(seeds.numel())).reshape(-1)
            if len(sel):
                out[torch.from_numpy(a + sel).to(self.dev)] = p
        if self.xchg.multi:
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.xchg.group)
        if was_training:
            self.model.train()
        return out
