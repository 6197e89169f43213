"""Device-resident graph + full-neighbourhood minibatch loader.

Mirrors the loader surface the reference uses (kgwas/kgwas.py:99-113,129-142; kgwas/utils.py:25-31):
``NeighborLoader(data, num_neighbors=[-1]*L, input_nodes=('SNP', ids), batch_size=..., drop_last=...)``
supports ``len()`` / iteration and yields batches with ``batch['SNP'].batch_size``, ``.x_dict``,
``.edge_index_dict``, ``batch['SNP'].y``, ``batch['SNP']['n_id']`` and ``.to(device)``.

What differs from PyG by design (MI355X-first):
  * the graph (CSR per relation, features, labels) is resident in HBM; a batch is sampled by
    ``kgw_sample_batch`` on a side HIP stream, one batch ahead of the consumer;
  * a batch carries the per-layer block structure the fused kernels consume (chunk lists,
    src-major transpose) and exposes COO ``edge_index_dict`` / gathered ``x_dict`` lazily, only for
    callers that ask for them;
  * no shuffle (the reference passes none, kgwas.py:93-94), ``num_workers`` is accepted and ignored.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import (KGW_CHUNK, KGW_MAX_LAYERS, KGW_MAX_RELS, KGW_MAX_TYPES, KGW_TILE, KgwBatchBuf, KgwBatchMeta,
                   KgwGraph)
from .graph import GraphSchema, HeteroGraph, build_csr

EdgeType = Tuple[str, str, str]


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class DeviceGraph:
    """The knowledge graph resident on one GPU: one (dst, src)-sorted CSR per relation
    (== the CSC PyG's NeighborLoader builds per edge type, kgwas/kgwas.py:99), node features,
    labels.  Built once; every loader / model call on the same device shares it."""

    def __init__(self, data: HeteroGraph, num_layers: int, device, out_type: str = 'SNP',
                 full_graph: bool = False):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.KgwasHipError('kgwas_amd needs a ROCm device (no CPU fallback); got ' + str(device))
        _lib.lib()
        self.data = data
        self.schema = GraphSchema(data.node_types, data.edge_types)
        sc = self.schema
        if sc.NT > KGW_MAX_TYPES or sc.NR > KGW_MAX_RELS or num_layers > KGW_MAX_LAYERS:
            raise ValueError('graph exceeds KGW_MAX_TYPES / KGW_MAX_RELS / KGW_MAX_LAYERS')
        if max(sc.R_src.max(), 0) > KGW_MAX_RELS // 2:
            raise ValueError('too many relations share one source type')
        self.num_layers = num_layers
        self.full_graph = full_graph
        self.n_hops = 1 if full_graph else num_layers
        self.n_nodes = [int(data[t].num_nodes) for t in sc.node_types]
        rowptrs, cols, rp_off, col_off = [], [], [], []
        seg_cap = edge_cap = chunk_cap = multi_cap = 0
        ro = co = 0
        out_edges = [0] * sc.NT                        # edges leaving the nodes of each type (all relations)
        self.multi_dst_types = set()                    # destination types that have a row of more than KGW_CHUNK in-edges
        for r, et in enumerate(sc.edge_types):
            cached = data._extra.get('csr', {}).get(tuple(et)) if hasattr(data, '_extra') else None
            if cached is not None:                    # kgwas_amd/ingest.py: CSR straight from the on-disk cache
                rp, col = np.asarray(cached[0]), np.asarray(cached[1])
            else:
                rp, col = build_csr(data[et].edge_index, self.n_nodes[sc.src_type[r]], self.n_nodes[sc.dst_type[r]])
            if rp[-1] >= 2 ** 31:
                raise ValueError('relation too large for int32 row pointers')
            deg = np.diff(rp)
            seg_cap += len(deg)
            edge_cap += int(rp[-1])
            out_edges[sc.src_type[r]] += int(rp[-1])
            nch = (deg + KGW_CHUNK - 1) // KGW_CHUNK
            chunk_cap += int(nch.sum())
            multi_cap += int((nch > 1).sum())
            if len(nch) and int(nch.max()) > 1:
                self.multi_dst_types.add(sc.dst_type[r])
            rowptrs.append(rp.astype(np.int32)); cols.append(col)
            rp_off.append(ro); col_off.append(co)
            ro += len(rp); co += len(col)
        self.seg_cap, self.edge_cap = int(seg_cap), int(edge_cap)
        # node types whose nodes average <= 4 out-edges: their source rows go to the backward's 8-rows-per-wavefront path
        # (KgwGraph.short_types: a scheduling hint, rows that do not qualify fall back)
        self.short_type_mask = sum(1 << t for t in range(sc.NT) if out_edges[t] <= 4 * max(self.n_nodes[t], 1))
        self.chunk_cap, self.multi_cap = int(chunk_cap) + 1, int(multi_cap) + 1
        self.trow_cap = int(sum(n * int(rs) for n, rs in zip(self.n_nodes, sc.R_src)))
        self.g_rowptr = torch.from_numpy(np.concatenate(rowptrs)).to(self.device)
        self.g_col = torch.from_numpy(np.concatenate(cols) if edge_cap else np.zeros(1, np.int32)).to(self.device)
        # KGW_TILE-aligned per-type regions of the concatenated node maps
        self.node_base = [0]
        for n in self.n_nodes:
            self.node_base.append(self.node_base[-1] + ((n + KGW_TILE - 1) // KGW_TILE) * KGW_TILE)
        self.node_slots = self.node_base[-1]
        self.live_rel, self.live_types = sc.live_relations(num_layers, out_type)
        self.rels_by_src_t = [torch.tensor(sc.rels_by_src[t], dtype=torch.long, device=self.device) for t in range(sc.NT)]
        self.rels_by_dst_t = [torch.tensor(sc.rels_by_dst[t], dtype=torch.long, device=self.device) for t in range(sc.NT)]

        g = KgwGraph()
        g.n_types, g.n_rels, g.n_layers, g.n_hops = sc.NT, sc.NR, num_layers, self.n_hops
        g.short_types = self.short_type_mask
        for t in range(sc.NT):
            g.n_nodes[t] = self.n_nodes[t]
            g.node_base[t] = self.node_base[t]
            g.R_dst[t] = int(sc.R_dst[t]); g.R_src[t] = int(sc.R_src[t])
        g.node_base[sc.NT] = self.node_base[sc.NT]
        for r in range(sc.NR):
            g.rel_src[r] = int(sc.src_type[r]); g.rel_dst[r] = int(sc.dst_type[r])
            g.rel_slot_dst[r] = int(sc.slot_dst[r]); g.rel_slot_src[r] = int(sc.slot_src[r])
            g.rowptr_off[r] = rp_off[r]; g.col_off[r] = col_off[r]
        for l in range(1, num_layers + 1):
            for r in self.live_rel[l]:
                g.rel_live[l - 1][r] = 1
        g.g_rowptr = self.g_rowptr.data_ptr()
        g.g_col = self.g_col.data_ptr()
        self.kg = g

        # resident features / labels
        self.x = {t: data[t].x.to(self.device, torch.float32).contiguous() for t in sc.node_types if 'x' in data[t]}
        self.y = {t: data[t].y.to(self.device) for t in sc.node_types if 'y' in data[t]}

    def with_static_caps(self, caps: "BatchCaps") -> "DeviceGraph":
        """Same resident graph, but batches are laid out with fixed row-block capacities (``caps``) so that
        every buffer address and launch geometry is batch independent -- the precondition for capturing
        the whole training step in a HIP graph."""
        import copy
        dg = copy.copy(self)                      # shares g_rowptr / g_col / features / labels
        g = KgwGraph()
        C.memmove(C.addressof(g), C.addressof(self.kg), C.sizeof(KgwGraph))
        g.static_layout = 1
        sc, L = self.schema, self.num_layers
        for l in range(1, L + 1):
            hd = min(L - l, self.n_hops - 1)
            for t in range(sc.NT):
                g.cap_rows[l - 1][t] = int(caps.node_off[t][hd + 1])
                g.cap_src[l - 1][t] = int(caps.node_off[t][hd + 2])
        dg.kg = g
        dg.caps = caps
        return dg

    def with_all_relations_live(self) -> "DeviceGraph":
        """Same resident graph with EVERY relation computed in EVERY layer (no structural pruning): what plain
        inference over all node types needs (the attention export, kgwas/utils.py:437-461)."""
        import copy
        dg = copy.copy(self)
        g = KgwGraph()
        C.memmove(C.addressof(g), C.addressof(self.kg), C.sizeof(KgwGraph))
        for l in range(self.num_layers):
            for r in range(self.schema.NR):
                g.rel_live[l][r] = 1
        dg.kg = g
        dg.live_rel = [None] + [list(range(self.schema.NR)) for _ in range(self.num_layers)]
        return dg

    def static_meta(self) -> KgwBatchMeta:
        """Host-side KgwBatchMeta holding the LAYOUT of a static-caps graph (same formulas as the device's
        k_layer_tables); the counts of a particular batch stay on the device."""
        m = KgwBatchMeta()
        sc, L, g = self.schema, self.num_layers, self.kg
        for l in range(1, L + 1):
            live = [r for r in range(sc.NR) if g.rel_live[l - 1][r]]
            zb = sb = tb = 0
            for t in range(sc.NT):
                dst_live = any(int(sc.dst_type[r]) == t for r in live)
                src_live = any(int(sc.src_type[r]) == t for r in live) or dst_live   # (k_layer_tables: same rule)
                lr = int(g.cap_rows[l - 1][t]) if dst_live else 0
                ls = int(g.cap_src[l - 1][t]) if src_live else 0
                m.lay_rows[l - 1][t], m.lay_src[l - 1][t] = lr, ls
                m.n_rows[l - 1][t], m.n_src[l - 1][t] = lr, ls
                m.z_base[l - 1][t] = zb; zb += lr * int(sc.R_dst[t])
                m.src_base[l - 1][t] = sb; sb += ls
                m.t_base[l - 1][t] = tb; tb += ls * int(sc.R_src[t])
            m.z_base[l - 1][sc.NT], m.src_base[l - 1][sc.NT], m.t_base[l - 1][sc.NT] = zb, sb, tb
            m.n_chunks[l - 1] = int(self.caps.chunks[l - 1])
            m.n_edges[l - 1] = int(self.caps.edges[l - 1])
        for t in range(sc.NT):
            for k in range(L + 2):
                m.node_off[t][k] = int(self.caps.node_off[t][k])
        return m

    @staticmethod
    def get(data: HeteroGraph, num_layers: int, device, full_graph: bool = False) -> "DeviceGraph":
        cache = data._extra.setdefault('_device_graphs', {})
        key = (str(torch.device(device)), num_layers, full_graph)
        if key not in cache:
            cache[key] = DeviceGraph(data, num_layers, device, full_graph=full_graph)
        return cache[key]


class BatchCaps:
    """Fixed capacities of a static batch layout: per node type the cumulative node counts per hop
    (``node_off[t][k]``), per layer the edge and chunk counts.  Built by ``NeighborLoader.measure_caps``."""

    def __init__(self, node_off, edges, chunks):
        self.node_off = node_off        # [NT][L+2] ints
        self.edges = edges              # [L]
        self.chunks = chunks            # [L]

    def __repr__(self):
        return f'BatchCaps(node_off={self.node_off}, edges={self.edges}, chunks={self.chunks})'


class BatchBuffers:
    """Worst-case sized device buffers of one in-flight batch (sized once; 288 GB of HBM makes
    re-allocation pointless)."""

    def __init__(self, dg: DeviceGraph, grid_blocks: int = 0):
        """``grid_blocks``: blocks of the sampler's launches into this buffer (0 = whole-GPU default); a sampler that runs
        BESIDE a training step (GraphTrainStep / GraphEvalStep) passes graph_step.SIDE_SAMPLER_GRID."""
        dev = dg.device
        L = dg.num_layers
        i32 = dict(dtype=torch.int32, device=dev)
        r4 = lambda n: (int(n) + 3) // 4 * 4        # (whole 16-byte units: graph_step.BatchCache copies the arrays in int4s)
        self.g2l = torch.empty(dg.node_slots, **i32)
        self.n_id = torch.zeros(dg.node_slots, **i32)
        self.seg_deg = torch.empty(dg.seg_cap + 1, **i32)
        self.seg_nch = torch.empty(dg.seg_cap + 1, **i32)
        self.seg_ptr = torch.empty(r4(dg.seg_cap + 2), **i32)
        self.seg_chptr = torch.empty(r4(dg.seg_cap + 2), **i32)
        self.col_local = torch.empty(r4(dg.edge_cap + 1), **i32)
        self.chunks = torch.empty((dg.chunk_cap + 1) * 8, **i32)
        self.multi = torch.empty(dg.n_hops * dg.multi_cap * 4, **i32)
        self.t_cnt = [torch.empty(r4(dg.trow_cap + 2), **i32) for _ in range(L)]
        self.t_ptr = [torch.empty(r4(dg.trow_cap + 2), **i32) for _ in range(L)]
        self.t_edge = [torch.empty(r4(dg.edge_cap + 1), **i32) for _ in range(L)]
        self.t_zrow = [torch.empty(r4(dg.edge_cap + 1), **i32) for _ in range(L)]
        self.t_rel = [torch.empty((dg.edge_cap + 16) // 16 * 16, dtype=torch.uint8, device=dev) for _ in range(L)]
        self.scan_cap = max(2 * (max(dg.seg_cap, dg.node_slots, dg.trow_cap) // KGW_TILE + 4),
                            int(_lib.lib().kgw_sampler_scan_ints(dg.seg_cap, dg.node_slots, dg.trow_cap)))
        self.scan_tmp = torch.empty(self.scan_cap, **i32)
        self.t_tmp = torch.empty(8 * (dg.edge_cap + 1), **i32)       # (the src-major sort's keys / chunk ids / sorted pairs of two layers)
        nbytes = C.sizeof(KgwBatchMeta)
        self.meta = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        self.meta_host = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
        b = KgwBatchBuf()
        b.g2l, b.n_id = self.g2l.data_ptr(), self.n_id.data_ptr()
        b.seg_deg, b.seg_nch = self.seg_deg.data_ptr(), self.seg_nch.data_ptr()
        b.seg_ptr, b.seg_chptr = self.seg_ptr.data_ptr(), self.seg_chptr.data_ptr()
        b.col_local, b.chunks, b.multi = self.col_local.data_ptr(), self.chunks.data_ptr(), self.multi.data_ptr()
        for l in range(L):
            b.t_cnt[l] = self.t_cnt[l].data_ptr(); b.t_ptr[l] = self.t_ptr[l].data_ptr()
            b.t_edge[l] = self.t_edge[l].data_ptr(); b.t_zrow[l] = self.t_zrow[l].data_ptr()
            b.t_rel[l] = self.t_rel[l].data_ptr()
        b.scan_tmp, b.meta, b.meta_host = self.scan_tmp.data_ptr(), self.meta.data_ptr(), self.meta_host.data_ptr()
        b.t_tmp = self.t_tmp.data_ptr()
        b.seg_cap, b.edge_cap, b.chunk_cap = dg.seg_cap, dg.edge_cap, dg.chunk_cap
        b.multi_cap, b.trow_cap, b.scan_cap = dg.multi_cap, dg.trow_cap, self.scan_cap
        b.grid_blocks = int(grid_blocks)
        self.c = b
        self.ready = torch.cuda.Event()      # sampling finished
        self.released: Optional[torch.cuda.Event] = None   # consumer finished with the previous contents

    def tensors(self):
        return [self.g2l, self.n_id, self.seg_deg, self.seg_nch, self.seg_ptr, self.seg_chptr, self.col_local,
                self.chunks, self.multi, self.scan_tmp, self.t_tmp, self.meta] + self.t_cnt + self.t_ptr + self.t_edge + self.t_zrow + self.t_rel

    def record_stream(self, stream):
        """The buffers were allocated on the consumer's stream but are written on the sampler's side stream:
        tell the caching allocator, so a buffer freed while sampling is still in flight is not handed out."""
        for t in self.tensors():
            t.record_stream(stream)

    def read_meta(self) -> KgwBatchMeta:
        m = KgwBatchMeta()
        C.memmove(C.addressof(m), self.meta_host.data_ptr(), C.sizeof(KgwBatchMeta))
        return m


class _NodeView(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class SampledBatch:
    """One minibatch: local node ids per type (seeds first), per-layer block structure for the fused
    kernels, and lazily materialised PyG-style views."""

    def __init__(self, dg: DeviceGraph, buf: BatchBuffers, meta: KgwBatchMeta, input_type: str, batch_size: int,
                 static: bool = False):
        """``meta``: host struct whose LAYOUT fields (z_base / src_base / t_base / lay_*) are valid; in a static
        layout its counts are capacities and the batch's own counts live only on the device (buf.meta)."""
        self.dg, self.buf, self.meta = dg, buf, meta
        self.input_type = input_type
        self.batch_size = batch_size
        self.static = static
        sc = dg.schema
        self.n_nodes = {t: int(meta.node_off[i][dg.n_hops + 1]) for i, t in enumerate(sc.node_types)}
        self._x = None
        self._ei = None
        self._views = {}
        self.exchange = None        # kgwas_amd.shard.ShardExchange in the SNP-sharded multi-GPU mode

    # --- PyG-batch duck type -------------------------------------------------------------------
    def to(self, device, *a, **k):
        return self

    def n_id(self, t: str) -> torch.Tensor:
        """int32 global ids of the sampled nodes of type t, local order (seeds first)."""
        i = self.dg.schema.type_id[t]
        b = self.dg.node_base[i]
        return self.buf.n_id[b:b + self.n_nodes[t]]

    def __getitem__(self, t):
        if isinstance(t, tuple):
            return _NodeView(edge_index=self.edge_index_dict[tuple(t)])
        if t not in self._views:
            v = _NodeView()
            v['n_id'] = self.n_id(t).long()
            if t == self.input_type:
                v['batch_size'] = self.batch_size
            if t in self.dg.y:
                v['y'] = self.dg.y[t][v['n_id']]
            self._views[t] = v
        v = self._views[t]
        if 'x' not in v and t in self.dg.x:
            v['x'] = self.x_dict[t]
        return v

    @property
    def x_dict(self) -> Dict[str, torch.Tensor]:
        """Raw input features of the sampled nodes (the loader's x[n_id] slicing), sliced per node type on
        first access: the fused model never asks for the 20 KB-per-row gene slice."""
        if self._x is None:
            self._x = LazyFeatureDict(self)
        return self._x

    @property
    def edge_index_dict(self):
        """Local COO (src_local, dst_local) int64 per relation, PyG layout [2, E_r]; materialised on
        first element access only (the fused path never needs it)."""
        if self._ei is None:
            self._ei = LazyEdgeIndexDict(self)
        return self._ei

    def _build_coo(self):
        dg, m = self.dg, self.meta
        sc = dg.schema
        out = OrderedDict()
        seg_ptr = self.buf.seg_ptr
        for r, et in enumerate(sc.edge_types):
            d = int(sc.dst_type[r])
            srcs, dsts = [], []
            for h in range(dg.n_hops):
                a, b = int(m.seg_off[h][r]), int(m.seg_off[h][r + 1])
                if b <= a:
                    continue
                sp = seg_ptr[a:b + 1].long()
                e0, e1 = int(sp[0]), int(sp[-1])
                deg = sp[1:] - sp[:-1]
                rows = torch.arange(b - a, device=dg.device) + int(m.node_off[d][h])
                dsts.append(torch.repeat_interleave(rows, deg))
                srcs.append(self.buf.col_local[e0:e1].long())
            if srcs:
                out[et] = torch.stack([torch.cat(srcs), torch.cat(dsts)])
            else:
                out[et] = torch.zeros(2, 0, dtype=torch.long, device=dg.device)
        return out

    # layout of layer l (1-based): row-block sizes and bases as python ints
    def lay_rows(self, l: int, t: int) -> int:
        return int(self.meta.lay_rows[l - 1][t])

    def lay_src(self, l: int, t: int) -> int:
        return int(self.meta.lay_src[l - 1][t])

    def rows_dev(self, t: str):
        """Static layout only: 1-element int32 device view of the number of rows of type t the batch really has in
        the first layer's input (KgwBatchMeta.n_src[0][t], written by the sampler); None for exact-size batches."""
        if not self.static:
            return None
        i = self.dg.schema.type_id[t]
        off = KgwBatchMeta.n_src.offset // 4 + i
        return self.buf.meta.view(torch.int32)[off:off + 1]

    @property
    def n_edges_per_layer(self):
        return [int(self.meta.n_edges[l]) for l in range(self.dg.num_layers)]

    @property
    def n_edges_sampled(self) -> int:
        return int(self.meta.edge_end[self.dg.n_hops - 1])


class BatchDict(dict):
    """dict that remembers the batch it came from, so HeteroGNN.forward can take the fused path when
    it is handed ``batch.x_dict`` / ``batch.edge_index_dict`` (kgwas/kgwas.py:138)."""

    def __init__(self, batch: SampledBatch, *a, **k):
        super().__init__(*a, **k)
        self.kgw_batch = batch


class LazyFeatureDict(BatchDict):
    def __init__(self, batch: SampledBatch):
        super().__init__(batch)
        self._types = [t for t in batch.dg.schema.node_types if t in batch.dg.x]

    def __missing__(self, t):
        if t not in self._types:
            raise KeyError(t)
        b = self.kgw_batch
        v = gather_rows(b.dg.x[t], b.n_id(t))
        dict.__setitem__(self, t, v)
        return v

    def get(self, t, default=None):
        return self[t] if t in self._types else default

    def __contains__(self, t):
        return t in self._types

    def __iter__(self):
        return iter(self._types)

    def __len__(self):
        return len(self._types)

    def keys(self):
        return list(self._types)

    def values(self):
        return [self[t] for t in self._types]

    def items(self):
        return [(t, self[t]) for t in self._types]


class LazyEdgeIndexDict(BatchDict):
    def __init__(self, batch: SampledBatch):
        super().__init__(batch)
        self._built = False

    def _build(self):
        if not self._built:
            self._built = True
            dict.update(self, self.kgw_batch._build_coo())

    def __getitem__(self, k):
        self._build(); return dict.__getitem__(self, k)

    def get(self, k, default=None):
        self._build(); return dict.get(self, k, default)

    def __iter__(self):
        self._build(); return dict.__iter__(self)

    def __len__(self):
        self._build(); return dict.__len__(self)

    def __contains__(self, k):
        self._build(); return dict.__contains__(self, k)

    def keys(self):
        self._build(); return dict.keys(self)

    def values(self):
        self._build(); return dict.values(self)

    def items(self):
        self._build(); return dict.items(self)


def gather_rows(src: torch.Tensor, ids: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """out[i] = src[ids[i]] via kgw_gather_rows (ids int32); ``out``: contiguous [n, width] destination."""
    n, w = int(ids.numel()), int(src.shape[1])
    if out is None:
        out = torch.empty(n, w, dtype=torch.float32, device=src.device)
    assert out.shape == (n, w) and out.is_contiguous()
    if n:
        _lib.check(_lib.lib().kgw_gather_rows(_ptr(src), _ptr(ids), n, w, _ptr(out), _lib.stream_ptr()),
                   'kgw_gather_rows')
    return out


def gather_rows_multi(jobs) -> None:
    """``jobs``: up to 8 (src [N_j, w], ids int32 [n_j], out [n_j, w] contiguous) triples of one row width w:
    out_j[i] = src_j[ids_j[i]] for all of them in ONE launch (kgw_gather_rows_multi)."""
    jobs = [(s, i, o) for s, i, o in jobs if int(i.numel())]
    if not jobs:
        return
    w = int(jobs[0][0].shape[1])
    n = len(jobs)
    S = (C.c_void_p * n)(); I = (C.c_void_p * n)(); O = (C.c_void_p * n)(); N = (C.c_int64 * n)()
    for k, (s, i, o) in enumerate(jobs):
        assert s.dtype == torch.float32 and s.is_contiguous() and i.dtype == torch.int32 and int(s.shape[1]) == w
        assert o.shape == (int(i.numel()), w) and o.is_contiguous()
        S[k], I[k], O[k], N[k] = _ptr(s).value, _ptr(i).value, _ptr(o).value, int(i.numel())
    _lib.check(_lib.lib().kgw_gather_rows_multi(n, S, I, N, w, O, _lib.stream_ptr()), 'kgw_gather_rows_multi')


def sample_into(dg: DeviceGraph, buf: BatchBuffers, seeds: Optional[torch.Tensor], seed_type: int, stream=None,
                record: bool = True):
    st = stream if stream is not None else torch.cuda.current_stream()
    n = 0 if seeds is None else int(seeds.numel())
    rc = _lib.lib().kgw_sample_batch(C.byref(dg.kg), C.byref(buf.c), _ptr(seeds), n, seed_type,
                                     1 if dg.full_graph else 0, C.c_void_p(st.cuda_stream))
    _lib.check(rc, 'kgw_sample_batch')
    if record:
        buf.ready.record(st)


def finish_sample(dg: DeviceGraph, buf: BatchBuffers, input_type: str, batch_size: int) -> SampledBatch:
    torch.cuda.current_stream().wait_event(buf.ready)
    buf.ready.synchronize()
    meta = buf.read_meta()
    if meta.error:
        raise _lib.KgwasHipError(f'sampler capacity exceeded (error mask {meta.error})')
    return SampledBatch(dg, buf, meta, input_type, batch_size)


def sample_full_graph(data: HeteroGraph, num_layers: int, device) -> SampledBatch:
    """Every node of every type is a seed: the block structure of the whole graph (used by
    HeteroGNN.forward on plain x_dict / edge_index_dict inputs and by the attention export)."""
    dg = DeviceGraph.get(data, num_layers, device, full_graph=True)
    buf = BatchBuffers(dg)
    sample_into(dg, buf, None, 0)
    return finish_sample(dg, buf, 'SNP', dg.n_nodes[dg.schema.type_id['SNP']] if 'SNP' in dg.schema.type_id else 0)


class NeighborLoader:
    def __init__(self, data: HeteroGraph, num_neighbors: Sequence[int], input_nodes, batch_size: int = 512,
                 drop_last: bool = False, num_workers: int = 0, shuffle: bool = False, sampler=None,
                 device=None, prefetch: bool = True, **kwargs):
        if any(int(k) != -1 for k in num_neighbors):
            raise NotImplementedError('only full-neighbourhood sampling ([-1]*L) is supported, as used at '
                                      'kgwas/kgwas.py:99-113')
        if shuffle:
            raise NotImplementedError('the reference never shuffles (kgwas/kgwas.py:93-94)')
        self.data = data
        self.num_layers = len(num_neighbors)
        self.input_type, ids = input_nodes
        ids = np.asarray(ids.cpu() if torch.is_tensor(ids) else ids, dtype=np.int64).reshape(-1)
        self.batch_size = int(batch_size)
        self.drop_last = drop_last
        self.device = torch.device(device if device is not None else 'cuda')
        self.dg = DeviceGraph.get(data, self.num_layers, self.device)
        self.seed_type = self.dg.schema.type_id[self.input_type]
        if ids.size and (ids.min() < 0 or ids.max() >= self.dg.n_nodes[self.seed_type]):
            raise ValueError('input_nodes out of range')
        self.ids = torch.from_numpy(ids).to(self.device)
        self.prefetch = prefetch
        self._bufs = None
        self._stream = None

    def __len__(self):
        n = self.ids.numel()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _ensure(self):
        if self._bufs is None:
            self._bufs = [BatchBuffers(self.dg) for _ in range(2 if self.prefetch else 1)]
            self._stream = torch.cuda.Stream(device=self.device) if self.prefetch else None
            if self._stream is not None:
                # freshly allocated blocks may still be in use by earlier work of the allocating stream
                self._stream.wait_stream(torch.cuda.current_stream())
                for b in self._bufs:
                    b.record_stream(self._stream)

    def _launch(self, i: int, buf: BatchBuffers):
        seeds = self.ids[i * self.batch_size:(i + 1) * self.batch_size]
        if self._stream is not None:
            if buf.released is not None:
                self._stream.wait_event(buf.released)
            sample_into(self.dg, buf, seeds, self.seed_type, self._stream)
        else:
            sample_into(self.dg, buf, seeds, self.seed_type)
        return int(seeds.numel())

    def measure_caps(self, margin: float = 1.03, round_to: int = 64) -> BatchCaps:
        """Dry pass over every batch of this loader (the batch order is fixed, kgwas.py:93-94): the largest
        node / edge / chunk counts, plus a safety margin, become the capacities of a static layout."""
        dg = self.dg
        sc, L = dg.schema, dg.num_layers
        node_off = np.zeros((sc.NT, L + 2), dtype=np.int64)
        edges = np.zeros(L, dtype=np.int64)
        chunks = np.zeros(L, dtype=np.int64)
        for b in self:
            m = b.meta
            for t in range(sc.NT):
                for k in range(L + 2):
                    node_off[t, k] = max(node_off[t, k], int(m.node_off[t][min(k, dg.n_hops + 1)]))
            for l in range(L):
                edges[l] = max(edges[l], int(m.n_edges[l]))
                chunks[l] = max(chunks[l], int(m.n_chunks[l]))

        def up(v, hi=None, exact=False):
            if exact:
                return int(v)
            w = int(-(-int(v * margin + 1) // round_to) * round_to)
            return min(w, hi) if hi is not None else w
        seed_t = self.seed_type
        for t in range(sc.NT):
            for k in range(L + 2):
                if k == 0:
                    node_off[t, k] = 0
                elif t == seed_t and k == 1:
                    node_off[t, k] = self.batch_size          # the seeds: exact
                else:
                    node_off[t, k] = up(node_off[t, k], hi=dg.n_nodes[t])
            node_off[t] = np.maximum.accumulate(node_off[t])
        return BatchCaps(node_off.tolist(), [up(e) for e in edges], [up(c) for c in chunks])

    def __iter__(self):
        self._ensure()
        nb = len(self)
        if nb == 0:
            return
        nbuf = len(self._bufs)
        pending = {}
        if self.prefetch:
            pending[0] = self._launch(0, self._bufs[0])
        for i in range(nb):
            buf = self._bufs[i % nbuf]
            if not self.prefetch:
                pending[i] = self._launch(i, buf)
            elif i + 1 < nb:
                nxt = self._bufs[(i + 1) % nbuf]
                # the consumer finished enqueuing work on batch i-1 (it asked for batch i): its buffer may
                # be overwritten once that work has executed
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                nxt.released = ev
                pending[i + 1] = self._launch(i + 1, nxt)
            yield finish_sample(self.dg, buf, self.input_type, pending.pop(i))
