"""autograd bridge to the fused attention-aggregate kernels (include/kgwas_hip.h).

``gat_aggregate(batch, layer, H, U, V)`` is, for every live relation r=(s,rel,d) of the layer at once,
    e_ij  = leaky_relu(<H_s[j], U[r]> + <H_d[i], V[r]>)       kgwas/conv.py:150-152,205,217
    alpha = softmax over the in-edges of destination i        kgwas/conv.py:223
    Z[i, r, :] = sum_j alpha_ij H_s[j, :]                     kgwas/conv.py:227-228,182
with hand-written backward (dst-major + src-major HIP passes, no atomics)."""
from __future__ import annotations

import ctypes as C
import os
import weakref
import sys

import torch

from . import _lib
from ._lib import KGW_C, PART_STRIDE, KgwLayerArgs


def _p(t):
    return t.data_ptr() if t is not None else 0


class _TunedLibraryGemm:
    """Scope for the few LIBRARY GEMMs whose shape is the same every step (the 5120-wide first gene layer on the
    resident feature matrix and its weight gradient): PyTorch's TunableOp picks the fastest hipBLASLt / rocBLAS
    solution for that shape the first time it is seen (a few seconds, during warm-up) instead of the default
    heuristic's -- 1.2-1.4x on these two products.  Off outside the scope, so batch-dependent shapes never tune."""

    def __init__(self):
        self.on = os.environ.get('KGW_TUNABLE_GEMM', '1') == '1'
        self._ready = False

    def __enter__(self):
        if not self.on:
            return self
        try:
            import torch.cuda.tunable as tn
            if not self._ready:
                # results are kept per device under the temp dir: the next process starts from the same choice
                tn.set_filename(os.path.join(os.environ.get('KGW_CACHE_DIR', '/tmp'), 'kgwas_amd_tunableop.csv'), True)
                tn.set_max_tuning_duration(200)          # ms per candidate solution
                tn.set_max_tuning_iterations(20)
                self._ready = True
            tn.enable(True)
            tn.tuning_enable(True)
        except Exception as e:                           # tuning is an optimisation: never let it take the step down
            print(f'kgwas_amd: TunableOp unavailable ({e}); using the default library GEMM', file=sys.stderr)
            self.on = False
        return self

    def __exit__(self, *exc):
        if self.on:
            try:
                import torch.cuda.tunable as tn
                tn.enable(False)
            except Exception:
                self.on = False
        return False


_TUNED = _TunedLibraryGemm()


class LibraryGemmLog:
    """Every product this package hands to the framework's GEMM library (hipBLASLt / rocBLAS) instead of its own HIP kernels is
    recorded here -- ``calls`` (total) and ``by_site`` ({(site, shape): n}).  The default GAT / relation-sum training step of a
    KG-sized graph must not have any (bench.py reports ``config.library_gemm_calls``; the full-size tests assert 0);
    ``KGW_STRICT=1`` (or ``strict = True``) turns a library route into an error instead of a silent fallback.  Calls made while a
    HIP graph is being captured count once (the capture), like every other launch of a captured step."""

    def __init__(self):
        self.strict = os.environ.get('KGW_STRICT', '0') == '1'
        # Routing (round 4): this package's own kernel whenever one CAN take the shape -- whatever the row count.  The library
        # is left with the products no kernel here takes (K > 2 304 on few rows outside the resident first layer) and is an
        # OPT-IN beyond that: KGW_ALLOW_LIBRARY=1 restores the row thresholds of rounds 1-3, which hand problems of a few
        # hundred rows to hipBLASLt because one library launch beats a 128-row-tile kernel on a handful of workgroups (a
        # performance choice for toy graphs, never taken at the sizes of BASELINE.json's configurations).
        self.allow_library = os.environ.get('KGW_ALLOW_LIBRARY', '0') == '1'
        self.calls = 0
        self.by_site = {}

    @property
    def own_first(self) -> bool:
        return self.strict or not self.allow_library

    def note(self, site: str, *shape):
        key = (site, tuple(int(v) for v in shape))
        if self.strict:
            raise RuntimeError(f'KGW_STRICT: {site} {key[1]} would run on the library GEMM (no HIP kernel of this package takes '
                               f'that shape); unset KGW_STRICT to allow the fallback')
        self.calls += 1
        self.by_site[key] = self.by_site.get(key, 0) + 1

    def reset(self):
        self.calls = 0
        self.by_site = {}


LIBRARY_GEMM = LibraryGemmLog()
ROUTES = {}          # launches of selected own kernels by C-ABI name (tests assert that a product really took the route they check)


def _route(name: str):
    ROUTES[name] = ROUTES.get(name, 0) + 1


class KernelTimer:
    """Optional HIP-event timing of the main aggregate kernels (bench.py's live roofline measurement): raw
    hipEvent_t pairs handed to the C ABI (KgwLayerArgs.ev_before / ev_after), which records them on the launch
    stream immediately around k_agg_fwd / k_agg_bwd_dst / k_agg_bwd_src.  Disabled by default: no events."""

    def __init__(self):
        self.enabled = False
        self.records = []      # (tag, layer, start_event, end_event, n_edges, z_rows, n_src)
        self._hip = None
        self._pool = []

    def _event(self):
        if self._hip is None:
            self._hip = C.CDLL('libamdhip64.so')
            self._hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
            self._hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        ev = C.c_void_p()
        rc = self._hip.hipEventCreate(C.byref(ev))
        if rc:
            raise _lib.KgwasHipError(f'hipEventCreate failed ({rc})')
        return ev

    def attach(self, a: KgwLayerArgs, tag, layer, n_edges, z_rows, n_src):
        if not self.enabled:
            return
        e0, e1 = self._event(), self._event()
        a.ev_before, a.ev_after = e0, e1
        self.records.append((tag, layer, e0, e1, n_edges, z_rows, n_src))

    def summary(self):
        """{(tag, layer): dict(n, ms_total, edges, z_rows, n_src)} -- call after a device sync."""
        out = {}
        for tag, layer, e0, e1, ne, zr, ns in self.records:
            ms = C.c_float()
            rc = self._hip.hipEventElapsedTime(C.byref(ms), e0, e1)
            if rc:
                raise _lib.KgwasHipError(f'hipEventElapsedTime failed ({rc})')
            d = out.setdefault((tag, layer), dict(n=0, ms=0.0, edges=0, z_rows=0, n_src=0))
            d['n'] += 1; d['ms'] += ms.value; d['edges'] += ne; d['z_rows'] += zr; d['n_src'] += ns
        return out


TIMER = KernelTimer()


# d u_r / d v_r left as their eight level-1 pieces by the aggregate's backward (KGW_F_DUV_PIECES: no k_duv_fold launch): inside a
# ``duv_pieces_scope`` (a captured training step) _GatAggregate.backward hands autograd UNWRITTEN dU / dV tensors and notes here, by
# their addresses, where the pieces are; the two consumers that can add them on the fly (_FoldFC.backward -> k_fold_bwd,
# _RelVectorsMulti.backward -> k_relvec_bwd) look their incoming gradients up and take the entry.  An entry still here when the
# backward pass is over means a gradient went somewhere else: the trainer then switches the scheme off (GradSinkMismatch).
DUV_PIECES = None
_DUV_PIECES = os.environ.get('KGW_DUV_PIECES', '1') != '0'


class duv_pieces_scope:
    def __init__(self, table):
        self.table = table

    def __enter__(self):
        global DUV_PIECES
        self.prev, DUV_PIECES = DUV_PIECES, self.table
        return self.table

    def __exit__(self, *exc):
        global DUV_PIECES
        DUV_PIECES = self.prev
        return False


def _take_pieces(t):
    """(pointer to the [rows][8][128] pieces, keep-alive) if ``t`` is an unwritten d u_r / d v_r tensor of the active scope."""
    if DUV_PIECES is None or t is None:
        return None
    return DUV_PIECES.pop(t.data_ptr(), None)


class GradSinkMismatch(RuntimeError):
    """A gradient whose last reduction was left to the optimiser's launch did not arrive at a parameter as the tensor its producer
    wrote the record for (autograd copied or accumulated it): the fused optimiser launch cannot be used for this model."""


class GradSink:
    """While one is active (``grad_sink_scope``: the backward pass of a captured single-GPU training step), the weight-gradient
    products that feed NOTHING but the optimiser stop after their first launch and leave a KgwGradSrc record here, keyed by the
    address of the gradient tensor they return to autograd; ``FusedAdam.step_fused`` looks every parameter's ``.grad`` up, finishes
    the sums inside the optimiser's own launch (kgw_adam_fused: same order, bit-identical values, the gradient tensors filled in
    as a by-product) and fails loudly if a record is left over.  The records keep the partial-sum workspaces alive -- inside a
    graph capture a freed block would be handed to the next allocation of the same capture."""

    def __init__(self):
        self.records = {}           # data_ptr of the gradient tensor -> (KgwGradSrc, numel, workspace)
        self.products = []          # weight-gradient products not launched yet, see defer_product
        # the parameter-only END of the backward pass -- kgw_fold_bwd and kgw_relvec_bwd_multi: they feed nothing but the optimiser --
        # not launched where autograd reaches them but by ``flush``, as blocks of the deferred products' launch (kgw_param_tail)
        self.fold_bwd = None        # (KgwFoldArgs, keep-alive, [(address, numel, storage) of the leaf gradients it writes])
        self.relvec_bwd = None      # (n jobs, jobs, keep-alive, [(address, numel, storage)])
        self.tail_taken = 0         # kgw_param_tail launches issued
        self.pending_reduce = None  # (KgwTnReducePlan, keep-alive) of a product group whose second launch is waiting for a carrier
        self.reduces_ridden = 0     # plans that rode in a later launch
        self.keep = []              # operands of launches issued on behalf of others: alive until the optimiser's launch is enqueued
        self.pending_fold = None    # (KgwReadoutFold, keep-alive): the read-out node's second launch, waiting for a carrier
        self.folds_ridden = 0

    def defer_product(self, dY: torch.Tensor, X: torch.Tensor, rows_dev=None):
        """(dW [out, in], db [out]) = (dY^T X, column sums of dY) of a Linear whose gradients feed only the optimiser -- not launched
        now: the MLPs' backward passes end in one such product each (gene 20 k rows, SNP 122 k, GO 2 x 7 k), 15 - 40 us launches that
        depend on nothing but their own operands; ``flush`` (called by FusedAdam.step_fused) issues them as ONE kgw_tn_gemm_multi
        launch ahead of the optimiser's, where the short ones fill the slots the tall one leaves.  None: the shape is not the grouped
        kernel's (the caller launches it itself)."""
        rows, M = dY.shape
        N = X.shape[1]
        # (a tall product -- the SNP MLP's 122 k rows: 512 blocks that fill the chip for 43 us -- is launched where it is: grouped with
        #  it the short ones queue behind its blocks, measured in round 4)
        if not (_DEFER_PRODUCTS and 0 < rows < 32768 and X.shape[0] == rows and dY.dtype == torch.float32 and X.dtype == torch.float32 and
                dY.stride(1) == 1 and X.stride(1) == 1 and M % 2 == 0 and N % 2 == 0 and M >= 64 and N >= 64 and dY.stride(0) % 2 == 0 and
                X.stride(0) % 2 == 0 and dY.data_ptr() % 8 == 0 and X.data_ptr() % 8 == 0):
            return None
        dW = torch.empty(M, N, device=dY.device)
        db = torch.empty(M, device=dY.device)
        # (addresses + STORAGES, not the tensors: a second reference to the tensor would make autograd copy the gradient instead of
        #  adopting it -- the storage reference only keeps the memory from being handed out again before the product has run)
        self.products.append((dY, X, dW.data_ptr(), db.data_ptr(), dW.untyped_storage(), db.untyped_storage(), rows_dev))
        return dW, db

    def _tn_jobs(self, chunk):
        L = _lib.lib()
        jobs = (_lib.KgwTnJob * max(len(chunk), 1))()
        src = (_lib.KgwGradSrc * max(2 * len(chunk), 1))()
        keep = []
        for j, (dY, X, dW_ptr, db_ptr, _sw, _sb, rows_dev) in zip(jobs, chunk):
            rows, M = dY.shape
            N = X.shape[1]
            nws = int(L.kgw_tn_gemm_workspace_floats(rows, M, N))
            ws = torch.empty(nws, device=dY.device)
            keep.append(ws)
            j.A, j.lda, j.B, j.ldb, j.rows = _p(dY), dY.stride(0), _p(X), X.stride(0), rows
            j.C, j.ldc, j.colsum_a, j.colsum_ld = dW_ptr, N, db_ptr, M
            j.workspace, j.workspace_floats, j.rows_dev = _p(ws), nws, _p(rows_dev)
            j.M, j.N, j.c_transposed, j.colsum_repeat = M, N, 0, 1
        return jobs, src, keep

    def take_reduce(self):
        """The pending reduce plan for a launch that will carry it (or None); the caller reports with ``rode``."""
        r, self.pending_reduce = self.pending_reduce, None
        if r is not None:
            self.keep.append(r)
        return r

    def take_fold(self):
        f, self.pending_fold = self.pending_fold, None
        if f is not None:
            self.keep.append(f)
        return f

    def launch_pending_fold(self):
        f = self.take_fold()
        if f is not None:
            _lib.check(_lib.lib().kgw_readout_train_fold(C.byref(f[0]), _lib.stream_ptr()), 'kgw_readout_train_fold')

    def launch_pending_reduce(self):
        r = self.take_reduce()
        if r is not None:
            _lib.check(_lib.lib().kgw_tn_reduce_launch(C.byref(r[0]), _lib.stream_ptr()), 'kgw_tn_reduce_launch')

    def launch_pending_tail(self):
        """kgw_fold_bwd / kgw_relvec_bwd_multi left pending, as launches of their own, in that order (a consumer of their outputs is
        about to be launched, or the merged launch does not take them)."""
        L = _lib.lib()
        fb, rb = self.fold_bwd, self.relvec_bwd
        self.fold_bwd = self.relvec_bwd = None
        if fb is not None or rb is not None:
            self.launch_pending_reduce()                 # (they read the transform's weight gradient)
        if fb is not None:
            _lib.check(L.kgw_fold_bwd(C.byref(fb[0]), _lib.stream_ptr()), 'kgw_fold_bwd')
        if rb is not None:
            _lib.check(L.kgw_relvec_bwd_multi(rb[0], rb[1], _lib.stream_ptr()), 'kgw_relvec_bwd_multi')
        for t in (fb, rb):
            if t is not None:
                for ptr, numel, st in t[-1]:
                    self.records[ptr] = (None, numel, st)

    def flush(self):
        """Launch the deferred products (first launch only: their row blocks' partial sums are taken by the optimiser's launch) and,
        in the same launch, the parameter-only end of the backward pass (kgw_param_tail)."""
        L = _lib.lib()
        self.launch_pending_reduce()                     # (nothing came by to carry it: the tail reads what it finishes)
        self.launch_pending_fold()
        todo, self.products = sorted(self.products, key=lambda p: -p[0].shape[0]), []       # the tall ones first in the grid
        chunks = [todo[i:i + 4] for i in range(0, len(todo), 4)]
        fb, rb = self.fold_bwd, self.relvec_bwd
        if (fb is not None or rb is not None) and not chunks:
            chunks = [[]]
        for ci, chunk in enumerate(chunks):
            jobs, src, keep = self._tn_jobs(chunk)
            merged = False
            if ci == 0 and rb is not None:
                # (the fold's outputs must BE the inputs of one of the relation-vector jobs: that job's blocks then compute them)
                fold_job = -1
                if fb is not None:
                    a = fb[0]
                    fold_job = next((k for k in range(rb[0]) if rb[1][k].dU_full == a.dU and rb[1][k].dV == a.dV and
                                     rb[1][k].dw_src_acc == a.dws and not rb[1][k].duv_pieces), -1)
                if fb is None or fold_job >= 0:
                    rc = L.kgw_param_tail(len(chunk), jobs, src, C.byref(fb[0]) if fb is not None else None, rb[0], rb[1], fold_job,
                                          _lib.stream_ptr())
                    if rc != _lib.KGW_E_UNSUPPORTED:
                        _lib.check(rc, 'kgw_param_tail')
                        merged = True
                        self.tail_taken += 1
                        for t in (fb, rb):
                            if t is not None:
                                for ptr, numel, st in t[-1]:
                                    self.records[ptr] = (None, numel, st)
                        self._tail_keep = (fb, rb)           # (operands alive until the optimiser's launch is enqueued)
                        self.fold_bwd = self.relvec_bwd = None
            if ci == 0 and not merged:
                self.launch_pending_tail()
            if not merged and chunk:
                _lib.check(L.kgw_tn_gemm_multi_partial(len(chunk), jobs, src, _lib.stream_ptr()), 'kgw_tn_gemm_multi_partial')
            for q, (dY, X, dW_ptr, db_ptr, _sw, _sb, rows_dev) in enumerate(chunk):
                M, N = dY.shape[1], X.shape[1]
                for ptr, numel, rec_src in ((dW_ptr, M * N, src[2 * q]), (db_ptr, M, src[2 * q + 1])):
                    # (a product with one row block is complete -- KGW_GRAD_DIRECT -- but was still written to the address handed to
                    #  autograd: it gets a record WITHOUT a source, so that a gradient autograd copied instead of adopting is noticed)
                    rec = None
                    if rec_src.kind != 0:
                        rec = _lib.KgwGradSrc()
                        C.memmove(C.byref(rec), C.byref(rec_src), C.sizeof(rec))
                    self.records[ptr] = (rec, numel, keep[q])
        if self.fold_bwd is not None or self.relvec_bwd is not None:      # (a fold without relation vectors behind it)
            self.launch_pending_tail()

    def add(self, grad: torch.Tensor, src, ws: torch.Tensor):
        if src.kind != 0:           # (KGW_GRAD_DIRECT: the producer finished the tensor itself)
            rec = _lib.KgwGradSrc()
            C.memmove(C.byref(rec), C.byref(src), C.sizeof(rec))
            # (NOT the tensor: autograd must be able to steal it -- but its STORAGE, like defer_product: were autograd to clone the
            #  gradient and drop the original, the caching allocator could hand the block to another gradient of the same size,
            #  which would then inherit this record; with the block held a clone is noticed as a left-over record)
            self.records[grad.data_ptr()] = (rec, grad.numel(), ws, grad.untyped_storage())

    def take(self, grad: torch.Tensor):
        rec = self.records.pop(grad.data_ptr(), None)
        if rec is not None and rec[1] != grad.numel():
            raise GradSinkMismatch('a deferred gradient record does not describe the tensor found at its address')
        return rec


GRAD_SINK = None           # the GradSink of the backward pass being issued, or None (every product finishes its own sums)
_FUSED_ADAM = os.environ.get('KGW_FUSED_ADAM', '1') != '0'         # 0: k_tn_reduce / k_mlp2_bwd_fold / stats as launches of their own
# The SHORT weight-gradient products of the MLPs (gene 20 k rows, GO 2 x 7 k: 316 + 448 blocks, neither fills the chip's 512 slots)
# are not launched where their backward passes end but as ONE grouped launch ahead of the optimiser's (GradSink.defer_product):
# 29.7 us against 16.8 + 17.2 in two launches, step 1.1142 / 1.1173 against 1.1218 / 1.1201 ms (A/B on one box).  With the TALL
# product of the SNP MLP in the group too (round 4, knob removed) nothing is gained -- 1.0830 / 1.0838 against 1.0832 / 1.0834, the grouped
# launch 83 us against 16.8 + 43.7 + 16.7: its 512 blocks already fill the chip for 43 us and the short products' blocks of the later
# tiles queue behind them.  KGW_DEFER_PRODUCTS=0: every product where it is.
_DEFER_PRODUCTS = os.environ.get('KGW_DEFER_PRODUCTS', '1') != '0'
# The parameter-only END of the backward pass (kgw_fold_bwd 28 us, kgw_relvec_bwd_multi 9 us: neither fills the chip, each waits for the
# one before) is not launched where autograd reaches it but as blocks of the deferred products' launch (kgw_param_tail, round 5).
# KGW_PARAM_TAIL=0: every launch where it is.
_PARAM_TAIL = os.environ.get('KGW_PARAM_TAIL', '1') != '0'
# The SECOND launch of the relation transform's weight-gradient products (k_tn_reduce: 5 - 8 us for a few dozen blocks, once per
# layer) finishes gradients that nothing reads before the end of the backward pass: in a captured step it is not issued but left as
# a KgwTnReducePlan whose blocks ride in the next launch that can carry them -- layer 2's 1 536 blocks in layer 1's kgw_transform_bwd
# (round 5: -4 us, one launch less).  A plan of more than _RIDE_MAX_BLOCKS blocks is launched where it is: layer 1's 9 088 blocks riding
# in the SNP MLP's tall weight-gradient product took the carrier's registers and LDS and ran two per CU -- 78.5 us against 8.3 + 47.1.
# KGW_DEFER_REDUCE=0: every second launch where it is.
_DEFER_REDUCE = os.environ.get('KGW_DEFER_REDUCE', '1') != '0'
_RIDE_MAX_BLOCKS = 2048
# The resident first layer's backward writes d(pre-activation) directly as the kgw_gemm3 operand image of the weight gradient
# (kgw_mlp2_bwd_first_packed): the kgw_gemm3_pack launch (7 us) and the fp32 rows it read disappear.  KGW_PACK_FUSED=0: as before.
_PACK_FUSED = os.environ.get('KGW_PACK_FUSED', '1') != '0'
# The read-out node's second launch in a training step (one block that folds the partial sums into d w_lin, d b_lin and the loss: 5 us)
# is one more block of the launch that follows it in a captured step (layer 2's kgw_transform_bwd_ex).  KGW_DEFER_READOUT_FOLD=0: a
# launch of its own.
_DEFER_READOUT_FOLD = os.environ.get('KGW_DEFER_READOUT_FOLD', '1') != '0'
READOUT_FOLD_DEFERRED = False      # set by ``readout_fold_deferred`` (a trainer that ALWAYS runs the backward pass: GraphTrainStep)


class readout_fold_deferred:
    """Inside this scope the fused read-out node (``unit_grad=True``) leaves its fold -- and with it the loss value, d w_lin and
    d b_lin -- to the backward pass.  Outside (KGWAS.train_step, tests, any caller that may read or log the loss before calling
    backward, or never call it) the node finishes everything in its forward."""

    def __enter__(self):
        global READOUT_FOLD_DEFERRED
        self.prev, READOUT_FOLD_DEFERRED = READOUT_FOLD_DEFERRED, True
        return self

    def __exit__(self, *exc):
        global READOUT_FOLD_DEFERRED
        READOUT_FOLD_DEFERRED = self.prev
        return False


class grad_sink_scope:
    def __init__(self, sink):
        self.sink = sink

    def __enter__(self):
        global GRAD_SINK
        self.prev, GRAD_SINK = GRAD_SINK, self.sink
        return self.sink

    def __exit__(self, *exc):
        global GRAD_SINK
        GRAD_SINK = self.prev
        return False


_DUV_RIDERS = os.environ.get('KGW_DUV_RIDERS', '1') != '0'        # 0: d u_r / d v_r through the [d a_src | d a_dst] rows + product
_GEMM3 = os.environ.get('KGW_GEMM3', '1') != '0'                   # 0: the first gene Linear and its weight gradient on the library's fp32 product
_SHORT_ROWS = os.environ.get('KGW_SHORT_ROWS', '1') != '0'     # 0: every source row on the general path (timing experiments)


def _layer_args(batch, layer: int, neg_slope: float, inv_temp: float) -> KgwLayerArgs:
    dg, buf, m = batch.dg, batch.buf, batch.meta
    a = KgwLayerArgs()
    a.layer = layer
    a.n_chunks = int(m.n_chunks[layer - 1])
    a.n_multi_hops = min(dg.num_layers - layer, dg.n_hops - 1) + 1
    a.n_src_rows = int(m.src_base[layer - 1][dg.schema.NT])
    a.neg_slope, a.inv_temp = neg_slope, inv_temp
    a.graph_host = C.addressof(dg.kg)
    a.meta_host = C.addressof(m)
    a.meta_dev = _p(buf.meta)
    a.chunks = _p(buf.chunks)
    # hub rows (segments of more than one chunk) need the combine launches -- unless no destination row this layer can have is
    # one: the top layer of a minibatch only aggregates into the seeds' type (a static property of the graph)
    seeds_only = (not dg.full_graph) and a.n_multi_hops == 1 and batch.input_type is not None and \
        dg.schema.type_id[batch.input_type] not in dg.multi_dst_types
    if not seeds_only:
        a.multi = _p(buf.multi)
        a.multi_cap = dg.multi_cap
    a.col_local = _p(buf.col_local)
    a.t_ptr = _p(buf.t_ptr[layer - 1])
    a.t_edge = _p(buf.t_edge[layer - 1])
    a.t_zrow = _p(buf.t_zrow[layer - 1])
    if _SHORT_ROWS:
        a.t_rel = _p(buf.t_rel[layer - 1])          # relation id per src-major entry: the 8-rows-per-wavefront backward path
        a.oct_flags = _p(buf.t_cnt[layer - 1])      # (the histogram scratch holds the sampler's octet flags afterwards)
    return a


class _GatAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, H, U, V, batch, layer, neg_slope, inv_temp, raw_weights=False, relu_input=False, zbuf=None, lbias=None):
        dg, m = batch.dg, batch.meta
        NT = dg.schema.NT
        z_rows = int(m.z_base[layer - 1][NT])
        n_edges = int(m.n_edges[layer - 1])
        n_chunks = int(m.n_chunks[layer - 1])
        n_src = int(m.src_base[layer - 1][NT])
        H = H.contiguous(); U = U.contiguous(); V = V.contiguous()
        assert H.dtype == torch.float32 and H.shape == (n_src, KGW_C), (H.shape, n_src)
        assert U.shape == (dg.schema.NR, KGW_C) and V.shape == U.shape
        dev = H.device
        # Z, stat and the backward's d a_dst all start from zero (rows without edges are never visited): one fill
        zr = max(z_rows, 1)
        if zbuf is None:
            zbuf = torch.zeros(zr * (KGW_C + 3), device=dev)
        else:                                       # cleared by the caller (rel_vectors did it in its launch)
            assert zbuf.dtype == torch.float32 and zbuf.is_contiguous() and zbuf.numel() >= zr * (KGW_C + 3)
        Z = zbuf[:zr * KGW_C].view(zr, KGW_C)
        stat = zbuf[zr * KGW_C:zr * (KGW_C + 2)].view(zr, 2)
        ctx.da_dst = zbuf[zr * (KGW_C + 2):zr * (KGW_C + 3)]
        e_edge = torch.empty(max(n_edges, 1), device=dev)
        any_multi = batch.static or any(int(m.multi_cnt[h]) for h in range(dg.n_hops))
        part = torch.empty(max(n_chunks, 1) * PART_STRIDE if any_multi else 4, device=dev)
        a = _layer_args(batch, layer, neg_slope, inv_temp)
        a.H, a.V, a.U = _p(H), _p(V), _p(U)
        if lbias is not None:
            lbias = lbias.contiguous()
            assert lbias.dtype == torch.float32 and lbias.numel() == dg.schema.NR
            a.logit_bias = _p(lbias)
        ctx.has_lbias = lbias is not None
        a.flags = 1 if raw_weights else 0          # KGW_F_RAW_WEIGHTS
        ctx.raw_weights = raw_weights
        ctx.relu_input = relu_input
        a.Z, a.stat, a.e_edge, a.part = _p(Z), _p(stat), _p(e_edge), _p(part)
        TIMER.attach(a, 'fwd', layer, n_edges, z_rows, n_src)
        # SNP-sharded multi-GPU mode (kgwas_amd/shard.py): relations whose sources this rank holds only in part leave
        # partial softmax states, merged across the ranks before anything reads Z / stat
        xchg = getattr(batch, 'exchange', None)
        if xchg is not None and not raw_weights:
            a.partial_rels = xchg.mask[layer]
        _lib.check(_lib.lib().kgw_gat_aggregate_fwd(C.byref(a), _lib.stream_ptr()), 'kgw_gat_aggregate_fwd')
        if xchg is not None and not raw_weights and xchg.mask[layer]:
            xchg.forward(batch, layer, Z, stat)
        ctx.set_materialize_grads(False)          # no zero tensors for the (non-differentiable) stat / e_edge outputs
        ctx.save_for_backward(H, U, V, Z, stat, e_edge)
        ctx.batch, ctx.layer, ctx.neg_slope, ctx.inv_temp = batch, layer, neg_slope, inv_temp
        ctx.mark_non_differentiable(stat, e_edge)
        return stat, e_edge, Z[:z_rows]

    @staticmethod
    def backward(ctx, _dstat, _de, dZ):
        if ctx.raw_weights:
            raise RuntimeError('raw-logit aggregation (attention export) is inference only')
        if dZ is None:
            return (None,) * 11
        H, U, V, Z, stat, e_edge = ctx.saved_tensors
        batch, layer = ctx.batch, ctx.layer
        dg, m = batch.dg, batch.meta
        sc = dg.schema
        NT = sc.NT
        z_rows = int(m.z_base[layer - 1][NT])
        n_edges = int(m.n_edges[layer - 1])
        n_chunks = int(m.n_chunks[layer - 1])
        n_src = int(m.src_base[layer - 1][NT])
        t_rows = int(m.t_base[layer - 1][NT])
        dev = H.device
        dZf = dZ.contiguous() if z_rows else torch.zeros(1, KGW_C, device=dev)
        xchg = getattr(batch, 'exchange', None)
        if xchg is not None and z_rows and xchg.mask[layer] and not xchg.staged:
            # sharded mode: this rank's upstream gradient is partial; its own edges of the exchanged segments need the sum
            # (staged form: the trainer summed it before it got here, see gat_aggregate)
            dZf = xchg.backward(batch, layer, dZf)
        adp = torch.empty(max(n_edges, 1), 2, device=dev)
        da_dst, ctx.da_dst = ctx.da_dst, None            # zeroed with Z in forward; consumed once
        if da_dst is None:
            da_dst = torch.zeros(max(z_rows, 1), device=dev)
        part_da = torch.empty(4 * max(n_chunks, 1), device=dev)      # (per chunk of a hub row: the four sums of k_agg_bwd_dst)
        dH = torch.empty(max(n_src, 1), KGW_C, device=dev)
        ld_da = (sc.NR + 3) & ~3
        a = _layer_args(batch, layer, ctx.neg_slope, ctx.inv_temp)
        a.H, a.U, a.V, a.Z, a.stat, a.e_edge = _p(H), _p(U), _p(V), _p(Z), _p(stat), _p(e_edge)
        a.dZ, a.adp, a.da_dst, a.part_da = _p(dZf), _p(adp), _p(da_dst), _p(part_da)
        a.dH = _p(dH)
        if xchg is not None:
            a.partial_rels = xchg.mask[layer]      # (their rows' d a_dst stay plain sums over this rank's edges: kgw_gat_aggregate_bwd_dst)
        riders = _DUV_RIDERS and n_src > 0 and n_chunks > 0
        if riders:
            # d u_r / d v_r from per-chunk sums the dst-major pass leaves and extra blocks of the src-major launch -- no
            # [d a_src | d a_dst] row per node, no tall-skinny product over H afterwards
            part_du = torch.empty(n_chunks, KGW_C, device=dev)
            duv_ws = torch.empty(2 * sc.NR * 8 * KGW_C, device=dev)
            dU, dV = torch.empty(sc.NR, KGW_C, device=dev), torch.empty(sc.NR, KGW_C, device=dev)
            a.part_du, a.seg_chptr, a.duv_ws, a.dU, a.dV = _p(part_du), _p(batch.buf.seg_chptr), _p(duv_ws), _p(dU), _p(dV)
        else:
            da_src = torch.empty(max(n_src, 1), 2 * ld_da, device=dev)   # [node row, (d a_src | d a_dst) by relation id]
            a.da_src = _p(da_src)
        a.flags = 2 if ctx.relu_input else 0       # KGW_F_RELU_INPUT: fold the ReLU that produced H into dH
        pieces = riders and DUV_PIECES is not None
        if pieces:
            a.flags |= 4                           # KGW_F_DUV_PIECES: dU / dV stay UNWRITTEN, their consumers add the pieces
            DUV_PIECES[dU.data_ptr()] = (duv_ws.data_ptr(), duv_ws)
            DUV_PIECES[dV.data_ptr()] = (duv_ws.data_ptr() + sc.NR * 8 * KGW_C * 4, duv_ws)
        L = _lib.lib()
        TIMER.attach(a, 'bwd_dst', layer, n_edges, z_rows, n_src)
        _lib.check(L.kgw_gat_aggregate_bwd_dst(C.byref(a), _lib.stream_ptr()), 'kgw_gat_aggregate_bwd_dst')
        TIMER.attach(a, 'bwd_src', layer, n_edges, z_rows, n_src)
        dlb = None
        if ctx.has_lbias:
            # d(constant of relation r) = sum of d pre-activation over ALL its edges = the column sums of d a_dst: extra
            # blocks of the src-major launch
            dlb = torch.empty(sc.NR, device=dev)
            a.rel_sums = _p(dlb)
        _lib.check(L.kgw_gat_aggregate_bwd_src(C.byref(a), _lib.stream_ptr()), 'kgw_gat_aggregate_bwd_src')
        if riders:
            pass
        elif n_src:
            # d u_r = sum_j d a_src[j, r] H[j], d v_r = sum_i d a_dst[i, r] H[i]: all relations, both sides, as ONE
            # tall-skinny product over H
            dUV = tn_gemm(da_src[:n_src], H[:n_src])
            dU, dV = dUV[:sc.NR], dUV[ld_da:ld_da + sc.NR]
        else:
            dU, dV = torch.zeros_like(U), torch.zeros_like(V)
        if ctx.has_lbias and not n_src:                  # (no source rows: the launch above did not run)
            _lib.check(L.kgw_relation_sums(C.byref(a), _p(da_dst), _p(dlb), _lib.stream_ptr()), 'kgw_relation_sums')
        return dH[:n_src], dU, dV, None, None, None, None, None, None, None, dlb


def aggregate_workspace(batch, layer: int, device) -> torch.Tensor:
    """UNINITIALISED Z / stat / d a_dst workspace of ``gat_aggregate(batch, layer, ...)``: hand it to ``rel_vectors(...,
    zero=ws)`` (which clears it in its own launch) and then to ``gat_aggregate(..., zbuf=ws)``."""
    zr = max(int(batch.meta.z_base[layer - 1][batch.dg.schema.NT]), 1)
    return torch.empty(zr * (KGW_C + 3) + (-(zr * (KGW_C + 3))) % 4, device=device)


def gat_aggregate(batch, layer: int, H: torch.Tensor, U: torch.Tensor, V: torch.Tensor,
                  neg_slope: float = 0.2, temperature: float = 1.0, raw_weights: bool = False,
                  relu_input: bool = False, zbuf: torch.Tensor = None, logit_bias: torch.Tensor = None):
    """Z[i, r] = sum_j softmax_j(leaky_relu(<H_src[j], u_r> + <H_dst[i], v_r>) / T) H_src[j] for every live relation
    of the layer.  H [n_src_rows,128]: layer input, type-major (``meta.src_base``; a destination node is row i of
    its own type's block); U, V [n_rels,128] by relation id.  Returns (Z [z_rows,128], stat [z_rows,2] =
    (row max, denominator), e_edge [n_edges]); Z is type-major: the block of destination type t starts at row
    ``meta.z_base[layer-1][t]`` and holds ``lay_rows * R_dst[t]`` rows ([row, relation slot, 128]).
    ``raw_weights``: Z[i, r] = sum_j e_ij H_src[j] with e the leaky_relu logits, no softmax (inference only; what
    the reference's attention export propagates, kgwas/utils.py:446-461 + conv.py:221-228).
    ``relu_input``: every row of H is the output of a ReLU (the previous layer, model.py:75) and the node that
    produced it expects its incoming gradient ALREADY multiplied by (H > 0): the source-side backward does it while
    writing dH (see layer_transform's ``premasked``).  ``zbuf``: an ``aggregate_workspace`` that is ALREADY zero.
    ``logit_bias`` [n_rels]: constant added to the pre-activation logit of every edge of a relation (FC_output folded
    into layer 1, see fold_fc_output_hip); differentiable."""
    stat, e_edge, Z = _GatAggregate.apply(H, U, V, batch, layer, float(neg_slope), 1.0 / float(temperature), raw_weights,
                                          relu_input, zbuf, logit_bias)
    xchg = getattr(batch, 'exchange', None)
    if xchg is not None and xchg.staged and xchg.mask[layer] and Z.requires_grad:
        # SNP-sharded mode, step captured in segments: cut the autograd graph at the merged Z -- the trainer sums the gradient
        # of this leaf over the ranks (a collective BETWEEN two graph segments) and continues the backward from Z
        Zx = Z.detach().requires_grad_()
        xchg.cuts.append((layer, Z, Zx))
        Z = Zx
    return Z, stat, e_edge


gat_aggregate_flat = gat_aggregate


def edge_alpha(batch, layer: int, stat: torch.Tensor, e_edge: torch.Tensor, temperature: float = 1.0):
    """alpha per local edge of one layer from the saved softmax statistics."""
    a = _layer_args(batch, layer, 0.2, 1.0 / float(temperature))
    n_edges = int(batch.meta.n_edges[layer - 1])
    out = torch.zeros(max(n_edges, 1), device=e_edge.device)
    a.stat, a.e_edge = _p(stat), _p(e_edge)
    _lib.check(_lib.lib().kgw_edge_alpha(C.byref(a), C.c_void_p(out.data_ptr()), _lib.stream_ptr()), 'kgw_edge_alpha')
    return out[:n_edges]


# ------------------------------------------------------------------------------------------------------
# dense helpers on the split-K MFMA kernel (kgw_tn_gemm)
# ------------------------------------------------------------------------------------------------------
_TN_MIN_ROWS = 1024       # below this a library GEMM is fine


def tn_gemm(A: torch.Tensor, B: torch.Tensor, colsum: bool = False, out: torch.Tensor = None,
            transpose_out: bool = False, colsum_out: torch.Tensor = None, rows_dev: torch.Tensor = None, defer: bool = False):
    """C = A^T @ B for tall row-major A [rows, M], B [rows, N] (fp32, inner stride 1); optionally also the
    column sums of A.  Deterministic split-K on fp32 MFMA.  ``out``: write C (or C^T with ``transpose_out``) into
    this row-major 2-D view instead of a new tensor; ``colsum_out`` [q, M]: write the column sums into each of its
    q rows."""
    assert A.dim() == 2 and B.dim() == 2 and A.shape[0] == B.shape[0]
    assert A.dtype == torch.float32 and B.dtype == torch.float32
    if A.stride(1) != 1:
        A = A.contiguous()
    if B.stride(1) != 1:
        B = B.contiguous()
    rows, M = A.shape
    N = B.shape[1]
    dev = A.device
    if out is None:
        out = torch.empty((N, M) if transpose_out else (M, N), device=dev)
    assert out.shape == ((N, M) if transpose_out else (M, N)) and out.stride(1) == 1
    cs = None
    rep, cs_ld = 1, M
    if colsum_out is not None:
        assert colsum_out.dim() == 2 and colsum_out.shape[1] == M and colsum_out.stride(1) == 1
        cs, rep, cs_ld = colsum_out, colsum_out.shape[0], colsum_out.stride(0)
    elif colsum:
        cs = torch.empty(M, device=dev)
    if rows == 0:
        out.zero_()
        if cs is not None:
            cs.zero_()
        return (out, cs) if (colsum or colsum_out is not None) else out
    L = _lib.lib()
    nws = int(L.kgw_tn_gemm_workspace_floats(rows, M, N))
    ws = torch.empty(nws, device=dev)
    sink = GRAD_SINK
    if defer and sink is not None and colsum_out is None and out.is_contiguous():
        # ``defer``: the caller hands (out, cs) to autograd as the gradients of two parameters and nothing else reads them before
        # the optimiser: the row blocks' partial sums are added inside its launch (GradSink)
        src = (_lib.KgwGradSrc * 2)()
        ride = sink.take_reduce()
        if ride is not None:           # (a product group's pending second launch rides in this product's: GradSink.pending_reduce)
            _lib.check(L.kgw_tn_gemm_partial_ride(_p(A), A.stride(0), M, _p(B), B.stride(0), N, rows, _p(out), out.stride(0),
                                                  1 if transpose_out else 0, _p(cs), _p(ws), nws, _p(rows_dev), src, C.byref(ride[0]),
                                                  _lib.stream_ptr()), 'kgw_tn_gemm_partial_ride')
            sink.reduces_ridden += 1
        else:
            _lib.check(L.kgw_tn_gemm_partial(_p(A), A.stride(0), M, _p(B), B.stride(0), N, rows, _p(out), out.stride(0),
                                             1 if transpose_out else 0, _p(cs), _p(ws), nws, _p(rows_dev), src, _lib.stream_ptr()),
                       'kgw_tn_gemm_partial')
        sink.add(out, src[0], ws)
        if cs is not None:
            sink.add(cs, src[1], ws)
        return (out, cs) if colsum else out
    _lib.check(L.kgw_tn_gemm_ex(_p(A), A.stride(0), M, _p(B), B.stride(0), N, rows, _p(out), out.stride(0),
                                1 if transpose_out else 0, _p(cs), rep, cs_ld, _p(ws), nws, _p(rows_dev), _lib.stream_ptr()),
               'kgw_tn_gemm_ex')
    return (out, cs) if (colsum or colsum_out is not None) else out


def _tn_gemm_group(jobs) -> bool:
    """Several C^T = (A^T B)^T products with column sums of A (``jobs``: [(A [rows, M], B [rows, N], out [N, M] row-major view,
    colsum_out [q, M])]) in one kgw_tn_gemm_multi launch pair.  False: not taken (one job, shapes outside the grouped tiling)."""
    if not (2 <= len(jobs) <= 4):
        return False
    for A, B, out, cs in jobs:
        if not (A.dtype == torch.float32 and B.dtype == torch.float32 and A.stride(1) == 1 and B.stride(1) == 1 and out.stride(1) == 1 and
                cs.stride(1) == 1 and A.shape[0] == B.shape[0] and A.shape[0] > 0 and A.shape[1] % 2 == 0 and B.shape[1] % 2 == 0 and
                A.shape[1] >= 64 and B.shape[1] >= 64 and A.stride(0) % 2 == 0 and B.stride(0) % 2 == 0 and A.data_ptr() % 8 == 0 and
                B.data_ptr() % 8 == 0 and out.shape == (B.shape[1], A.shape[1]) and cs.shape[1] == A.shape[1]):
            return False
    L = _lib.lib()
    arr = (_lib.KgwTnJob * len(jobs))()
    keep = []
    for q, (A, B, out, cs) in enumerate(jobs):
        rows, M = A.shape
        N = B.shape[1]
        nws = int(L.kgw_tn_gemm_workspace_floats(rows, M, N))
        ws = torch.empty(nws, device=A.device)
        keep.append(ws)
        j = arr[q]
        j.A, j.lda, j.B, j.ldb, j.rows = _p(A), A.stride(0), _p(B), B.stride(0), rows
        j.C, j.ldc, j.colsum_a, j.colsum_ld = _p(out), out.stride(0), _p(cs), cs.stride(0)
        j.workspace, j.workspace_floats, j.rows_dev = _p(ws), nws, None
        j.M, j.N, j.c_transposed, j.colsum_repeat = M, N, 1, cs.shape[0]
    _lib.check(L.kgw_tn_gemm_multi(len(jobs), arr, _lib.stream_ptr()), 'kgw_tn_gemm_multi')
    return True


def weight_grads(pairs, rows_dev: torch.Tensor = None):
    """[(dW [out,in], db [out])] of Y = X W^T + b for several (dY [rows,out], X [rows,in]) pairs that are available
    TOGETHER (the Linears of one MLP at the end of its backward): one kgw_tn_gemm_multi launch pair for all of them
    instead of one pair each; falls back to per-product calls where the grouped kernel does not apply."""
    pairs = [(dY if dY.stride(1) == 1 else dY.contiguous(), X if X.stride(1) == 1 else X.contiguous()) for dY, X in pairs]
    if GRAD_SINK is not None and _DEFER_PRODUCTS:          # (not launched now: GradSink.defer_product)
        return [linear_weight_grad(dY, X, rows_dev=rows_dev) for dY, X in pairs]
    # (tall products fill the chip on their own: grouping only pays while a product is launch-bound)
    ok = 1 < len(pairs) <= 4 and all(_TN_MIN_ROWS <= X.shape[0] < 32768 and X.shape[1] <= 1024 and X.shape[1] % 2 == 0 and dY.shape[1] % 2 == 0 and
                                     X.stride(0) % 2 == 0 and dY.stride(0) % 2 == 0 and X.data_ptr() % 8 == 0 and dY.data_ptr() % 8 == 0
                                     for dY, X in pairs)
    if not ok:
        return [linear_weight_grad(dY, X, rows_dev=rows_dev) for dY, X in pairs]
    L = _lib.lib()
    jobs = (_lib.KgwTnJob * len(pairs))()
    outs, keep = [], []
    for q, (dY, X) in enumerate(pairs):
        rows, M = dY.shape
        N = X.shape[1]
        dW = torch.empty(M, N, device=dY.device)
        db = torch.empty(M, device=dY.device)
        nws = int(L.kgw_tn_gemm_workspace_floats(rows, M, N))
        ws = torch.empty(nws, device=dY.device)
        keep.append(ws)
        j = jobs[q]
        j.A, j.lda, j.B, j.ldb, j.rows = _p(dY), dY.stride(0), _p(X), X.stride(0), rows
        j.C, j.ldc, j.colsum_a, j.colsum_ld = _p(dW), N, _p(db), M
        j.workspace, j.workspace_floats, j.rows_dev = _p(ws), nws, _p(rows_dev)
        j.M, j.N, j.c_transposed, j.colsum_repeat = M, N, 0, 1
        outs.append((dW, db))
    sink = GRAD_SINK
    if sink is not None:            # (every caller returns these to autograd as parameter gradients: see tn_gemm(defer=True))
        src = (_lib.KgwGradSrc * (2 * len(pairs)))()
        _lib.check(L.kgw_tn_gemm_multi_partial(len(pairs), jobs, src, _lib.stream_ptr()), 'kgw_tn_gemm_multi_partial')
        for q, (dW, db) in enumerate(outs):
            sink.add(dW, src[2 * q], keep[q])
            sink.add(db, src[2 * q + 1], keep[q])
        return outs
    _lib.check(L.kgw_tn_gemm_multi(len(pairs), jobs, _lib.stream_ptr()), 'kgw_tn_gemm_multi')
    return outs


def _library_linear(X, W, bias, relu, mask, w_kn, out, fixed_shape):
    LIBRARY_GEMM.note('linear', X.shape[0], X.shape[1], W.shape[1] if w_kn else W.shape[0])
    Wop = W if w_kn else W.t()
    if fixed_shape and mask is None and out is None:          # same shape every step: tuned library solution
        with _TUNED:
            if bias is not None and relu:
                return torch._addmm_activation(bias, X, Wop)          # bias + ReLU in the GEMM epilogue
            Y = torch.addmm(bias, X, Wop) if bias is not None else X @ Wop
        return torch.relu_(Y) if relu else Y
    if bias is not None and relu and X.is_cuda:
        Y = torch._addmm_activation(bias, X, Wop, out=out) if (out is not None and mask is None) else \
            torch._addmm_activation(bias, X, Wop)
        if mask is None:
            return Y
        relu = False
    elif mask is None and out is not None:
        Y = torch.addmm(bias, X, Wop, out=out) if bias is not None else torch.mm(X, Wop, out=out)
        return torch.relu_(Y) if relu else Y
    else:
        Y = torch.addmm(bias, X, Wop) if bias is not None else X @ Wop
    if relu:
        Y = torch.relu_(Y)
    if mask is not None:
        Y = Y * (mask > 0)
    if out is not None:
        out.copy_(Y)
        return out
    return Y


_LIN_MAX_K = 2304        # wider reductions (the 5120 / 57742-wide gene layer) go to the library GEMM
_SPLITK_MAX_ROWS = 8192     # below: kgw_linear_splitk


def linear(X: torch.Tensor, W: torch.Tensor, bias=None, relu: bool = False, mask=None, w_kn: bool = False,
           out: torch.Tensor = None, fixed_shape: bool = False, rows_dev: torch.Tensor = None):
    """Y = act(X @ Wop + bias) * (mask > 0) on the fp32-MFMA kernel (kgw_linear); Wop = W^T for W [N,K]
    (nn.Linear forward) or W for W [K,N] (w_kn: the dX product).  Shapes the kernel does not take
    (K or leading dimensions not multiples of 4, very wide K) run on the library GEMM with identical math.
    ``rows_dev``: device int32 holding the number of rows that are real when X is padded to a static capacity (a
    captured step): the kernel skips the padding rows and writes zeros there."""
    rows, K = X.shape
    N = W.shape[1] if w_kn else W.shape[0]
    # few rows, K and N multiples of 128: the per-relation transform of a layer and its dZ twin at the shapes a 512-seed
    # batch has (~1.2 k gene rows x 17 relations, 512 SNP rows x 6) -- one wavefront per (row tile, column tile, relation slab)
    if (0 < rows < _SPLITK_MAX_ROWS and K % 128 == 0 and N % 128 == 0 and (K == 128 or N == 128) and mask is None
            and X.dtype == torch.float32
            and X.stride(1) == 1 and W.stride(1) == 1 and X.stride(0) % 4 == 0 and W.stride(0) % 4 == 0
            and X.data_ptr() % 16 == 0 and W.data_ptr() % 16 == 0 and (bias is None or bias.data_ptr() % 16 == 0)
            and (out is None or (out.stride(1) == 1 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0))):
        Y = torch.empty(rows, N, device=X.device) if out is None else out
        assert Y.shape == (rows, N)
        L = _lib.lib()
        nws = int(L.kgw_linear_splitk_workspace_floats(rows, K, N))
        ws = torch.empty(nws, device=X.device) if nws else None
        _lib.check(L.kgw_linear_splitk(_p(X), X.stride(0), _p(W), W.stride(0), _p(bias), _p(Y), Y.stride(0), rows, K, N,
                                       1 if relu else 0, 1 if w_kn else 0, _p(ws), nws, _p(rows_dev), _lib.stream_ptr()),
                   'kgw_linear_splitk')
        return Y
    # (the package's own kernel whenever it CAN run; with KGW_ALLOW_LIBRARY=1 the row thresholds below apply instead)
    big = LIBRARY_GEMM.own_first or rows >= 8192 or (rows >= 4096 and K <= 128 and N <= 128)
    if K % 4 and K <= _LIN_MAX_K and X.dtype == torch.float32 and big and rows > 0:
        # a reduction length that is not a multiple of 4 (the 70-wide mode='full' SNP features, kgwas_data.py:167): zero-pad it --
        # an elementwise copy of X and of the (small) weight, then this package's kernel; never the library for this
        Kp = (K + 3) & ~3
        X = torch.nn.functional.pad(X, (0, Kp - K))
        W = torch.nn.functional.pad(W, (0, 0, 0, Kp - K)) if w_kn else torch.nn.functional.pad(W, (0, Kp - K))
        K = Kp
    ok = (X.dtype == torch.float32 and X.stride(1) == 1 and W.stride(1) == 1 and K % 4 == 0 and K <= _LIN_MAX_K
          and X.stride(0) % 4 == 0 and W.stride(0) % 4 == 0 and X.data_ptr() % 16 == 0 and W.data_ptr() % 16 == 0
          and (not w_kn or N % 4 == 0) and (mask is None or mask.stride(1) == 1)
          # 128-row tiles, no split over K: problems with few row tiles stay on the library (measured at 1171 rows,
          # K = 128, N = 1536 / 2176 -- the dZ product of a layer transform: 9 / 18.5 us there vs 23 / 24 us here)
          and big)
    if not ok:
        Y = _library_linear(X, W, bias, relu, mask, w_kn, out, fixed_shape)
        if rows_dev is not None:          # padding rows of a static layout: zeros, whatever the inputs held there
            Y.masked_fill_((torch.arange(rows, device=Y.device) >= rows_dev).unsqueeze(1), 0.0)
        return Y
    Y = torch.empty(rows, N, device=X.device) if out is None else out
    assert Y.shape == (rows, N) and Y.stride(1) == 1
    if rows:
        _lib.check(_lib.lib().kgw_linear(_p(X), X.stride(0), _p(W), W.stride(0), _p(bias), _p(mask),
                                         mask.stride(0) if mask is not None else 0, _p(Y), Y.stride(0), rows, K, N,
                                         1 if relu else 0, 1 if w_kn else 0, _p(rows_dev), _lib.stream_ptr()), 'kgw_linear')
    return Y


def gemm3_ok(M: int, K: int) -> bool:
    return _GEMM3 and K % 32 == 0 and M >= 2048 and K >= 1024


# Operand images that somebody ELSE keeps current: {data_ptr of an nn.Linear weight [128, K]: its packed image}, set (packed_scope)
# by a trainer whose optimiser launch rewrites the image whenever it updates the weight (kgw_adam_fused, KgwGradSrc.packed) and
# which re-packs it itself when the weight is changed behind its back (GraphTrainStep._refresh_images).  Inside the scope
# gemm3_pack returns the image without a launch.  PACK_PROBE: a list while the trainer finds out which weights are packed at all.
PACKED_W = None
PACK_PROBE = None


class packed_scope:
    def __init__(self, images, probe=None):
        self.images, self.probe = images, probe

    def __enter__(self):
        global PACKED_W, PACK_PROBE
        self.prev = (PACKED_W, PACK_PROBE)
        PACKED_W, PACK_PROBE = self.images, self.probe

    def __exit__(self, *exc):
        global PACKED_W, PACK_PROBE
        PACKED_W, PACK_PROBE = self.prev
        return False


class ParamRiders:
    """The parameter-only forward launches of a training step -- the relation vectors of all layers (kgw_relvec_fwd_multi) and the
    FC_output fold (kgw_fold_fwd) -- handed to the first gene Linear's kgw_gemm3 launch as RIDER blocks (kgw_gemm3_riders: they run on
    the compute units that product leaves idle, ~20 us off the step's critical path and two launches fewer).  While a queue is
    active (``param_riders_scope``: HeteroGNN.forward_loss computes the layers' parameters BEFORE the feature MLPs), the two
    autograd nodes allocate their outputs, record their job structs here and do not launch; ``gemm3`` takes what is pending with
    the layer's forward product; ``flush`` -- called before anything reads the outputs -- launches whatever no product took (a
    model whose gene layer does not take the resident route, a product without idle compute units) the ordinary way."""

    def __init__(self):
        self.relvec = None          # (n, jobs array, keep-alive)
        self.fold = None            # (KgwFoldArgs, fold_job, keep-alive)
        self.taken = 0              # launches that rode along (tests / bench)

    def pending(self):
        return self.relvec is not None or self.fold is not None

    def take(self):
        r, f = self.relvec, self.fold
        self.relvec = self.fold = None
        return r, f

    def flush(self):
        r, f = self.take()
        if r is not None:
            _lib.check(_lib.lib().kgw_relvec_fwd_multi(r[0], r[1], _lib.stream_ptr()), 'kgw_relvec_fwd_multi')
        if f is not None:
            _lib.check(_lib.lib().kgw_fold_fwd(C.byref(f[0]), _lib.stream_ptr()), 'kgw_fold_fwd')


PARAM_RIDERS = None        # the ParamRiders of the forward pass being issued, or None (every launch where it is)
_G3_RIDERS = os.environ.get('KGW_G3_RIDERS', '1') != '0'          # 0: relvec / fold as launches of their own (A/B, fallback tests)


class param_riders_scope:
    def __init__(self, q):
        self.q = q

    def __enter__(self):
        global PARAM_RIDERS
        self.prev, PARAM_RIDERS = PARAM_RIDERS, self.q
        return self.q

    def __exit__(self, *exc):
        global PARAM_RIDERS
        q, PARAM_RIDERS = PARAM_RIDERS, self.prev
        if q is not None and exc[0] is None:
            q.flush()
        return False


def gemm3_pack(S: torch.Tensor, K: int, s_is_kn: bool, k_valid: int = None, out: torch.Tensor = None) -> torch.Tensor:
    """B [K, 128] of a kgw_gemm3 product, split into its three bf16 pieces in the kernel's operand image.  ``s_is_kn``: S is
    B itself ([k_valid, 128]); else S = B^T ([128, k_valid], an nn.Linear weight).  ``k_valid`` < K (default K): S stops there,
    the rows of B up to K -- a multiple of 32 -- are zero.  ``out``: write the image there."""
    kv = K if k_valid is None else int(k_valid)
    assert S.dtype == torch.float32 and S.stride(1) == 1 and (S.shape == (kv, KGW_C) if s_is_kn else S.shape == (KGW_C, kv))
    L = _lib.lib()
    if not s_is_kn and kv == K and S.is_contiguous() and out is None:
        if PACKED_W is not None:
            img = PACKED_W.get(S.data_ptr())
            if img is not None:
                return img                                  # (kept current by the scope's owner: no launch)
        if PACK_PROBE is not None:
            PACK_PROBE.append(S)
    packed = out if out is not None else torch.empty(int(L.kgw_gemm3_packed_bytes(K)), dtype=torch.uint8, device=S.device)
    assert packed.numel() == int(L.kgw_gemm3_packed_bytes(K)) and packed.dtype == torch.uint8
    _lib.check(L.kgw_gemm3_pack(_p(S), S.stride(0), K, kv, 1 if s_is_kn else 0, _p(packed), _lib.stream_ptr()), 'kgw_gemm3_pack')
    return packed


def gemm3_tile(A: torch.Tensor) -> torch.Tensor:
    """A [M, K] -> its 32 x 32-tiled copy [ceil(M / 32), K / 32, 32, 32] (rows past M zero): the layout kgw_gemm3 reads fastest."""
    M, K = A.shape
    assert K % 32 == 0
    Mp = (M + 31) // 32 * 32
    if Mp != M:
        A = torch.cat([A, A.new_zeros(Mp - M, K)])
    return A.view(Mp // 32, 32, K // 32, 32).permute(0, 2, 1, 3).contiguous()


def gemm3(A: torch.Tensor, packed: torch.Tensor, bias=None, relu: bool = False, transpose_out: bool = False, out=None, tiled_rows: int = 0,
          row_map=None, out_rows=None, out_rows_real=None, defer: bool = False):
    """act(A [M, K] @ B [K, 128] + bias) -> [M, 128], or its transpose [128, M] (``transpose_out``): the tall resident product
    of the first gene Linear on the bf16 matrix pipe with fp32 error (three exact bf16 pieces per operand, kgw_gemm3).
    ``tiled_rows`` = M when A is a gemm3_tile() copy.  ``row_map`` [M] int32 + ``out_rows``: row m also goes to
    out_rows[row_map[m]] where row_map[m] >= 0 (the batch's rows of a resident layer output, no gather launch);
    ``out_rows_real`` (device int32): how many rows of ``out_rows`` are the batch's -- the rest (padding of a static layout) is
    zeroed by the same launch."""
    if tiled_rows:
        M, K, lda = tiled_rows, A.shape[1] * 32, 0
        assert A.dim() == 4 and A.shape[2:] == (32, 32) and A.is_contiguous() and A.shape[0] * 32 >= M
    else:
        M, K = A.shape
        lda = A.stride(0)
        assert A.stride(1) == 1
    assert A.dtype == torch.float32
    L = _lib.lib()
    if out is None:
        out = torch.empty((KGW_C, M) if transpose_out else (M, KGW_C), device=A.device)
    nws = int(L.kgw_gemm3_workspace_floats(M, K))
    ws = torch.empty(nws, device=A.device)
    _route('kgw_gemm3')
    sink = GRAD_SINK
    if defer and sink is not None and transpose_out and M % 32 == 0 and out.is_contiguous() and bias is None and not relu and row_map is None:
        # ``defer``: ``out`` goes to autograd as a parameter's gradient and to nothing else: the K ranges are added tile by tile
        # inside the optimiser's launch (GradSink, KGW_GRAD_G3T)
        src = _lib.KgwGradSrc()
        _lib.check(L.kgw_gemm3_partial(_p(A), lda, M, K, _p(packed), _p(ws), nws, _p(out), out.stride(0), C.byref(src), _lib.stream_ptr()),
                   'kgw_gemm3_partial')
        sink.add(out, src, ws)
        return out
    args = (_p(A), lda, M, K, _p(packed), _p(ws), nws, _p(bias), 1 if relu else 0, _p(out), out.stride(0),
            1 if transpose_out else 0, _p(row_map), _p(out_rows), out_rows.stride(0) if out_rows is not None else 0,
            out_rows.shape[0] if (out_rows is not None and out_rows_real is not None) else 0,
            _p(out_rows_real) if out_rows is not None else None)
    q = PARAM_RIDERS
    if q is not None and q.pending() and not transpose_out and int(L.kgw_gemm3_rider_blocks(M, K)) > 0:
        r, f = q.take()                     # (the step's parameter-only forward work rides on this launch's idle compute units)
        rc = L.kgw_gemm3_riders(*args, r[0] if r is not None else 0, r[1] if r is not None else None,
                                C.byref(f[0]) if f is not None else None, f[1] if f is not None else 0, _lib.stream_ptr())
        if rc != _lib.KGW_E_UNSUPPORTED:
            _lib.check(rc, 'kgw_gemm3_riders')
            q.taken += 1
            return out
        # (the riders' tables did not fit this launch -- an unaligned clear, a fold that is not the relation job's: nothing was
        #  launched.  The jobs go back into the queue, whose flush launches their own kernels, and the product runs plain)
        q.relvec, q.fold = r, f
    _lib.check(L.kgw_gemm3(*args, _lib.stream_ptr()), 'kgw_gemm3')
    return out


_RESIDENT_T = {}


def _resident_copies(X: torch.Tensor):
    """(A operand of the forward, X^T) for a resident feature matrix, built once per matrix (and again if it is modified in
    place: the tensor's version counter is part of the key).  X^T [K, Np] is the A operand of the weight-gradient product on
    kgw_gemm3 (a second resident copy, 0.4 GB for the 5 120-wide gene features; row-major like X -- the 32 x 32-tiled layout the
    kernel also takes, gemm3_tile, measured the same in isolation and in the step), its N node columns padded with zeros to a
    multiple of 32 (the product's reduction runs over the nodes; the real KG's gene count need not be one -- SURVEY 8d gives
    20 032 only as a lower bound -- and zero columns against zero rows of the packed dz leave dW exact).  The forward reads X
    itself when its width is a multiple of 32 (5 120), else a copy padded with zero columns (57 742 -> 57 760, mode='full')."""
    key = (X.data_ptr(), tuple(X.shape), X.device, X._version)
    ent = _RESIDENT_T.get(key)
    if ent is not None and ent[0]() is not None:         # the tensor the copies were made from is alive: same memory, same features
        return (X if ent[1][0] is None else ent[1][0]), ent[1][1]
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError('resident copies requested inside a graph capture: run one eager step first')
    for k in [k for k, e in _RESIDENT_T.items() if e[0]() is None or (k[0] == key[0] and k[3] != key[3])]:
        del _RESIDENT_T[k]                               # copies of matrices that are gone (their address may be reused) or changed
    N, K = X.shape
    Kp, Np = (K + 31) // 32 * 32, (N + 31) // 32 * 32
    direct = Kp == K and X.stride(0) % 4 == 0 and X.data_ptr() % 16 == 0
    xt = X.t().contiguous() if Np == N else torch.nn.functional.pad(X.t(), (0, Np - N))
    t = (None if direct else torch.nn.functional.pad(X, (0, Kp - K)), xt)      # (no strong reference to X itself)
    _RESIDENT_T[key] = (weakref.ref(X), t)
    return (X if t[0] is None else t[0]), t[1]


def _resident_ok(X: torch.Tensor, W: torch.Tensor) -> bool:
    M, K = X.shape
    return (gemm3_ok(M, (K + 31) // 32 * 32) and W.shape[0] == KGW_C and X.dtype == torch.float32
            and X.stride(1) == 1 and W.stride(1) == 1)


class GeneLayerShard:
    """The first Linear over a RESIDENT wide feature matrix (the gene features: [20 032, 5 120] x [5 120, 128], forward and
    weight gradient -- the two largest products of a step, kgwas/model.py:13,19) split by node rows over the ranks of a
    multi-GPU job.  Its output does not depend on the batch and the weights are replicated, so every rank would compute the
    SAME 2 x 26 GFLOP; instead rank p computes rows [p chunk, (p + 1) chunk):
        forward    h[rows_p] = relu(X[rows_p] W^T + b)              then ALL-GATHER of h (10 MB in total)
        backward   dz summed over the ranks' batches by REDUCE-SCATTER (rank p receives the sum of rows_p), then
                   dW_p = dz_sum[rows_p]^T X[rows_p]  -- a PARTIAL of the weight gradient: the sum over ranks of dW_p is the sum
                   over ranks of the full gradients, which is what the parameter-gradient all-reduce that follows computes
                   anyway (SUM in the SNP-sharded mode, AVG = SUM / world in the seed-parallel mode: both linear).
    ``inline``: the collectives are issued inside the autograd node (eager steps: kgwas_amd/shard.py, KGWAS.train_step);
    otherwise the caller runs the stages itself around captured graphs (kgwas_amd/graph_step.py): ``forward_partial`` ->
    ``gather`` -> (the node reads ``h_all``, leaves ``dz``) -> ``scatter`` -> ``weight_grad_partial``."""

    def __init__(self, rank: int, world: int, group=None, inline: bool = True):
        import torch.distributed as dist
        self.rank, self.world, self.group, self.inline = int(rank), int(world), group, inline
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        if self.backend == 'fake':                              # (bench.py --as-rank: collectives that move nothing)
            self.backend = 'nccl'
        self.h_all = self.dz = self.dz_mine = None
        self.n = self.chunk = 0
        self.last = None                                       # (X, W, b) of the layer this shard was last applied to
        self.bytes = {'all_gather(first gene layer output)': [0, 0], 'reduce_scatter(first gene layer dz)': [0, 0]}

    def selftest(self, device) -> bool:
        """The two collectives of the split on a few KB, checked against what they must deliver, the verdict agreed over the
        ranks (MIN): an in-place all-gather or a reduce-scatter that misbehaves on some backend / topology switches the split
        off on EVERY rank instead of training on garbage.  Every rank calls it (three collectives)."""
        import torch.distributed as dist
        w, r = self.world, self.rank
        keep = self.h_all, self.dz, self.dz_mine, self.n, self.chunk
        try:
            self.n, self.chunk = w * 32, 32
            self.h_all = torch.zeros(w * 32, KGW_C, device=device)
            self.h_all[r * 32:(r + 1) * 32] = float(r + 1)
            self.dz = torch.arange(w * 32, device=device, dtype=torch.float32).view(-1, 1).expand(w * 32, KGW_C).contiguous() * (r + 1)
            self.dz_mine = torch.empty(32, KGW_C, device=device)
            self.gather()
            self.scatter()
            want_h = torch.arange(1, w + 1, device=device, dtype=torch.float32).repeat_interleave(32).view(-1, 1).expand(w * 32, KGW_C)
            want_d = torch.arange(r * 32, (r + 1) * 32, device=device, dtype=torch.float32).view(-1, 1).expand(32, KGW_C) * (w * (w + 1) / 2)
            ok = bool(torch.equal(self.h_all, want_h)) and bool(torch.allclose(self.dz_mine, want_d))
            if self.backend == 'nccl' and dist.get_backend(self.group) == 'fake':
                ok = True                                   # (bench.py --as-rank: the collectives move nothing by design)
        finally:
            self.h_all, self.dz, self.dz_mine, self.n, self.chunk = keep
            for v in self.bytes.values():
                v[0] = v[1] = 0
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t[0]))

    def _setup(self, X):
        N = X.shape[0]
        if self.h_all is not None and self.n == N:
            return
        self.n = N
        self.chunk = -(-N // (self.world * 32)) * 32                   # rows per rank, a multiple of 32
        dev = X.device
        self.h_all = torch.zeros(self.world * self.chunk, KGW_C, device=dev)
        self.dz = torch.zeros(self.world * self.chunk, KGW_C, device=dev)     # rows >= N stay zero
        self.dz_mine = torch.empty(self.chunk, KGW_C, device=dev)

    def rows(self):
        lo = self.rank * self.chunk
        return lo, max(min(self.n, lo + self.chunk), lo)

    def forward_partial(self, X, W, b):
        self._setup(X)
        self.last = (X, W, b)
        lo, hi = self.rows()
        if hi > lo:
            Xf = _resident_copies(X)[0]
            Kp = Xf.shape[1]
            gemm3(Xf[lo:hi], gemm3_pack(W, Kp, False, k_valid=W.shape[1]), bias=b, relu=True, out=self.h_all[lo:hi])

    def gather(self):
        import torch.distributed as dist
        lo = self.rank * self.chunk
        mine = self.h_all[lo:lo + self.chunk]
        if self.backend == 'nccl':
            dist.all_gather_into_tensor(self.h_all, mine, group=self.group)      # in place: the rank's rows already lie in their slot
        else:
            dist.all_gather(list(self.h_all.view(self.world, self.chunk, KGW_C).unbind(0)), mine.clone(), group=self.group)
        c = self.bytes['all_gather(first gene layer output)']
        c[0] += 1; c[1] += self.h_all.numel() * 4

    def scatter(self):
        """dz (this rank's, dense over the resident rows) -> dz_mine = sum over ranks of the rows this rank owns."""
        import torch.distributed as dist
        lo = self.rank * self.chunk
        if self.backend == 'nccl':
            dist.reduce_scatter_tensor(self.dz_mine, self.dz, op=dist.ReduceOp.SUM, group=self.group)
        else:                                                     # gloo (tests): no reduce-scatter on device tensors
            t = self.dz.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self.dz_mine.copy_(t[lo:lo + self.chunk])
        c = self.bytes['reduce_scatter(first gene layer dz)']
        c[0] += 1; c[1] += self.dz.numel() * 4

    def weight_grad_partial(self, X, out=None):
        """[128, K] partial of the weight gradient from the rows this rank owns (zeros if it owns none)."""
        lo, hi = self.rows()
        Xt = _resident_copies(X)[1]                                # [K, Np]
        kin = min(self.chunk, Xt.shape[1] - lo)
        if out is None:
            out = torch.empty(KGW_C, X.shape[1], device=X.device)
        if hi <= lo or kin <= 0:
            return out.zero_()
        return gemm3(Xt[:, lo:lo + kin], gemm3_pack(self.dz_mine[:hi - lo], kin, True, k_valid=hi - lo), transpose_out=True, out=out)


def gene_layer_split_pays(world: int, width: int) -> bool:
    """Default of the gene-layer split over the ranks.  Its output does not depend on the batch, so EVERY multi-rank mode -- per-GPU
    batches (weak) as much as one batch split over the GPUs (strong) -- repeats the same 2 x 26 GFLOP (width 5 120; x 11 at 57 742)
    on every rank without it.  It costs an all-gather and a reduce-scatter of 10 MB and three more graph boundaries (~0.17 ms
    together, measured / modelled); it saves (world - 1) / world of 0.275 ms x width / 5 120.  Measured with bench.py --as-rank
    (profiles/r4): width 5 120 -- rank compute 1.231 -> 1.158 / 1.097 / 1.069 ms at 2 / 4 / 8 ranks; width 57 742 -- 4.185 ->
    2.855 / 1.661 ms at 2 / 8 ranks."""
    if width <= 0:                     # (feature width unknown -- a data object without gene_init_dim_size: the measured 5 120-wide rule)
        return world >= 4
    # break-even 0.19 ms: on from 4 ranks at width 5 120 (0.206 saved; 3 ranks: 0.183, measured a wash), from 2 at 57 742
    return world >= 2 and (world - 1) / world * 0.275 * (width / 5120.0) > 0.19


GENE_SHARD = None          # the GeneLayerShard of the training step being issued (gene_shard_scope), or None
RESIDENT_SEEN = None       # a list while a trainer probes which layer takes the resident route (gene_shard_scope(None, probe))


class gene_shard_scope:
    """``with gene_shard_scope(gs):`` -- the forward passes issued inside take ``gs`` for the resident first gene Linear; outside
    (another model, an evaluation pass, KGWAS.train_step after a trainer has finished) no shard is active: the layer is computed
    locally and nothing stale is read.  ``probe``: a list that receives (X, W, b) of every training forward that takes the
    resident route -- how a trainer learns, WITHOUT issuing a collective, whether (and on which layer) the route is taken, so that
    the ranks can agree on it before the first collective depends on it."""

    def __init__(self, gs, probe=None):
        self.gs, self.probe = gs, probe

    def __enter__(self):
        global GENE_SHARD, RESIDENT_SEEN
        self.prev = (GENE_SHARD, RESIDENT_SEEN)
        GENE_SHARD, RESIDENT_SEEN = self.gs, self.probe
        return self.gs

    def __exit__(self, *exc):
        global GENE_SHARD, RESIDENT_SEEN
        GENE_SHARD, RESIDENT_SEEN = self.prev
        return False


def resident_first_linear(X, W, b, g2l=None, rows_out=None, gs=None, rows_real=None):
    """relu(X W^T + b) over ALL rows of a resident wide feature matrix (kgwas/model.py:19 on the gene features).  ``g2l`` +
    ``rows_out``: the batch's rows (rows_out[g2l[r]] = row r where g2l[r] >= 0) are written by the same launch; returns
    (h, True) then, (h, False) when the caller still has to gather them."""
    if _resident_ok(X, W):
        if gs is not None:
            if gs.inline:
                gs.forward_partial(X, W, b)
                gs.gather()
            return gs.h_all[:X.shape[0]], False            # (staged: the trainer ran forward_partial + gather for this step)
        Xf = _resident_copies(X)[0]                      # built outside any graph capture, on the first eager step
        Kp = Xf.shape[1]
        fused = g2l is not None and rows_out is not None and rows_out.numel() > 0
        # (a width that is not a multiple of 32: the packing kernel reads the weight's real columns and pads with zeros)
        h = gemm3(Xf, gemm3_pack(W, Kp, False, k_valid=W.shape[1]), bias=b, relu=True, row_map=g2l if fused else None,
                  out_rows=rows_out if fused else None, out_rows_real=rows_real if fused else None)
        return h, fused
    return linear(X, W, b, relu=True, fixed_shape=True), False


def active_gene_shard(ctx, w_index: int, X=None, W=None, b=None):
    """The GeneLayerShard of the step being issued while a TRAINING forward runs (the autograd node is asked for the layer's weight
    gradient); inference passes -- captured forward graphs, loaders of different lengths per rank -- always compute the whole layer
    locally.  A shard serves ONE layer: it is bound to the (X, W) it is first applied to, and a second resident layer of the same
    model (wide GO features) computes locally instead of clobbering its buffers."""
    if not ctx.needs_input_grad[w_index]:
        return None
    if RESIDENT_SEEN is not None and X is not None:
        RESIDENT_SEEN.append((X, W, b))
    gs = GENE_SHARD
    if gs is None:
        return None
    if gs.last is not None and X is not None and not (gs.last[0] is X and gs.last[1] is W):
        return None
    return gs


def resident_first_weight_grad(dz, X, W, gs=None):
    """dW [128, K] = dz^T X for the same layer (``gs``, the GeneLayerShard the forward used: this rank's PARTIAL of it, or None
    when the trainer computes the partial itself after its reduce-scatter stage)."""
    if _resident_ok(X, W) and dz.is_contiguous():
        if gs is not None:
            assert dz.data_ptr() == gs.dz.data_ptr(), 'the sharded first layer differentiates into GeneLayerShard.dz'
            if not gs.inline:
                return None
            gs.scatter()
            return gs.weight_grad_partial(X)
        Xt = _resident_copies(X)[1]                      # [K, Np], Np = the node count rounded up to 32, zero columns past it
        return gemm3(Xt, gemm3_pack(dz, Xt.shape[1], True, k_valid=X.shape[0]), transpose_out=True, defer=True)
    if LIBRARY_GEMM.own_first and 0 < dz.shape[0] and X.dtype == torch.float32:
        return tn_gemm(dz, X)
    LIBRARY_GEMM.note('resident_first_weight_grad', dz.shape[0], dz.shape[1], X.shape[1])
    with _TUNED:
        return dz.t().mm(X)


def _dz_buffer(h, gs):
    """Where the dense d(pre-activation) of the resident first layer is written: GeneLayerShard.dz (its first N rows) when the
    layer is sharded over ranks, else a fresh [N, 128]."""
    if gs is not None:
        assert gs.dz is not None and gs.n == h.shape[0]
        return gs.dz[:h.shape[0]]
    return torch.empty_like(h)


class _MLPTail(torch.autograd.Function):
    """y = FC_output(relu(FC_hidden2(h1)))  (kgwas/model.py:19-21) as ONE autograd node: MFMA Linear kernels
    forward and for dX (ReLU mask fused in the epilogue), split-K MFMA kernel for the weight / bias gradients."""

    @staticmethod
    def forward(ctx, h1, W2, b2, W3, b3, out=None):
        h2 = linear(h1, W2, b2, relu=True)
        y = linear(h2, W3, b3, out=out.view() if out is not None else None)
        ctx.save_for_backward(h1, h2, W2, W3)
        return y

    @staticmethod
    def backward(ctx, dy):
        h1, h2, W2, W3 = ctx.saved_tensors
        dy = dy.contiguous()
        dh2 = linear(dy, W3, mask=h2, w_kn=True)                 # (dy @ W3) * (h2 > 0)
        dh1 = linear(dh2, W2, w_kn=True) if ctx.needs_input_grad[0] else None
        (dW3, db3), (dW2, db2) = weight_grads([(dy, h2), (dh2, h1)])      # both products in one launch pair
        return dh1, dW2, db2, dW3, db3, None


class RowBlock:
    """Rows [lo, lo+n) of a preallocated activation matrix: a node that is handed one writes its output there (the
    next layer's type-major input is then assembled without a concatenation copy)."""

    def __init__(self, buf: torch.Tensor, lo: int, n: int):
        self.buf, self.lo, self.n = buf, int(lo), int(n)

    def view(self):
        # an ALIAS of the rows, not an autograd view of ``buf``: blocks are written independently of each other and
        # view/in-place version tracking across them would (rightly, in general) refuse that
        b = self.buf
        t = torch.empty(0, dtype=b.dtype, device=b.device)
        return t.set_(b.untyped_storage(), b.storage_offset() + self.lo * b.stride(0), (self.n,) + tuple(b.shape[1:]),
                      b.stride())


class _JoinBlocks(torch.autograd.Function):
    """The type-major layer input from its per-type blocks; blocks already written in place (RowBlock outputs) cost
    nothing, others are copied.  Backward hands each block its rows of dH as a view."""

    @staticmethod
    def forward(ctx, holder, spans, *parts):
        buf = holder.buf
        for (lo, n), p in zip(spans, parts):
            dst = buf[lo:lo + n]
            if p.data_ptr() != dst.data_ptr() or p.stride(0) != buf.stride(0):
                dst.copy_(p)
        ctx.spans = spans
        return holder.view()

    @staticmethod
    def backward(ctx, dH):
        return (None, None) + tuple(dH[lo:lo + n] for lo, n in ctx.spans)


class _SplitRows(torch.autograd.Function):
    """torch.split along rows whose backward does not concatenate when the incoming gradients already are adjacent
    row blocks of one tensor (the slices _JoinBlocks hands back): it returns that tensor's rows as they lie."""

    @staticmethod
    def forward(ctx, y, sizes):
        ctx.sizes = sizes
        ctx.set_materialize_grads(False)
        outs, off = [], 0
        for n in sizes:
            outs.append(RowBlock(y, off, n).view())
            off += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        sizes = ctx.sizes
        if all(g is not None for g in gs):
            g0 = gs[0]
            adj, ptr = g0.is_contiguous(), g0.data_ptr()
            for g, n in zip(gs, sizes):
                adj = adj and g.is_contiguous() and g.data_ptr() == ptr and g.untyped_storage().data_ptr() == g0.untyped_storage().data_ptr()
                ptr += n * g.shape[1] * g.element_size()
            if adj:
                total = sum(sizes)
                t = torch.empty(0, dtype=g0.dtype, device=g0.device)
                return t.set_(g0.untyped_storage(), g0.storage_offset(), (total, g0.shape[1]), (g0.shape[1], 1)), None
        first = next(g for g in gs if g is not None)
        parts = [g if g is not None else torch.zeros(n, first.shape[1], device=first.device) for g, n in zip(gs, sizes)]
        return torch.cat(parts, 0), None


def split_rows(y: torch.Tensor, sizes):
    return _SplitRows.apply(y, tuple(int(n) for n in sizes))


def join_blocks(buf: torch.Tensor, spans, parts):
    return _JoinBlocks.apply(RowBlock(buf, 0, buf.shape[0]), tuple(spans), *parts)


class _MLP3(torch.autograd.Function):
    """SimpleMLP (kgwas/model.py:17-21) on a feature matrix that needs no gradient, as ONE autograd node: three
    MFMA Linear launches forward; backward = two dX launches with the ReLU masks fused in their epilogues and three
    split-K weight-gradient launches (no stand-alone ReLU-backward / bias-sum kernels)."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, W3, b3, out, rows_dev):
        h1 = linear(x, W1, b1, relu=True, rows_dev=rows_dev)
        h2 = linear(h1, W2, b2, relu=True, rows_dev=rows_dev)
        y = linear(h2, W3, b3, out=out.view() if out is not None else None, rows_dev=rows_dev)
        ctx.save_for_backward(x, h1, h2, W2, W3)
        ctx.rows_dev = rows_dev
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h1, h2, W2, W3 = ctx.saved_tensors
        rd = ctx.rows_dev
        dy = dy.contiguous()
        dh2 = linear(dy, W3, mask=h2, w_kn=True, rows_dev=rd)    # (dy @ W3) * (h2 > 0)
        dh1 = linear(dh2, W2, mask=h1, w_kn=True, rows_dev=rd)   # (dh2 @ W2) * (h1 > 0)
        (dW3, db3), (dW2, db2), (dW1, db1) = weight_grads([(dy, h2), (dh2, h1), (dh1, x)], rows_dev=rd)
        return None, dW1, db1, dW2, db2, dW3, db3, None, None


class _MLP2(torch.autograd.Function):
    """h2 = relu(FC_hidden2(relu(FC_hidden(x)))) -- SimpleMLP without its last Linear (kgwas/model.py:18-20), for a
    feature matrix that needs no gradient.  FC_output is folded into the layer-1 relation parameters (fold_fc_output_hip),
    and the consumer of h2 (gat_aggregate(relu_input=True)) hands back a gradient ALREADY multiplied by (h2 > 0)."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, out, rows_dev, ids=None):
        # ``ids``: x is a RESIDENT feature matrix and the input rows are x[ids] (the loader's slicing, kgwas/kgwas.py:135)
        rows, K1 = (x.shape if ids is None else (ids.numel(), x.shape[1]))
        h2 = out.view() if out is not None else None
        if (rows >= 16384 and K1 <= 20 and K1 % 4 == 0 and W1.shape[0] == KGW_C and W2.shape == (KGW_C, KGW_C)
                and x.dtype == torch.float32 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
                and W1.stride(1) == 1 and W2.stride(1) == 1 and W2.stride(0) % 4 == 0 and W2.data_ptr() % 16 == 0
                and b1 is not None and b2 is not None and (ids is None or ids.dtype == torch.int32)
                and (h2 is None or (h2.stride(1) == 1 and h2.stride(0) % 4 == 0 and h2.data_ptr() % 16 == 0))):
            # narrow first layer on many rows (the 20-wide SNP features): both layers in one launch, the hidden state handed
            # from the first product's accumulators to the second product's operand registers (kgw_mlp2_fwd); the row gather
            # rides along
            if h2 is None:
                h2 = torch.empty(rows, KGW_C, device=x.device)
            h1 = torch.empty(rows, KGW_C, device=x.device)
            xg = torch.empty(rows, K1, device=x.device) if ids is not None else None
            _lib.check(_lib.lib().kgw_mlp2_fwd(_p(x), x.stride(0), K1, _p(W1), W1.stride(0), _p(b1), _p(W2), W2.stride(0), _p(b2),
                                               _p(h1), h1.stride(0), _p(h2), h2.stride(0), rows, _p(rows_dev), _p(ids), _p(xg),
                                               K1, _lib.stream_ptr()), 'kgw_mlp2_fwd')
            ctx.save_for_backward(x if ids is None else xg, h1, W2)
            ctx.rows_dev = rows_dev
            return h2
        if ids is not None:
            from .sampler import gather_rows
            x = gather_rows(x, ids)
        h1 = linear(x, W1, b1, relu=True, rows_dev=rows_dev)
        h2 = linear(h1, W2, b2, relu=True, out=h2, rows_dev=rows_dev)
        ctx.save_for_backward(x, h1, W2)
        ctx.rows_dev = rows_dev
        return h2

    @staticmethod
    def backward(ctx, dh2):
        x, h1, W2 = ctx.saved_tensors
        rd = ctx.rows_dev
        dh2 = dh2.contiguous()
        rows, K1 = x.shape
        if (rows >= 16384 and K1 <= 31 and h1.shape[1] == KGW_C and W2.shape == (KGW_C, KGW_C) and x.stride(1) == 1
                and h1.stride(1) == 1 and dh2.stride(0) % 4 == 0 and h1.stride(0) % 4 == 0 and W2.stride(0) % 4 == 0
                and dh2.data_ptr() % 16 == 0 and h1.data_ptr() % 16 == 0 and W2.data_ptr() % 16 == 0):
            # narrow first layer, no input gradient: dh1 = (dh2 @ W2) * (h1 > 0) is consumed tile by tile by the d W1 / d b1
            # product inside ONE kernel (kgw_mlp2_bwd_first) -- neither written nor re-read
            L = _lib.lib()
            dW1 = torch.empty(KGW_C, K1, device=x.device)
            db1 = torch.empty(KGW_C, device=x.device)
            nws = int(L.kgw_mlp2_bwd_first_workspace_floats(rows))
            ws = torch.empty(nws, device=x.device)
            sink = GRAD_SINK
            if sink is not None:        # the blocks' partial d W1 / d b1 are added inside the optimiser's launch (GradSink)
                src = (_lib.KgwGradSrc * 2)()
                _lib.check(L.kgw_mlp2_bwd_first_partial(_p(dh2), dh2.stride(0), _p(W2), W2.stride(0), _p(h1), h1.stride(0), _p(x), x.stride(0),
                                                        K1, rows, _p(rd), _p(dW1), K1, _p(db1), _p(ws), nws, None, None, 0, src,
                                                        _lib.stream_ptr()), 'kgw_mlp2_bwd_first_partial')
                sink.add(dW1, src[0], ws)
                sink.add(db1, src[1], ws)
            else:
                _lib.check(L.kgw_mlp2_bwd_first(_p(dh2), dh2.stride(0), _p(W2), W2.stride(0), _p(h1), h1.stride(0), _p(x), x.stride(0), K1,
                                                rows, _p(rd), _p(dW1), K1, _p(db1), _p(ws), nws, None, None, 0, _lib.stream_ptr()), 'kgw_mlp2_bwd_first')
            dW2, db2 = linear_weight_grad(dh2, h1, rows_dev=rd)
            return None, dW1, db1, dW2, db2, None, None, None
        dh1 = linear(dh2, W2, mask=h1, w_kn=True, rows_dev=rd)   # (dh2 @ W2) * (h1 > 0)
        (dW2, db2), (dW1, db1) = weight_grads([(dh2, h1), (dh1, x)], rows_dev=rd)
        return None, dW1, db1, dW2, db2, None, None, None


def mlp2(x, W1, b1, W2, b2, out=None, rows_dev=None, ids=None):
    """``ids`` (int32): the input is ``x[ids]`` for a resident feature matrix ``x`` (gathered inside the fused kernel)."""
    return _MLP2.apply(x, W1, b1, W2, b2, out, rows_dev, ids)


class _MLP2Gathered(torch.autograd.Function):
    """_MLP2 for a 128-wide input whose rows are gathered from several resident feature matrices (the three GO node types
    share go_feat_mlp, kgwas/model.py:58-60): gather + FC_hidden + FC_hidden2 in ONE launch (kgw_mlp2w_fwd); backward as
    _MLP2's (the kernel leaves the gathered rows and h1 for it)."""

    @staticmethod
    def forward(ctx, W1, b1, W2, b2, out, *jobs):
        srcs, idss = jobs[0::2], jobs[1::2]
        n = len(srcs)
        rows = sum(int(i.numel()) for i in idss)
        dev = W1.device
        h2 = out.view() if out is not None else torch.empty(rows, KGW_C, device=dev)
        xg = torch.empty(rows, KGW_C, device=dev)
        h1 = torch.empty(rows, KGW_C, device=dev)
        S = (C.c_void_p * n)(*[_p(x) for x in srcs])
        I = (C.c_void_p * n)(*[_p(i) for i in idss])
        N = (C.c_int64 * n)(*[int(i.numel()) for i in idss])
        assert h2.stride(0) == KGW_C or rows == 0 or h2.stride(0) % 4 == 0
        if h2.stride(0) != KGW_C:                       # (outputs share one row stride in the kernel)
            h2c = torch.empty(rows, KGW_C, device=dev)
        else:
            h2c = h2
        _lib.check(_lib.lib().kgw_mlp2w_fwd(n, S, I, N, srcs[0].stride(0), _p(W1), W1.stride(0), _p(b1), _p(W2), W2.stride(0), _p(b2),
                                            _p(xg), _p(h1), _p(h2c), KGW_C, _lib.stream_ptr()), 'kgw_mlp2w_fwd')
        if h2c is not h2:
            h2.copy_(h2c)
        ctx.save_for_backward(xg, h1, W2)
        return h2

    @staticmethod
    def backward(ctx, dh2):
        x, h1, W2 = ctx.saved_tensors
        dh2 = dh2.contiguous()
        dh1 = linear(dh2, W2, mask=h1, w_kn=True)       # (dh2 @ W2) * (h1 > 0)
        (dW2, db2), (dW1, db1) = weight_grads([(dh2, h1), (dh1, x)])
        return (dW1, db1, dW2, db2, None) + (None,) * (len(ctx.needs_input_grad) - 5)


def mlp2_gathered(jobs, W1, b1, W2, b2, out=None):
    """``jobs``: [(resident feature matrix [N_t, 128], int32 ids)]: h2 = relu(FC_hidden2(relu(FC_hidden(cat_t X_t[ids_t]))))."""
    flat = []
    for x, i in jobs:
        flat += [x, i]
    return _MLP2Gathered.apply(W1, b1, W2, b2, out, *flat)


def mlp2_gathered_ok(jobs, W1, W2) -> bool:
    return (1 <= len(jobs) <= 4 and W1.shape == (KGW_C, KGW_C) and W2.shape == (KGW_C, KGW_C)
            and all(x.dtype == torch.float32 and x.shape[1] == KGW_C and x.stride(1) == 1 and x.stride(0) % 4 == 0 and
                    x.data_ptr() % 16 == 0 and i.dtype == torch.int32 and x.stride(0) == jobs[0][0].stride(0) for x, i in jobs)
            and 0 < sum(int(i.numel()) for _, i in jobs) <= 16384)


class _MLPTail2(torch.autograd.Function):
    """h2 = relu(FC_hidden2(h1)) (the folded counterpart of _MLPTail); incoming gradient premasked like _MLP2's."""

    @staticmethod
    def forward(ctx, h1, W2, b2, out=None):
        h2 = linear(h1, W2, b2, relu=True, out=out.view() if out is not None else None)
        ctx.save_for_backward(h1, W2)
        return h2

    @staticmethod
    def backward(ctx, dh2):
        h1, W2 = ctx.saved_tensors
        dh2 = dh2.contiguous()
        dh1 = linear(dh2, W2, w_kn=True) if ctx.needs_input_grad[0] else None
        dW2, db2 = linear_weight_grad(dh2, h1)
        return dh1, dW2, db2, None


def mlp_tail2(h1, W2, b2, out=None):
    return _MLPTail2.apply(h1, W2, b2, out)


def mlp3(x, W1, b1, W2, b2, W3, b3, out=None, rows_dev=None):
    return _MLP3.apply(x, W1, b1, W2, b2, W3, b3, out, rows_dev)


class _LinearReLU(torch.autograd.Function):
    """h = relu(x W^T + b) with the weight gradient on the split-K kernel; x gets a gradient only if asked."""

    @staticmethod
    def forward(ctx, x, W, b, fixed_shape=False):
        h = linear(x, W, b, relu=True, fixed_shape=fixed_shape)
        ctx.save_for_backward(x, h, W)
        ctx.fixed_shape = fixed_shape
        return h

    @staticmethod
    def backward(ctx, dh):
        x, h, W = ctx.saved_tensors
        dz = torch.ops.aten.threshold_backward(dh.contiguous(), h, 0.0)      # dh * (h > 0), one launch
        dW, db = linear_weight_grad(dz, x, ctx.fixed_shape)
        dx = linear(dz, W, w_kn=True) if ctx.needs_input_grad[0] else None
        return dx, dW, db, None


class _ResidentLinearReLURows(torch.autograd.Function):
    """rows ``ids`` of relu(X W^T + b) where X is a RESIDENT feature matrix (the 5120 / 57742-wide gene features): the
    product runs on all of X (kgw_gemm3; small or unaligned shapes: the tuned library GEMM) and the batch takes its rows -- no x[n_id]
    copy of 20 KB rows (kgwas/kgwas.py:135).  ``g2l`` [N] int32: local index of each row of X in the batch, -1 = not
    sampled (the inverse of ``ids``, kept by the sampler).  Backward: kgw_scatter_relu_rows builds the dense dz (zero
    rows for unsampled nodes, ReLU mask applied) and the bias gradient in two launches; dW = dz^T X on kgw_gemm3 (or the library)."""

    @staticmethod
    def forward(ctx, X, W, b, ids, g2l):
        n = int(ids.numel())
        # zeros: the rows past the batch's real node count (static capacity of a captured step) are written by nobody on the
        # fused route, and 0 x garbage must stay 0 in the weight gradients downstream
        out = torch.zeros(n, KGW_C, device=X.device) if _resident_ok(X, W) else torch.empty(n, KGW_C, device=X.device)
        ctx.shard = active_gene_shard(ctx, 1, X, W, b) if _resident_ok(X, W) else None
        h, done = resident_first_linear(X, W, b, g2l, out, ctx.shard)
        if n and not done:
            _lib.check(_lib.lib().kgw_gather_rows(_p(h), _p(ids), n, h.shape[1], _p(out), _lib.stream_ptr()), 'kgw_gather_rows')
        ctx.save_for_backward(X, h, g2l, W)
        return out

    @staticmethod
    def backward(ctx, g):
        X, h, g2l, W = ctx.saved_tensors
        g = g.contiguous()
        N = h.shape[0]
        assert h.shape[1] == KGW_C and g2l.numel() == N and g2l.dtype == torch.int32
        dz = _dz_buffer(h, ctx.shard)
        db = torch.empty(KGW_C, device=h.device)
        ws = torch.empty(int(_lib.lib().kgw_scatter_relu_rows_workspace_floats(N)), device=h.device)
        _lib.check(_lib.lib().kgw_scatter_relu_rows(_p(g), _p(g2l), _p(h), N, _p(dz), _p(db), _p(ws), _lib.stream_ptr()),
                   'kgw_scatter_relu_rows')
        dW = resident_first_weight_grad(dz, X, W, ctx.shard)
        return None, dW, db, None, None


class _ResidentMLP2(torch.autograd.Function):
    """h2 = relu(FC_hidden2(relu(FC_hidden(X))[ids])) for a RESIDENT wide feature matrix X (the gene features): the two nodes
    _ResidentLinearReLURows + _MLPTail2 as one, so that the backward forms  dz = ((dh2[g2l] @ W2) * (h > 0))  -- dense over
    the resident rows, zero where a node is not in the batch -- and d b1 in ONE kernel (kgw_mlp2_bwd_first with row
    indirection) instead of dX product + scatter/mask pass + column-sum fold; d W1 = dz^T X on kgw_gemm3 (or the library).
    Contract (like _MLP2): the incoming gradient is ALREADY multiplied by (h2 > 0) -- the node sits below the layer-1 aggregate,
    whose backward applies the ReLU mask of its input (KGW_F_RELU_INPUT); tests/test_gpu_gemm3.py checks it against float64."""

    @staticmethod
    def forward(ctx, X, W1, b1, W2, b2, ids, g2l, out, rows_real=None):
        n = int(ids.numel())
        ctx.shard = active_gene_shard(ctx, 1, X, W1, b1) if _resident_ok(X, W1) else None
        # the rows past the batch's real node count (static capacity of a captured step) are written by nobody on the fused
        # route and must be zero (see _ResidentLinearReLURows): with ``rows_real`` (their count on the device) the product's
        # range-sum launch zeroes exactly those rows itself -- no fill launch over the whole block
        in_kernel = rows_real is not None and n > 0 and ctx.shard is None and _resident_ok(X, W1)
        h1g = (torch.zeros(n, KGW_C, device=X.device) if (_resident_ok(X, W1) and not in_kernel)
               else torch.empty(n, KGW_C, device=X.device))
        h, done = resident_first_linear(X, W1, b1, g2l, h1g, ctx.shard, rows_real if in_kernel else None)
        if n and not done:
            _lib.check(_lib.lib().kgw_gather_rows(_p(h), _p(ids), n, h.shape[1], _p(h1g), _lib.stream_ptr()), 'kgw_gather_rows')
        h2 = linear(h1g, W2, b2, relu=True, out=out.view() if out is not None else None)
        ctx.save_for_backward(X, h, h1g, W2, g2l, W1)
        return h2

    @staticmethod
    def backward(ctx, dh2):
        X, h, h1g, W2, g2l, W1 = ctx.saved_tensors
        dh2 = dh2.contiguous()
        N = h.shape[0]
        L = _lib.lib()
        db1 = torch.empty(KGW_C, device=h.device)
        nws = int(L.kgw_mlp2_bwd_first_workspace_floats(N))
        ws = torch.empty(nws, device=h.device)
        sink = GRAD_SINK
        if _PACK_FUSED and ctx.shard is None and _resident_ok(X, W1):
            # the masked dh1 rows leave the kernel as kgw_gemm3's B operand image (the weight gradient's): no fp32 rows, no
            # kgw_gemm3_pack launch
            Xt = _resident_copies(X)[1]                  # [K, Np], Np = the node count rounded up to 32
            packed = torch.empty(int(L.kgw_gemm3_packed_bytes(Xt.shape[1])), dtype=torch.uint8, device=h.device)
            src = (_lib.KgwGradSrc * 2)()
            rc = L.kgw_mlp2_bwd_first_packed(_p(dh2), dh2.stride(0), _p(W2), W2.stride(0), _p(h), h.stride(0), None, 0, 0, N, None, 0,
                                             _p(db1), _p(ws), nws, _p(g2l), None, 0, _p(packed), int(L.kgw_gemm3_flip()),
                                             src if sink is not None else None, _lib.stream_ptr())
            if rc != _lib.KGW_E_UNSUPPORTED:
                _lib.check(rc, 'kgw_mlp2_bwd_first_packed')
                if sink is not None:
                    sink.add(db1, src[1], ws)
                dW1 = gemm3(Xt, packed, transpose_out=True, defer=True)
                dW2, db2 = linear_weight_grad(dh2, h1g)
                return None, dW1, db1, dW2, db2, None, None, None, None
        dz = _dz_buffer(h, ctx.shard)
        if sink is not None:
            src = (_lib.KgwGradSrc * 2)()
            _lib.check(L.kgw_mlp2_bwd_first_partial(_p(dh2), dh2.stride(0), _p(W2), W2.stride(0), _p(h), h.stride(0), None, 0, 0, N, None,
                                                    None, 0, _p(db1), _p(ws), nws, _p(g2l), _p(dz), dz.stride(0), src, _lib.stream_ptr()),
                       'kgw_mlp2_bwd_first_partial')
            sink.add(db1, src[1], ws)
        else:
            _lib.check(L.kgw_mlp2_bwd_first(_p(dh2), dh2.stride(0), _p(W2), W2.stride(0), _p(h), h.stride(0), None, 0, 0, N, None, None, 0,
                                            _p(db1), _p(ws), nws, _p(g2l), _p(dz), dz.stride(0), _lib.stream_ptr()), 'kgw_mlp2_bwd_first')
        dW1 = resident_first_weight_grad(dz, X, W1, ctx.shard)
        dW2, db2 = linear_weight_grad(dh2, h1g)
        return None, dW1, db1, dW2, db2, None, None, None, None


def resident_mlp2(X, W1, b1, W2, b2, ids, g2l, out=None, rows_real=None):
    return _ResidentMLP2.apply(X, W1, b1, W2, b2, ids, g2l, out, rows_real)


def resident_mlp2_ok(X, W1, W2, n_local: int) -> bool:
    return (W1.shape[0] == KGW_C and W2.shape == (KGW_C, KGW_C) and n_local > 0 and X.shape[0] >= 4096
            and W2.stride(1) == 1 and W2.stride(0) % 4 == 0 and W2.data_ptr() % 16 == 0)


def resident_linear_relu_rows(X, W, b, ids, g2l):
    return _ResidentLinearReLURows.apply(X, W, b, ids, g2l)


class _LinearAct(torch.autograd.Function):
    """y = [relu](x W^T + b) for W given TRANSPOSED as Wt [K,N] (the packed per-relation weights) -- the
    transform GEMM of a layer: per-relation lin_src + bias + relation sum (+ ReLU) in one launch."""

    @staticmethod
    def forward(ctx, x, Wt, b, relu):
        y = linear(x, Wt, b, relu=relu, w_kn=True)
        ctx.save_for_backward(x, y, Wt)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, Wt = ctx.saved_tensors
        dz = torch.ops.aten.threshold_backward(dy.contiguous(), y, 0.0) if ctx.relu else dy.contiguous()
        if x.shape[0] >= _TN_MIN_ROWS or (LIBRARY_GEMM.own_first and x.shape[0] > 0):
            dWt = tn_gemm(x, dz)                                                       # [K,N]
        else:
            LIBRARY_GEMM.note('linear_act.backward', x.shape[0], x.shape[1], dz.shape[1])
            dWt = x.t().mm(dz)
        db = dz.sum(0)
        dx = linear(dz, Wt) if ctx.needs_input_grad[0] else None                       # dz @ Wt^T: Wt is [K,N] = "[N',K']" form
        return dx, dWt, db, None


def linear_act(x, Wt, b, relu=True):
    return _LinearAct.apply(x, Wt, b, relu)


def linear_weight_grad(dY: torch.Tensor, X: torch.Tensor, fixed_shape: bool = False, rows_dev: torch.Tensor = None):
    """(dW [out,in], db [out]) of Y = X W^T + b given dY [rows,out], X [rows,in]."""
    rows, K = X.shape
    if (rows >= _TN_MIN_ROWS and K <= 1024) or (LIBRARY_GEMM.own_first and rows > 0 and X.dtype == torch.float32):
        if GRAD_SINK is not None:                   # (callers: parameter gradients only)
            dYc = dY if dY.stride(1) == 1 else dY.contiguous()
            Xc = X if X.stride(1) == 1 else X.contiguous()
            r = GRAD_SINK.defer_product(dYc, Xc, rows_dev)
            if r is not None:
                return r
        return tn_gemm(dY, X, colsum=True, rows_dev=rows_dev, defer=True)
    LIBRARY_GEMM.note('linear_weight_grad', rows, dY.shape[1], K)
    if fixed_shape:
        with _TUNED:
            return dY.t().mm(X), colsum(dY)
    return dY.t().mm(X), colsum(dY)


def colsum(X: torch.Tensor):
    """Column sums (bias gradient of a library-routed Linear).  A ticketed last-block-fold kernel was tried here and
    lost to the framework's reduction (29 vs 17 us at 20 k x 128: the device-scope fence of the hand-off costs more
    than a second launch would)."""
    return X.sum(0)


def mlp_tail(h1, W2, b2, W3, b3, out=None):
    return _MLPTail.apply(h1, W2, b2, W3, b3, out)


def linear_relu(x, W, b, fixed_shape=False):
    """``fixed_shape``: x is a resident matrix (same shape every step) -- lets a library-routed product use the
    tuned solution."""
    return _LinearReLU.apply(x, W, b, fixed_shape)


# ------------------------------------------------------------------------------------------------------
# per-layer parameter plumbing as two autograd nodes (instead of ~60 tiny framework ops per step)
# ------------------------------------------------------------------------------------------------------
class _RelVectors(torch.autograd.Function):
    """(U [NR,C], V [NR,C]), rows by relation id, from the packed relation parameters of a layer -- kgw_relvec_fwd / _bwd.
    ``pass_w``: also returns w_src_t itself (a view).  The layer's transform GEMM / the FC_output fold take THAT tensor
    instead of the parameter, so the gradient they produce for the weights comes back to this node and is added inside
    kgw_relvec_bwd_acc -- not by a framework add launch (the parameter would otherwise have two consumers)."""

    @staticmethod
    def forward(ctx, w_src_t, w_dst_t, att_src, att_dst, pack, blk_of_live, n_blk, zero=None, pass_w=False):
        n, C = att_src.shape
        if zero is not None:
            assert zero.dtype == torch.float32 and zero.is_contiguous() and zero.numel() % 4 == 0
        dev = att_src.device
        U = torch.empty(pack.n_rels_total, C, device=dev)
        V = torch.empty(pack.n_rels_total, C, device=dev)        # by relation id, like U
        bsum = torch.empty(max(n_blk, 1), C, device=dev) if blk_of_live is not None else None
        _lib.check(_lib.lib().kgw_relvec_fwd(pack.n_rels_total, _p(pack.live_of_rel_i32), _p(pack.bip_pos_i32), _p(w_src_t),
                                             _p(w_dst_t) if w_dst_t.numel() else 0, _p(att_src), _p(att_dst), _p(U), _p(V),
                                             1, n, _p(pack.bias.detach()) if bsum is not None else 0, _p(blk_of_live),
                                             n_blk if bsum is not None else 0, _p(bsum), _p(zero), zero.numel() if zero is not None else 0,
                                             _lib.stream_ptr()), 'kgw_relvec_fwd')
        ctx.save_for_backward(w_src_t, w_dst_t, att_src, att_dst)
        ctx.pack = pack
        ctx.has_bsum = bsum is not None
        ctx.set_materialize_grads(False)
        outs = [U, V]
        if bsum is not None:
            ctx.mark_non_differentiable(bsum)
            outs.append(bsum)
        if pass_w:
            outs.append(w_src_t.detach())          # (same storage; not a tracked view: no as_strided replay in the backward)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dU, dV, *rest):
        w_src_t, w_dst_t, att_src, att_dst = ctx.saved_tensors
        rest = list(rest)
        if ctx.has_bsum and rest:
            rest.pop(0)
        dW_in = rest[0] if rest else None
        if dU is None and dV is None:
            return (dW_in,) + (None,) * 8
        pack = ctx.pack
        n = att_src.shape[0]
        dU = dU.contiguous() if dU is not None else None
        dV = dV.contiguous() if dV is not None else None
        dW_in = dW_in.contiguous() if dW_in is not None else None
        dws = torch.empty_like(w_src_t)
        dwd = torch.empty_like(w_dst_t)
        das = torch.empty_like(att_src)
        dad = torch.empty_like(att_dst)
        sink = GRAD_SINK
        if sink is not None:
            # (inside a captured step a fold's kgw_fold_bwd or a transform's k_tn_reduce may still be PENDING in the sink -- parked
            #  there for _RelVectorsMulti / k_param_tail -- and this launch reads what they write: dU, dV, dW_in.  The per-layer node
            #  is the one a 1-layer model or KGW_RELVEC_ALL=0 takes: tests/test_gpu_fallbacks.py)
            sink.launch_pending_tail()
            sink.launch_pending_reduce()
        _lib.check(_lib.lib().kgw_relvec_bwd_acc(n, _p(pack.rel_ids_i32), _p(pack.bip_pos_i32), _p(w_src_t),
                                                 _p(w_dst_t) if w_dst_t.numel() else 0, _p(att_src), _p(att_dst), _p(dU), _p(dV),
                                                 _p(dW_in), _p(dws), _p(dwd) if dwd.numel() else 0, _p(das), _p(dad), 1,
                                                 _lib.stream_ptr()), 'kgw_relvec_bwd_acc')
        return dws, dwd, das, dad, None, None, None, None, None


class _RelVectorsMulti(torch.autograd.Function):
    """_RelVectors for ALL layers of the model as one node (round 4): the vectors depend on the parameters only, so the layers'
    kgw_relvec_fwd launches (~30 blocks of latency each) become ONE launch ahead of the first layer, and -- a node's backward
    runs once, when the gradients of all its outputs have arrived -- the layers' kgw_relvec_bwd_acc launches ONE launch after the
    first layer's backward.  Per layer the outputs are (U, V, summed biases, w_src_t as seen through this node); the inputs
    (w_src_t, w_dst_t, att_src, att_dst)."""

    @staticmethod
    def forward(ctx, packs, tabs, n_blks, zeros, *params):
        n = len(packs)
        jobs = (_lib.KgwRelvecJob * n)()
        outs, keep = [], []
        for l, (pack, j) in enumerate(zip(packs, jobs)):
            w_src_t, w_dst_t, att_src, att_dst = params[4 * l:4 * l + 4]
            C = att_src.shape[1]
            dev = att_src.device
            zero = zeros[l]
            if zero is not None:
                assert zero.dtype == torch.float32 and zero.is_contiguous() and zero.numel() % 4 == 0
            U = torch.empty(pack.n_rels_total, C, device=dev)
            V = torch.empty(pack.n_rels_total, C, device=dev)
            bsum = torch.empty(max(n_blks[l], 1), C, device=dev)
            bias = pack.bias.detach()
            keep.append(bias)
            j.n_rels_total, j.n_live, j.n_blk = pack.n_rels_total, att_src.shape[0], n_blks[l]
            j.live_of_rel, j.bip_pos = pack.live_of_rel_i32.data_ptr(), pack.bip_pos_i32.data_ptr()
            j.w_src_t, j.w_dst_t = w_src_t.data_ptr(), (w_dst_t.data_ptr() if w_dst_t.numel() else None)
            j.att_src, j.att_dst, j.U_full, j.V = att_src.data_ptr(), att_dst.data_ptr(), U.data_ptr(), V.data_ptr()
            j.bias, j.blk_of_live, j.bias_sum = bias.data_ptr(), tabs[l].data_ptr(), bsum.data_ptr()
            j.zero_buf, j.zero_floats = (zero.data_ptr(), zero.numel()) if zero is not None else (None, 0)
            outs += [U, V, bsum, w_src_t.detach()]       # (same storage; not a tracked view: no as_strided replay in the backward)
        ctx.mark_non_differentiable(*outs[2::4])         # (the summed biases: layer_transform produces d bias itself)
        q = PARAM_RIDERS
        if q is not None and q.relvec is None:           # (rides on the gene layer's kgw_gemm3 launch: ParamRiders)
            q.relvec = (n, jobs, (keep, outs, params, zeros))
        else:
            _lib.check(_lib.lib().kgw_relvec_fwd_multi(n, jobs, _lib.stream_ptr()), 'kgw_relvec_fwd_multi')
        ctx.save_for_backward(*params)
        ctx.packs = packs
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        params = ctx.saved_tensors
        packs = ctx.packs
        n = len(packs)
        jobs = (_lib.KgwRelvecJob * n)()
        ret, keep, made, live = [], [], [], 0
        for l, pack in enumerate(packs):
            w_src_t, w_dst_t, att_src, att_dst = params[4 * l:4 * l + 4]
            dU, dV, _, dW_in = grads[4 * l:4 * l + 4]
            if dU is None and dV is None:                 # (no gradient reaches the layer's attention: only the transform's share)
                ret += [dW_in, None, None, None]
                continue
            dU = dU.contiguous() if dU is not None else None
            dV = dV.contiguous() if dV is not None else None
            dW_in = dW_in.contiguous() if dW_in is not None else None
            dws, dwd = torch.empty_like(w_src_t), torch.empty_like(w_dst_t)
            das, dad = torch.empty_like(att_src), torch.empty_like(att_dst)
            keep += [dU, dV, dW_in]
            j = jobs[live]
            live += 1
            j.n_live, j.rel_ids, j.bip_pos = att_src.shape[0], pack.rel_ids_i32.data_ptr(), pack.bip_pos_i32.data_ptr()
            j.w_src_t, j.w_dst_t = w_src_t.data_ptr(), (w_dst_t.data_ptr() if w_dst_t.numel() else None)
            j.att_src, j.att_dst = att_src.data_ptr(), att_dst.data_ptr()
            j.dU_full, j.dV, j.dw_src_acc = _p(dU), _p(dV), _p(dW_in)
            pu, pv = _take_pieces(dU), _take_pieces(dV)
            if (pu is None) != (pv is None):
                raise GradSinkMismatch('d u_r and d v_r of a layer must both be pieces or both be complete')
            if pu is not None:                      # (the aggregate left eight pieces per value: added inside k_relvec_bwd)
                j.dU_full, j.dV, j.duv_pieces = pu[0], pv[0], 1
                keep += [pu[1], pv[1]]
            j.dw_src_t, j.dw_dst_t = dws.data_ptr(), (dwd.data_ptr() if dwd.numel() else None)
            j.datt_src, j.datt_dst = das.data_ptr(), dad.data_ptr()
            ret += [dws, dwd, das, dad]
            made += [dws, dwd, das, dad]
        if live:
            sink = GRAD_SINK
            if sink is not None and _PARAM_TAIL and _DEFER_PRODUCTS and sink.relvec_bwd is None:
                # (feeds only the optimiser: launched by GradSink.flush.  The OUTPUT tensors are kept alive through their storages
                #  only: a second reference to a gradient tensor would make autograd's accumulation clone it -- before the deferred
                #  launch has written it -- instead of adopting it)
                outs = [(t.data_ptr(), t.numel(), t.untyped_storage()) for t in made if t.numel()]
                sink.relvec_bwd = (live, jobs, (keep, params), outs)
            else:
                if sink is not None:
                    sink.launch_pending_tail()           # (a fold still pending feeds this launch)
                    sink.launch_pending_reduce()         # (so does the transform's weight gradient)
                _lib.check(_lib.lib().kgw_relvec_bwd_multi(live, jobs, _lib.stream_ptr()), 'kgw_relvec_bwd_multi')
        return (None, None, None, None) + tuple(ret)


def rel_vectors_all(packs, blocks_per_layer, zeros):
    """[(U, V, summed biases, w_src_t through the node)] for every layer in ONE launch (see _RelVectorsMulti); ``zeros``: per
    layer the aggregate workspace to clear in the same launch (or None)."""
    tabs, n_blks, params = [], [], []
    for pack, blocks in zip(packs, blocks_per_layer):
        key = ('blk',) + _block_key(blocks)
        tab = pack._sel_cache.get(key)
        if tab is None:
            blk = [-1] * pack.bias.shape[0]
            for b, (lo, hi) in enumerate(key[1:]):
                for i in range(lo, hi):
                    blk[i] = b
            tab = torch.tensor(blk, dtype=torch.int32, device=pack.bias.device)
            pack._sel_cache[key] = tab
        tabs.append(tab)
        n_blks.append(len(blocks))
        params += [pack.w_src_t, pack.w_dst_t, pack.att_src, pack.att_dst]
    outs = _RelVectorsMulti.apply(tuple(packs), tuple(tabs), tuple(n_blks), tuple(zeros), *params)
    return [tuple(outs[4 * l:4 * l + 4]) for l in range(len(packs))]


def _block_key(blocks):
    return tuple((lo, hi) for lo, hi, *_ in blocks)


def rel_vectors(pack, blocks=None, zero=None, pass_weights=False):
    """(U, V) by relation id; with ``blocks`` (the layer_transform block list, [(lo, hi, ...)]) also the summed bias
    of every block, [n blocks, C] (no gradient flows through it: layer_transform produces d bias itself).
    ``zero``: a float32 buffer to clear in the same launch (``aggregate_workspace``: the zero fill the aggregate needs).
    ``pass_weights``: one more output, pack.w_src_t as seen by autograd THROUGH this node (see _RelVectors)."""
    if blocks is None:
        return _RelVectors.apply(pack.w_src_t, pack.w_dst_t, pack.att_src, pack.att_dst, pack, None, 0, zero, pass_weights)
    key = ('blk',) + _block_key(blocks)
    tab = pack._sel_cache.get(key)
    if tab is None:
        blk = [-1] * pack.bias.shape[0]
        for b, (lo, hi) in enumerate(key[1:]):
            for i in range(lo, hi):
                blk[i] = b
        tab = torch.tensor(blk, dtype=torch.int32, device=pack.bias.device)
        pack._sel_cache[key] = tab
    return _RelVectors.apply(pack.w_src_t, pack.w_dst_t, pack.att_src, pack.att_dst, pack, tab, len(blocks), zero, pass_weights)


_MULTI_TRANSFORM = os.environ.get('KGW_MULTI_TRANSFORM', '1') != '0'     # (A/B: one launch per destination type as before)


def _transform_multi_forward(w_src_t, Z, blocks, bsum, out_blocks, gamma, stat, C):
    """All forward transforms of a layer through ONE kgw_linear_splitk_multi launch, or None when the layer does not qualify
    (a single destination type, a single relation into one, an unaligned operand, too many rows for the few-rows kernel)."""
    live = [(k, b) for k, b in enumerate(blocks) if b[3] > 0]
    if not _MULTI_TRANSFORM or len(live) < 2 or len(live) > 4 or len(live) != len(blocks):
        return None
    if any((hi - lo) < 2 or rows >= _SPLITK_MAX_ROWS for _, (lo, hi, z0, rows) in live):
        return None
    if Z.dtype != torch.float32 or Z.data_ptr() % 16 or w_src_t.data_ptr() % 16 or bsum.data_ptr() % 16 or (gamma is not None and stat is None):
        return None
    jobs = (_lib.KgwSplitKJob * len(live))()
    outs = []
    for j, (k, (lo, hi, z0, rows)) in zip(jobs, live):
        R = hi - lo
        x = Z[z0:z0 + rows * R].view(rows, R * C)
        ob = out_blocks[k] if out_blocks is not None else None
        y = ob.view() if ob is not None and ob.n == rows else torch.empty(rows, C, device=Z.device)
        W = w_src_t[lo:hi].view(R * C, C)
        if x.data_ptr() % 16 or y.data_ptr() % 16 or y.stride(0) % 4 or (gamma is not None and gamma[lo:hi].data_ptr() % 16):
            return None
        j.X, j.ldx, j.W, j.ldw, j.bias = x.data_ptr(), x.stride(0), W.data_ptr(), W.stride(0), bsum[k].data_ptr()
        j.Y, j.ldy, j.rows, j.K, j.N, j.relu, j.w_is_kn = y.data_ptr(), y.stride(0), rows, R * C, C, 1, 1
        if gamma is not None:
            j.seg_stat, j.gamma = stat.data_ptr() + 8 * z0, gamma[lo:hi].data_ptr()
        outs.append(y)
    _lib.check(_lib.lib().kgw_linear_splitk_multi(len(live), jobs, _lib.stream_ptr()), 'kgw_linear_splitk_multi')
    return outs


_MERGED_TRANSFORM_BWD = os.environ.get('KGW_MERGED_TRANSFORM_BWD', '1') != '0'     # (A/B: the three launches one after the other)


def _transform_bwd_merged(live_blocks, dW, db, dZ, w_src_t, gamma, stat, dgamma, C) -> bool:
    """_LayerTransform.backward's three kinds of work in one kgw_transform_bwd call; False (nothing launched) when a shape is
    outside what the merged kernel's blocks take."""
    n = len(live_blocks)
    if not 1 <= n <= 4:
        return False
    L = _lib.lib()
    tn = (_lib.KgwTnJob * n)()
    sk = (_lib.KgwSplitKJob * n)()
    cs = (_lib.KgwSplitKJob * n)()
    keep = []
    for q, (lo, hi, z0, rows, R, x, dz) in enumerate(live_blocks):
        out, bs = dW[lo:hi].view(R * C, C), db[lo:hi]
        A, B = dz, x                                         # C^T = (A^T B)^T with column sums of A: see _tn_gemm_group
        if not (A.dtype == torch.float32 and B.dtype == torch.float32 and A.stride(1) == 1 and B.stride(1) == 1 and out.stride(1) == 1 and
                bs.stride(1) == 1 and A.shape[0] == B.shape[0] and A.shape[0] > 0 and A.shape[1] % 2 == 0 and B.shape[1] % 2 == 0 and
                A.shape[1] >= 64 and B.shape[1] >= 64 and A.stride(0) % 2 == 0 and B.stride(0) % 2 == 0 and A.data_ptr() % 8 == 0 and
                B.data_ptr() % 8 == 0):
            return False
        if not (rows < _SPLITK_MAX_ROWS and dz.stride(0) % 4 == 0 and dz.data_ptr() % 16 == 0 and C == KGW_C):
            return False
        M, N = A.shape[1], B.shape[1]
        nws = int(L.kgw_tn_gemm_workspace_floats(rows, M, N))
        ws = torch.empty(nws, device=A.device)
        keep.append(ws)
        j = tn[q]
        j.A, j.lda, j.B, j.ldb, j.rows = _p(A), A.stride(0), _p(B), B.stride(0), rows
        j.C, j.ldc, j.colsum_a, j.colsum_ld = _p(out), out.stride(0), _p(bs), bs.stride(0)
        j.workspace, j.workspace_floats, j.rows_dev = _p(ws), nws, None
        j.M, j.N, j.c_transposed, j.colsum_repeat = M, N, 1, bs.shape[0]
        if dZ is not None:
            W = w_src_t[lo:hi].view(R * C, C)
            y = dZ[z0:z0 + rows * R].view(rows, R * C)
            if W.stride(0) % 4 or W.data_ptr() % 16 or y.stride(0) % 4 or y.data_ptr() % 16:
                return False
            k = sk[q]
            k.X, k.ldx, k.W, k.ldw, k.bias = dz.data_ptr(), dz.stride(0), W.data_ptr(), W.stride(0), None
            k.Y, k.ldy, k.rows, k.K, k.N, k.relu, k.w_is_kn = y.data_ptr(), y.stride(0), rows, C, R * C, 0, 0
        if gamma is not None:
            g = cs[q]
            g.seg_stat, g.Y, g.ldy, g.rows, g.K = stat.data_ptr() + 8 * z0, dz.data_ptr(), dz.stride(0), rows, R * C
            g.dgamma = dgamma[lo:hi].data_ptr()
    sink = GRAD_SINK
    if sink is not None and (_DEFER_REDUCE or sink.pending_fold is not None):
        # the second launch of these products is left pending (the gradients it finishes wait for the end of the backward pass
        # anyway) and the one an earlier layer left rides in this launch
        ride = sink.take_reduce()
        fold = sink.take_fold()                    # (the read-out node's second launch: one more block of this one)
        plan = _lib.KgwTnReducePlan()
        _lib.check(L.kgw_transform_bwd_ex(n, tn, n if dZ is not None else 0, sk, n if gamma is not None else 0, cs,
                                          _lib.C.byref(ride[0]) if ride is not None else None, _lib.C.byref(plan),
                                          _lib.C.byref(fold[0]) if fold is not None else None, _lib.stream_ptr()),
                   'kgw_transform_bwd_ex')         # (``C`` is the channel count here)
        if ride is not None:
            sink.reduces_ridden += 1
        if fold is not None:
            sink.folds_ridden += 1
        if plan.valid and (plan.blocks > _RIDE_MAX_BLOCKS or not _DEFER_REDUCE):
            # (too many blocks to ride: they would run a few at a time with the carrier's registers and LDS -- launched here)
            _lib.check(L.kgw_tn_reduce_launch(_lib.C.byref(plan), _lib.stream_ptr()), 'kgw_tn_reduce_launch')
        elif plan.valid:
            # (workspaces + the STORAGES of the gradients it writes: see _RelVectorsMulti.backward)
            sink.pending_reduce = (plan, keep, dW.untyped_storage(), db.untyped_storage())
            sink.records[db.data_ptr()] = (None, db.numel(), db.untyped_storage())      # (d bias goes straight to its parameter: a copy made on the way must be noticed)
        return True
    _lib.check(L.kgw_transform_bwd(n, tn, n if dZ is not None else 0, sk, n if gamma is not None else 0, cs, _lib.stream_ptr()),
               'kgw_transform_bwd')
    return True


class _LayerTransform(torch.autograd.Function):
    """h_d = relu([Z[:, r0] | Z[:, r1] | ...] @ [W_r0^T ; W_r1^T ; ...] + sum_r bias_r) for every destination
    type of a layer (lin_src of kgwas/conv.py:138/142 + bias :190 + HeteroConv sum model.py:74 + ReLU :75).
    ``Z`` is the aggregate's type-major output; ``blocks`` = [(lo, hi, z0, rows)]: relations [lo, hi) of the packed
    arrays feed the destination type whose block starts at Z row z0 and has ``rows`` destination rows."""

    @staticmethod
    def forward(ctx, w_src_t, bias, Z, blocks, bsum, out_blocks, premasked=False, gamma=None, stat=None):
        C = bias.shape[1]
        outs, ys = [], []
        # every destination type's transform of the layer in ONE launch (kgw_linear_splitk_multi) when there are several and
        # all are of the fused-forward kind: K = R * 128 > 128 rows-few problems that each leave most of the chip idle
        multi = _transform_multi_forward(w_src_t, Z, blocks, bsum, out_blocks, gamma, stat, C)
        if multi is not None:
            ctx.save_for_backward(w_src_t, Z, gamma, stat, *multi)
            ctx.blocks = blocks
            ctx.n_bias = bias.shape[0]
            ctx.premasked = premasked
            return tuple(multi)
        # bsum [n blocks, C]: bias of every relation into a type, summed (rel_vectors computes it in its launch)
        for k, (lo, hi, z0, rows) in enumerate(blocks):
            R = hi - lo
            x = Z[z0:z0 + rows * R].view(rows, R * C)
            ob = out_blocks[k] if out_blocks is not None else None
            out = ob.view() if ob is not None and ob.n == rows else None
            if gamma is not None and R > 1 and rows:
                # FC_output folded into this layer: + gamma[r] for every non-empty (row, relation) segment
                y = torch.empty(rows, C, device=Z.device) if out is None else out
                L = _lib.lib()
                nws = int(L.kgw_linear_splitk_workspace_floats(rows, R * C, C))
                ws = torch.empty(nws, device=Z.device)
                W = w_src_t[lo:hi].view(R * C, C)
                _lib.check(L.kgw_linear_splitk_ind(_p(x), x.stride(0), _p(W), W.stride(0), _p(bsum[k]), _p(y), y.stride(0), rows,
                                                   R * C, 1, stat.data_ptr() + 8 * z0, _p(gamma[lo:hi]), _p(ws), nws, None,
                                                   _lib.stream_ptr()), 'kgw_linear_splitk_ind')
            else:
                y = linear(x, w_src_t[lo:hi].view(R * C, C), bsum[k], relu=False if gamma is not None else True, w_kn=True, out=out)
                if gamma is not None and rows:      # (single relation into the type: no K split; rare, framework ops)
                    ind = (stat[z0:z0 + rows * R, 1] > 0).to(y.dtype).view(rows, R)
                    y = torch.relu_(y.add_((ind.unsqueeze(2) * gamma[lo:hi].unsqueeze(0)).sum(1)))
            outs.append(y)
            ys.append(y)
        ctx.save_for_backward(w_src_t, Z, gamma, stat, *ys)
        ctx.blocks = blocks
        ctx.n_bias = bias.shape[0]
        ctx.premasked = premasked
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dYs):
        w_src_t, Z, gamma, stat = ctx.saved_tensors[:4]
        ys = ctx.saved_tensors[4:]
        C = w_src_t.shape[-1]
        blocks = ctx.blocks
        covered = sum(hi - lo for lo, hi, _, _ in blocks)
        full = covered == w_src_t.shape[0] and all(d is not None for d in dYs)
        # (every relation belongs to a block that has a gradient: kgw_ind_colsum writes all of d gamma, no zero fill)
        dgamma = (torch.empty_like(gamma) if full else torch.zeros_like(gamma)) if gamma is not None else None
        dev = w_src_t.device
        dW = torch.empty_like(w_src_t) if full else torch.zeros_like(w_src_t)
        db = torch.empty(ctx.n_bias, C, device=dev) if full else torch.zeros(ctx.n_bias, C, device=dev)
        need_dz = ctx.needs_input_grad[2]
        z_rows = Z.shape[0]
        z_cov = sum(rows * (hi - lo) for lo, hi, _, rows in blocks)
        dZ = None
        if need_dz:
            dZ = torch.empty_like(Z) if (z_cov == z_rows and all(d is not None for d in dYs)) else torch.zeros_like(Z)
        live_blocks = []
        for k, ((lo, hi, z0, rows), dy) in enumerate(zip(blocks, dYs)):
            if dy is None or rows == 0:
                continue
            R = hi - lo
            x = Z[z0:z0 + rows * R].view(rows, R * C)
            # (premasked: the consumer of y already multiplied its gradient by (y > 0))
            dz = dy.contiguous() if ctx.premasked else torch.ops.aten.threshold_backward(dy.contiguous(), ys[k], 0.0)
            live_blocks.append((lo, hi, z0, rows, R, x, dz))
        # everything that is a function of dz alone -- weight / bias gradients, dZ twins, d gamma sums, of every destination type -- as
        # blocks of ONE launch (+ the split-K products' second) where the kernels' shape conditions hold (kgw_transform_bwd)
        if _MERGED_TRANSFORM_BWD and _transform_bwd_merged(live_blocks, dW, db, dZ if need_dz else None, w_src_t, gamma, stat, dgamma, C):
            return dW, db, dZ, None, None, None, None, dgamma, None
        # dWt = x^T dz lands transposed in place (the pack keeps [in, out]); db = colsum(dz) for each relation: the
        # destination types' products in ONE launch pair when the grouped kernel takes them
        if not _tn_gemm_group([(dz, x, dW[lo:hi].view(R * C, C), db[lo:hi]) for lo, hi, z0, rows, R, x, dz in live_blocks]):
            for lo, hi, z0, rows, R, x, dz in live_blocks:
                tn_gemm(dz, x, out=dW[lo:hi].view(R * C, C), transpose_out=True, colsum_out=db[lo:hi])
        if len(live_blocks) > 1 and len(live_blocks) <= 4 and _MULTI_TRANSFORM:
            # the dZ twins of all destination types in one launch, the d gamma sums in another
            L = _lib.lib()
            if need_dz and all(0 < rows < _SPLITK_MAX_ROWS and dz.stride(1) == 1 and dz.stride(0) % 4 == 0 and dz.data_ptr() % 16 == 0
                               for lo, hi, z0, rows, R, x, dz in live_blocks):
                jobs = (_lib.KgwSplitKJob * len(live_blocks))()
                for j, (lo, hi, z0, rows, R, x, dz) in zip(jobs, live_blocks):
                    W = w_src_t[lo:hi].view(R * C, C)
                    y = dZ[z0:z0 + rows * R].view(rows, R * C)
                    j.X, j.ldx, j.W, j.ldw, j.bias = dz.data_ptr(), dz.stride(0), W.data_ptr(), W.stride(0), None
                    j.Y, j.ldy, j.rows, j.K, j.N, j.relu, j.w_is_kn = y.data_ptr(), y.stride(0), rows, C, R * C, 0, 0
                _lib.check(L.kgw_linear_splitk_multi(len(live_blocks), jobs, _lib.stream_ptr()), 'kgw_linear_splitk_multi')
                need_dz = False
            if gamma is not None:
                jobs = (_lib.KgwSplitKJob * len(live_blocks))()
                for j, (lo, hi, z0, rows, R, x, dz) in zip(jobs, live_blocks):
                    j.seg_stat, j.Y, j.ldy, j.rows, j.K = stat.data_ptr() + 8 * z0, dz.data_ptr(), dz.stride(0), rows, R * C
                    j.dgamma = dgamma[lo:hi].data_ptr()
                _lib.check(L.kgw_ind_colsum_multi(len(live_blocks), jobs, _lib.stream_ptr()), 'kgw_ind_colsum_multi')
                gamma = None
        for lo, hi, z0, rows, R, x, dz in live_blocks:
            if need_dz:
                linear(dz, w_src_t[lo:hi].view(R * C, C), out=dZ[z0:z0 + rows * R].view(rows, R * C))
            if gamma is not None:
                _lib.check(_lib.lib().kgw_ind_colsum(stat.data_ptr() + 8 * z0, _p(dz), dz.stride(0), rows, R, _p(dgamma[lo:hi]),
                                                     _lib.stream_ptr()), 'kgw_ind_colsum')
        return dW, db, dZ, None, None, None, None, dgamma, None


def layer_transform(pack, Z, blocks, out_blocks=None, premasked=False, bias_sum=None, weight=None, gamma=None, stat=None):
    """``blocks`` = [(lo, hi, z0, rows)] (see _LayerTransform); ``out_blocks``: optional RowBlock per block to write
    the outputs into; ``premasked``: whoever consumes the outputs folds this node's ReLU backward into its own
    backward kernel (gat_aggregate(relu_input=True) / readout_weighted_mse(h_is_relu=True)), so no stand-alone
    threshold launch runs here; ``bias_sum``: per-block summed bias from rel_vectors(pack, blocks)."""
    if bias_sum is None:
        key = _block_key(blocks)
        sel = pack._sel_cache.get(key)
        if sel is None:
            sel = torch.zeros(len(blocks), pack.bias.shape[0], device=pack.bias.device)
            for k, (lo, hi) in enumerate(key):
                sel[k, lo:hi] = 1.0
            pack._sel_cache[key] = sel
        bias_sum = (sel.unsqueeze(2) * pack.bias.detach().unsqueeze(0)).sum(1)        # [blocks, n, 1] * [1, n, C]: no GEMM for a sum
    return _LayerTransform.apply(pack.w_src_t if weight is None else weight, pack.bias, Z, blocks, bias_sum, out_blocks, premasked,
                                 gamma, stat)


class _FoldFC(torch.autograd.Function):
    """fold_fc_output_hip's autograd node (kgw_fold_fwd / kgw_fold_bwd): two launches instead of ~40 framework ops."""

    @staticmethod
    def forward(ctx, w_src_t, U, V, pack, tab, *fc):
        n_mlp = len(fc) // 2
        n, NR = w_src_t.shape[0], U.shape[0]
        dev = U.device
        U = U.contiguous(); V = V.contiguous()
        Up = torch.empty_like(U); Vp = torch.empty_like(V)
        kappa = torch.empty(NR, device=dev)
        Wp = torch.empty_like(w_src_t)
        gamma = torch.empty(n, KGW_C, device=dev)
        a = _lib.KgwFoldArgs()
        a.n, a.n_rels, a.n_mlp = n, NR, n_mlp
        a.rel_ids_host, a.src_mlp_host, a.dst_mlp_host = tab[0].ctypes.data, tab[1].ctypes.data, tab[2].ctypes.data
        a.w_src_t, a.U, a.V = _p(w_src_t), _p(U), _p(V)
        for m in range(n_mlp):
            a.fc_weight[m], a.fc_bias[m] = _p(fc[2 * m]), _p(fc[2 * m + 1])
        a.Up, a.Vp, a.kappa, a.Wp, a.gamma = _p(Up), _p(Vp), _p(kappa), _p(Wp), _p(gamma)
        q = PARAM_RIDERS
        job = None
        if q is not None and q.fold is None and q.relvec is not None:
            # the fold rides with the relation vectors it reads: its U / V must BE the U_full / V of one of the pending jobs
            n_rv, jobs = q.relvec[0], q.relvec[1]
            job = next((k for k in range(n_rv) if jobs[k].U_full == U.data_ptr() and jobs[k].V == V.data_ptr() and
                        jobs[k].n_rels_total == NR), None)
        if job is not None:
            q.fold = (a, job, (w_src_t, U, V, fc, Up, Vp, kappa, Wp, gamma, tab))
        else:
            if q is not None:
                q.flush()                                # (its inputs are pending: launch them first, the ordinary way)
            _lib.check(_lib.lib().kgw_fold_fwd(C.byref(a), _lib.stream_ptr()), 'kgw_fold_fwd')
        ctx.save_for_backward(w_src_t, U, V, *fc)
        ctx.tab, ctx.n_mlp = tab, n_mlp
        ctx.set_materialize_grads(False)
        return Up, Vp, kappa, Wp, gamma

    @staticmethod
    def backward(ctx, dUp, dVp, dkappa, dWp, dgamma):
        w_src_t, U, V = ctx.saved_tensors[:3]
        fc = ctx.saved_tensors[3:]
        tab, n_mlp = ctx.tab, ctx.n_mlp
        n, NR = w_src_t.shape[0], U.shape[0]
        dev = U.device

        def z(g, ref_shape):
            return g.contiguous() if g is not None else torch.zeros(ref_shape, device=dev)
        dUp, dVp = z(dUp, U.shape), z(dVp, V.shape)
        dkappa, dWp, dgamma = z(dkappa, (NR,)), z(dWp, w_src_t.shape), z(dgamma, (n, KGW_C))
        dU = torch.empty_like(U); dV = torch.empty_like(V)
        dws = torch.empty_like(w_src_t)
        dfc = [torch.empty_like(t) for t in fc]
        a = _lib.KgwFoldArgs()
        a.n, a.n_rels, a.n_mlp = n, NR, n_mlp
        a.rel_ids_host, a.src_mlp_host, a.dst_mlp_host = tab[0].ctypes.data, tab[1].ctypes.data, tab[2].ctypes.data
        a.w_src_t, a.U, a.V = _p(w_src_t), _p(U), _p(V)
        for m in range(n_mlp):
            a.fc_weight[m], a.fc_bias[m] = _p(fc[2 * m]), _p(fc[2 * m + 1])
            a.d_fc_weight[m], a.d_fc_bias[m] = _p(dfc[2 * m]), _p(dfc[2 * m + 1])
        a.dUp, a.dVp, a.dkappa, a.dWp, a.dgamma = _p(dUp), _p(dVp), _p(dkappa), _p(dWp), _p(dgamma)
        pu, pv = _take_pieces(dUp), _take_pieces(dVp)
        if (pu is None) != (pv is None):
            raise GradSinkMismatch('d u_r and d v_r of a layer must both be pieces or both be complete')
        if pu is not None:                          # the aggregate left eight pieces per value: added inside k_fold_bwd
            a.dUp, a.dVp, a.duv_pieces = pu[0], pv[0], 1
        a.dU, a.dV, a.dws = _p(dU), _p(dV), _p(dws)
        sink = GRAD_SINK
        if sink is not None and _PARAM_TAIL and _DEFER_PRODUCTS and sink.fold_bwd is None and sink.relvec_bwd is None:
            # its outputs reach the relation vectors' backward (queued the same way, behind it) and the optimiser, nothing else:
            # launched by GradSink.flush (outputs: storages only, see _RelVectorsMulti.backward)
            keep = (dUp, dVp, dkappa, dWp, dgamma, w_src_t, U, V, fc, tab, pu, pv, [t.untyped_storage() for t in (dU, dV, dws)])
            sink.fold_bwd = (a, keep, [(t.data_ptr(), t.numel(), t.untyped_storage()) for t in dfc])
        else:
            if sink is not None:
                sink.launch_pending_tail()
                sink.launch_pending_reduce()
            _lib.check(_lib.lib().kgw_fold_bwd(C.byref(a), _lib.stream_ptr()), 'kgw_fold_bwd')
        return (dws, dU, dV, None, None) + tuple(dfc)


def fold_fc_output_hip(pack, U, V, fc_params, tab, weight=None):
    """Fold the last Linear of the feature MLPs, H = h2 T_m + c_m (FC_output, kgwas/model.py:15,21; m = the MLP of the node's
    type), into the layer-1 relation parameters -- exact, like aggregate-then-transform: H enters GATConv (which has no
    root term) only linearly, as the message sum_j alpha_ij H_j and through the logit projections <H_j, u_r>, <H_i, v_r>
    (kgwas/conv.py:150-152,227-228).  With layer 1 running on h2:
        U'_r = T_src U_r,  V'_r = T_dst V_r,  kappa_r = <c_src, U_r> + <c_dst, V_r>         (logits)
        W'_r = T_src W_r^T (the packed [in, out] form),  gamma_r = c_src W_r^T                (transform; gamma is added
        wherever the segment is not empty: sum_j alpha_ij = 1)
    on the HIP kernels kgw_fold_fwd / kgw_fold_bwd (tests/helpers.py:fold_fc_output_reference is the same in framework ops,
    float64).  ``fc_params``: [weight_0, bias_0, weight_1, bias_1, ...] = FC_output of the MLPs (nn.Linear layout); ``tab`` =
    (rel ids, source MLP, destination MLP) of the packed relations as int32 numpy arrays.
    Returns (U' [n_rels,C], V' [n_rels,C], kappa [n_rels] by relation id; W' [n,C,C], gamma [n,C] by packed slot)."""
    return _FoldFC.apply(pack.w_src_t if weight is None else weight, U, V, pack, tab, *fc_params)


# ------------------------------------------------------------------------------------------------------
# LD-score weighted loss of a step (kgwas/kgwas.py:139-145) as one node
# ------------------------------------------------------------------------------------------------------
class _WeightedMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, n_id, y_all, w_all):
        pred = pred.contiguous()
        n = pred.numel()
        assert pred.dtype == torch.float32 and n_id.dtype == torch.int32 and n_id.numel() >= n
        assert y_all.dtype == torch.float32 and w_all.dtype == torch.float64
        loss = torch.empty((), dtype=torch.float64, device=pred.device)
        _lib.check(_lib.lib().kgw_wmse_fwd(_p(pred), _p(n_id), _p(y_all), _p(w_all), n, _p(loss), _lib.stream_ptr()),
                   'kgw_wmse_fwd')
        ctx.save_for_backward(pred, n_id, y_all, w_all)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        pred, n_id, y_all, w_all = ctx.saved_tensors
        gloss = gloss.contiguous().to(torch.float64)
        dpred = torch.empty_like(pred)
        _lib.check(_lib.lib().kgw_wmse_bwd(_p(pred), _p(n_id), _p(y_all), _p(w_all), pred.numel(), _p(gloss), _p(dpred),
                                           _lib.stream_ptr()), 'kgw_wmse_bwd')
        return dpred, None, None, None


def weighted_mse(pred, n_id, y_all, w_all):
    """mean(w_all[n_id] * (pred - y_all[n_id])**2) (float64) for the first ``pred.numel()`` entries of ``n_id``
    (int32 global ids of the seeds); y_all float32 [N], w_all float64 [N]."""
    return _WeightedMSE.apply(pred, n_id, y_all, w_all)


_UNIT_GRADS = {}          # device -> resident float64 1.0 (the loss gradient of a plain ``loss.backward()``)


def unit_gradient(device) -> torch.Tensor:
    """A resident float64 1.0 on ``device``: ``loss.backward(gradient=unit_gradient(dev))`` is ``loss.backward()`` without the
    ones_like fill, and lets the fused read-out node recognise the gradient it precomputed for (``unit_grad=True``)."""
    dev = torch.device(device)
    t = _UNIT_GRADS.get(dev)
    if t is None:
        t = _UNIT_GRADS[dev] = torch.ones((), dtype=torch.float64, device=dev)
    return t


class _ReadoutWeightedMSE(torch.autograd.Function):
    """loss = mean(w[n_id] * ([relu](H[:n] @ w_lin^T + b_lin) - y[n_id])**2): the read-out Linear(128 -> 1) of the seed
    rows (kgwas/model.py:86) and the weighted MSE (kgwas/kgwas.py:139-145) as one node, two launches per step."""

    @staticmethod
    def forward(ctx, H, w_lin, b_lin, n_id, y_all, w_all, n, relu, h_is_relu=False, unit_grad=False):
        H = H.contiguous()
        assert H.dtype == torch.float32 and H.shape[1] == KGW_C and H.shape[0] >= n and w_lin.numel() == KGW_C
        assert n_id.dtype == torch.int32 and y_all.dtype == torch.float32 and w_all.dtype == torch.float64
        dev = H.device
        pred = torch.empty(n, device=dev)
        loss = torch.empty((), dtype=torch.float64, device=dev)
        terms = torch.empty(n, dtype=torch.float64, device=dev)
        ctx.n, ctx.relu, ctx.h_is_relu = n, relu, h_is_relu
        ctx.mark_non_differentiable(pred)
        ctx.set_materialize_grads(False)
        ctx.ready = None
        if unit_grad and ctx.needs_input_grad[0]:
            # the caller backpropagates a loss gradient of exactly 1: everything the backward returns is computed here
            dH, dw, db = torch.empty_like(H), torch.empty_like(w_lin), torch.empty(1, device=dev)
            part = torch.empty(((H.shape[0] + 3) // 4) * (KGW_C + 1), device=dev)
            ctx.fold = None
            if _DEFER_READOUT_FOLD and READOUT_FOLD_DEFERRED:
                # first launch only: the fold of the partial sums waits for the backward pass, where it rides in the next launch
                # of a captured step (GradSink.pending_fold) or is launched first thing otherwise.  The loss is NOT written until
                # then -- only a caller that always runs the backward takes this form (``readout_fold_deferred``)
                f = _lib.KgwReadoutFold()
                _lib.check(_lib.lib().kgw_readout_wmse_train_parts(_p(H), _p(w_lin), _p(b_lin), _p(n_id), _p(y_all), _p(w_all), n,
                                                                   H.shape[0], (1 if relu else 0) | (2 if h_is_relu else 0), _p(pred),
                                                                   _p(loss), _p(dH), _p(dw), _p(db), _p(terms), _p(part), C.byref(f),
                                                                   _lib.stream_ptr()), 'kgw_readout_wmse_train_parts')
                ctx.fold = (f, (part, terms, loss))
            else:
                _lib.check(_lib.lib().kgw_readout_wmse_train(_p(H), _p(w_lin), _p(b_lin), _p(n_id), _p(y_all), _p(w_all), n, H.shape[0],
                                                             (1 if relu else 0) | (2 if h_is_relu else 0), _p(pred), _p(loss), _p(dH),
                                                             _p(dw), _p(db), _p(terms), _p(part), _lib.stream_ptr()),
                           'kgw_readout_wmse_train')
            ctx.ready = (dH, dw, db)
            return loss, pred
        _lib.check(_lib.lib().kgw_readout_wmse_fwd(_p(H), _p(w_lin), _p(b_lin), _p(n_id), _p(y_all), _p(w_all), n,
                                                   1 if relu else 0, _p(pred), _p(loss), _p(terms), _lib.stream_ptr()),
                   'kgw_readout_wmse_fwd')
        ctx.save_for_backward(H, w_lin, pred, n_id, y_all, w_all)
        return loss, pred

    @staticmethod
    def backward(ctx, gloss, _gpred):
        if gloss is None:
            return (None,) * 10
        if ctx.ready is not None:
            dH, dw, db = ctx.ready
            ctx.ready = None
            unit = _UNIT_GRADS.get(gloss.device)
            is_unit = unit is not None and gloss.data_ptr() == unit.data_ptr()
            fold, ctx.fold = getattr(ctx, 'fold', None), None
            if fold is not None:
                sink = GRAD_SINK
                if sink is not None and is_unit and sink.pending_fold is None:
                    # (d w_lin / d b_lin go straight to their parameters: storages only + a record, see _RelVectorsMulti.backward)
                    sink.pending_fold = (fold[0], fold[1], dw.untyped_storage(), db.untyped_storage())
                    sink.records[dw.data_ptr()] = (None, dw.numel(), dw.untyped_storage())
                    sink.records[db.data_ptr()] = (None, db.numel(), db.untyped_storage())
                else:
                    _lib.check(_lib.lib().kgw_readout_train_fold(C.byref(fold[0]), _lib.stream_ptr()), 'kgw_readout_train_fold')
            if not is_unit:
                # the caller did NOT backpropagate the resident 1.0 (a scaled loss, gradient accumulation, plain loss.backward()
                # with its ones_like): the precomputed gradients are for a loss gradient of 1 -- scale them
                k = gloss.to(torch.float32)
                dH, dw, db = dH * k, dw * k, db * k
            return dH, dw, db, None, None, None, None, None, None, None
        H, w_lin, pred, n_id, y_all, w_all = ctx.saved_tensors
        gloss = gloss.contiguous().to(torch.float64)
        dH = torch.empty_like(H)
        dw = torch.empty_like(w_lin)
        db = torch.empty(1, device=H.device)
        part = torch.empty(((H.shape[0] + 3) // 4) * (KGW_C + 1), device=H.device)
        _lib.check(_lib.lib().kgw_readout_wmse_bwd(_p(H), _p(w_lin), _p(pred), _p(n_id), _p(y_all), _p(w_all), ctx.n,
                                                   H.shape[0], (1 if ctx.relu else 0) | (2 if ctx.h_is_relu else 0), _p(gloss),
                                                   _p(dH), _p(dw), _p(db), _p(part), _lib.stream_ptr()),
                   'kgw_readout_wmse_bwd')
        return dH, dw, db, None, None, None, None, None, None, None


def readout_weighted_mse(H, w_lin, b_lin, n_id, y_all, w_all, n: int, relu: bool = True, h_is_relu: bool = False,
                         unit_grad: bool = False):
    """Returns (loss float64 scalar, pred float32 [n]); ``w_lin`` [1,128] / ``b_lin`` [1] = HeteroGNN.lin.
    ``h_is_relu``: see layer_transform's ``premasked``.  ``unit_grad``: the caller expects to backpropagate a loss gradient of
    exactly 1: forward and backward of this node then share two launches instead of four.  Passing the resident
    ``unit_gradient(device)`` to ``backward(gradient=...)`` takes the precomputed gradients as they are; any other loss gradient
    (``(k * loss).backward()``) multiplies them -- correct either way."""
    return _ReadoutWeightedMSE.apply(H, w_lin, b_lin, n_id, y_all, w_all, int(n), bool(relu), bool(h_is_relu), bool(unit_grad))
