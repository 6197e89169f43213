"""FusedAdam: torch.optim.Adam(lr, betas, eps, weight_decay) semantics (kgwas/kgwas.py:116) in ONE launch over all
parameter tensors (kgw_adam), with the step counter on the device so it can sit inside a captured HIP graph."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class FusedAdam:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.state = {}
        dev = self.params[0].device if self.params else torch.device('cuda')
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.packed_images = {}        # parameter -> persistent kgw_gemm3 operand image its fused update keeps current (step_fused)
        self.param_groups = [{'params': self.params, 'lr': lr, 'betas': betas, 'eps': eps, 'weight_decay': weight_decay}]

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _state(self, p):
        st = self.state.get(p)
        if st is None:      # like torch: state is created the first time a parameter has a gradient
            st = {'exp_avg': torch.zeros_like(p, memory_format=torch.preserve_format),
                  'exp_avg_sq': torch.zeros_like(p, memory_format=torch.preserve_format)}
            self.state[p] = st
        return st

    def init_state(self):
        for p in self.params:
            if p.requires_grad:
                self._state(p)

    @torch.no_grad()
    def step_fused(self, sink, meta_ptr=None, n_layers=0, n_hops=0, stats=None):
        """The optimiser launch of a captured single-GPU step (kgw_adam_fused): the update of every parameter that has a gradient,
        the last reduction of the gradients whose producers left partial sums with ``sink`` (ops.GradSink), the step counter and
        -- ``meta_ptr``: device KgwBatchMeta of the batch, ``stats``: the trainer's int64 totals -- kgw_accumulate_stats_tick, in ONE
        launch.  Raises ops.GradSinkMismatch (nothing launched) when the fused form does not apply; the caller then runs the
        unfused step."""
        from . import ops
        if sink is not None:
            sink.flush()                     # the deferred weight-gradient products: one launch, just ahead of this one
        live = [p for p in self.params if p.grad is not None]
        recs = {}
        for p in live:
            r = sink.take(p.grad) if sink is not None else None
            if r is not None and r[0] is not None:           # (r[0] None: a deferred product that came out complete -- only matched)
                recs[p] = r
        if sink is not None and sink.records:
            n = len(sink.records)
            sink.records.clear()
            raise ops.GradSinkMismatch(f'{n} deferred gradient(s) did not reach a parameter as written')
        # a weight whose kgw_gemm3 operand image this launch keeps current must arrive as a KGW_GRAD_G3T record: updated from any other
        # kind of gradient (a complete tensor, a TN product, something autograd accumulated) the weight would change and the image the
        # next forward reads would not -- the HIP launches do not move the tensor's version counter, so nobody would notice
        for p in live:
            if p in self.packed_images and (p not in recs or recs[p][0].kind != 5):
                raise ops.GradSinkMismatch('the gradient of a weight with a persistent kgw_gemm3 operand image did not arrive as the '
                                           'partial sums of kgw_gemm3_partial: its image would go stale')
        if len(live) > _lib.ADAM_FUSED_MAX or len(recs) > _lib.ADAM_FUSED_SRC:
            raise ops.GradSinkMismatch(f'{len(live)} tensors / {len(recs)} deferred gradients exceed the fused launch\'s tables')
        live.sort(key=lambda p: 0 if p in recs else 1)           # the longer work units first
        n = len(live)
        if not hasattr(self, 'done_dev'):
            self.done_dev = torch.zeros(_lib.ADAM_FUSED_COUNTERS, dtype=torch.int32, device=self.step_dev.device)
        P = (C.c_void_p * n)(); G = (C.c_void_p * n)(); M = (C.c_void_p * n)(); V = (C.c_void_p * n)()
        N = (C.c_int64 * n)()
        S = (_lib.KgwGradSrc * max(n, 1))()
        for k, p in enumerate(live):
            g = p.grad
            if not (p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32 and g.dtype == torch.float32):
                raise _lib.KgwasHipError('FusedAdam needs contiguous fp32 parameters and gradients')
            st = self._state(p)
            P[k], G[k], M[k], V[k], N[k] = p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel()
            if p in recs:
                C.memmove(C.byref(S[k]), C.byref(recs[p][0]), C.sizeof(_lib.KgwGradSrc))
                img = self.packed_images.get(p)
                if img is not None and S[k].kind == 5:               # KGW_GRAD_G3T: the updated weight's operand image rides along
                    S[k].packed, S[k].flip = img.data_ptr(), int(_lib.lib().kgw_gemm3_flip())
        rc = _lib.lib().kgw_adam_fused(n, P, G, M, V, N, S, self.step_dev.data_ptr(), self.lr, self.betas[0], self.betas[1], self.eps,
                                       self.weight_decay, meta_ptr, n_layers, n_hops, stats.data_ptr() if stats is not None else None,
                                       self.done_dev.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, 'kgw_adam_fused')
        # (``recs`` -- and with it the partial-sum workspaces -- lives until here: the launch that reads them is enqueued)

    @torch.no_grad()
    def finish_into(self, sink, pairs):
        """Multi-GPU step: ``pairs`` = [(parameter, its slot in a flat all-reduce bucket)].  One launch (kgw_grad_finish) writes every
        parameter's FINISHED gradient into its slot -- a copy of ``p.grad`` where that is complete, the sum of the partial records its
        producer left with ``sink`` otherwise -- instead of the producers' second launches + the bucket's concatenation.  Raises
        ops.GradSinkMismatch (nothing launched) like step_fused."""
        from . import ops
        sink.flush()
        recs = {}
        for p, _ in pairs:
            r = sink.take(p.grad)
            if r is not None and r[0] is not None:
                recs[p] = r
        if sink.records:
            n = len(sink.records)
            sink.records.clear()
            raise ops.GradSinkMismatch(f'{n} deferred gradient(s) did not reach a parameter as written')
        if len(recs) > _lib.ADAM_FUSED_SRC:
            raise ops.GradSinkMismatch(f'{len(recs)} deferred gradients exceed the fused launch\'s table')
        pairs = sorted(pairs, key=lambda pd: 0 if pd[0] in recs else 1)
        for i in range(0, len(pairs), _lib.ADAM_FUSED_MAX):
            chunk = pairs[i:i + _lib.ADAM_FUSED_MAX]
            n = len(chunk)
            D = (C.c_void_p * n)(); G = (C.c_void_p * n)(); N = (C.c_int64 * n)()
            S = (_lib.KgwGradSrc * n)()
            for k, (p, dst) in enumerate(chunk):
                g = p.grad
                if not (g.is_contiguous() and dst.is_contiguous() and g.dtype == torch.float32 and dst.dtype == torch.float32 and
                        dst.numel() == g.numel()):
                    raise _lib.KgwasHipError('finish_into needs contiguous fp32 gradients and bucket slots of the same size')
                D[k], G[k], N[k] = dst.data_ptr(), g.data_ptr(), g.numel()
                if p in recs:
                    C.memmove(C.byref(S[k]), C.byref(recs[p][0]), C.sizeof(_lib.KgwGradSrc))
            _lib.check(_lib.lib().kgw_grad_finish(n, D, G, N, S, _lib.stream_ptr()), 'kgw_grad_finish')

    @torch.no_grad()
    def step(self, grads=None, tick=True):
        """``grads``: optional {parameter: gradient tensor} to use instead of ``p.grad`` (the all-reduced bucket's
        views in the multi-GPU step); parameters missing from it are skipped.  ``tick=False``: the caller advances
        ``step_dev`` itself after this call (kgw_accumulate_stats_tick in the captured step: one launch less)."""
        if grads is not None:
            live = [p for p in self.params if p in grads]
        else:
            live = [p for p in self.params if p.grad is not None]     # grad None => skipped, exactly like torch
        if not live:
            return
        for i in range(0, len(live), 64):
            chunk = live[i:i + 64]
            n = len(chunk)
            P = (C.c_void_p * n)(); G = (C.c_void_p * n)(); M = (C.c_void_p * n)(); V = (C.c_void_p * n)()
            N = (C.c_int64 * n)()
            for k, p in enumerate(chunk):
                g = grads[p] if grads is not None else p.grad
                if not (p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32 and g.dtype == torch.float32):
                    raise _lib.KgwasHipError('FusedAdam needs contiguous fp32 parameters and gradients')
                st = self._state(p)
                P[k], G[k], M[k], V[k], N[k] = p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel()
            # chunks > 0 must not advance the step counter again: only the last call ticks it
            last = i + 64 >= len(live)
            fn = _lib.lib().kgw_adam if (tick and last) else _lib.lib().kgw_adam_notick      # only the last chunk ticks
            rc = fn(n, P, G, M, V, N, self.step_dev.data_ptr(), self.lr, self.betas[0], self.betas[1],
                    self.eps, self.weight_decay, _lib.stream_ptr())
            _lib.check(rc, 'kgw_adam')
