"""Multi-GPU: one process per GPU over torch.distributed (backend "nccl" == RCCL on ROCm, xGMI).

The path shards by SEEDS: the seeds of a batch are independent given the weights,
so each rank expands / aggregates its own slice of every batch against its own resident copy of the
graph, and the only exchange is ONE all-reduce of the parameter gradients per step
(~2.6 M floats ~= 10.5 MB in fast mode) in a single flat bucket -- on the point-to-point xGMI mesh a
single large message beats many per-parameter ones.  With equal slices,
mean-over-ranks(mean-over-slice) == mean over the reference's 512-seed batch, so the SGD trajectory is
the reference's (up to summation order)."""
from __future__ import annotations

import os
import weakref

import numpy as np
import torch
import torch.distributed as dist


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_batches(ids: np.ndarray, batch_size: int, rank: int, world: int) -> np.ndarray:
    """Seeds of rank ``rank``: slice ``rank`` (of ``world`` equal slices) of every full batch of
    ``batch_size`` ids, in batch order (drop_last semantics of kgwas/kgwas.py:93)."""
    ids = np.asarray(ids)
    if world == 1:
        return ids
    if batch_size % world:
        raise ValueError(f'batch_size {batch_size} must be divisible by world size {world}')
    nb = len(ids) // batch_size
    per = batch_size // world
    return ids[:nb * batch_size].reshape(nb, world, per)[:, rank].reshape(-1)


def _grads(model):
    return [p.grad for p in model.parameters() if p.grad is not None]


COLLECTIVES = {}          # name -> [calls, bytes this rank handed to the collective] (bench.py reports them per step)


def count_collective(name: str, nbytes: int):
    c = COLLECTIVES.setdefault(name, [0, 0])
    c[0] += 1
    c[1] += int(nbytes)


class CollectiveTimer:
    """``with CollectiveTimer() as ct:`` -- device time of every torch.distributed collective issued inside, by HIP events on the
    stream the call is made on (a synchronous collective makes that stream wait for the backend's own: the second event lands
    after the data has).  bench.py wraps its timed steps in one so that an N > 1 line carries MEASURED milliseconds per collective
    next to the bytes; nothing else uses it.  Keys: '<op>(<bytes of the tensor handed in> B)'."""
    OPS = ('all_reduce', 'all_gather', 'all_gather_into_tensor', 'reduce_scatter_tensor', 'broadcast')

    def __init__(self):
        self.events, self._orig = [], {}

    def __enter__(self):
        for op in self.OPS:
            fn = getattr(dist, op, None)
            if fn is None:
                continue
            self._orig[op] = fn
            setattr(dist, op, self._wrap(op, fn))
        return self

    def _wrap(self, op, fn):
        def timed(*a, **k):
            t = next((x for x in a if torch.is_tensor(x)), None)
            if t is None and a and isinstance(a[0], (list, tuple)) and a[0] and torch.is_tensor(a[0][0]):
                t = a[1] if len(a) > 1 and torch.is_tensor(a[1]) else a[0][0]
            if t is None or not t.is_cuda or k.get('async_op'):
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            self.events.append((f'{op}({t.numel() * t.element_size()} B)', e0, e1))
            return r
        return timed

    def __exit__(self, *exc):
        for op, fn in self._orig.items():
            setattr(dist, op, fn)
        return False

    def summary(self, per: int = 1):
        """{key: {'calls_per_step', 'ms_per_call', 'ms_per_step'}} with ``per`` = the number of steps the block ran."""
        torch.cuda.synchronize()
        out = {}
        for key, e0, e1 in self.events:
            d = out.setdefault(key, [0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
        return {k: {'calls_per_step': n / per, 'ms_per_call': ms / n, 'ms_per_step': ms / per} for k, (n, ms) in out.items()}


def allreduce_flat(flat: torch.Tensor, world: int, what: str = 'parameter gradients'):
    """Average ``flat`` over the ranks in place: ONE collective (RCCL's AVG where the backend has it)."""
    if world <= 1 and not (os.environ.get('KGW_FORCE_MULTIRANK_PATH') == '1' and dist.is_initialized()):
        return
    count_collective(f'all_reduce_avg({what})', flat.numel() * flat.element_size())
    if dist.get_backend() in ('nccl', 'fake'):       # ("fake": bench.py --as-rank, collectives that move nothing)
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
    else:                                            # gloo (CPU tests) has no AVG
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)


_LIVE = weakref.WeakKeyDictionary()   # model -> liveness of its requires_grad parameters, as agreed over the ranks
_FLAG = weakref.WeakKeyDictionary()   # model -> (pinned host copy of the last bucket's "new live parameter" flag, its event)
_SEEN = weakref.WeakKeyDictionary()   # model -> parameters that had a gradient on THIS rank since the last agreement (sticky)


class _Done:
    def synchronize(self):
        pass


def allreduce_grads(model, world: int):
    """Average the parameter gradients over ranks: one flat bucket, one all-reduce.  The averaged gradients are
    handed back as VIEWS of the bucket (``p.grad`` re-pointed: no copy-back launch per parameter).
    The bucket covers a FIXED list -- every parameter with requires_grad, zero where this rank's backward produced no gradient (a
    node type absent from its slice of the batch) -- so the collective has the same size on every rank; a parameter is live (keeps
    a gradient, is stepped by Adam) if it was on ANY rank at the first reduction, else its gradient stays None as in the
    reference (kgwas/kgwas.py:150-151 skips parameters without a gradient)."""
    params = [p for p in model.parameters() if p.requires_grad]
    if not params:
        return
    multi = world > 1 or (os.environ.get('KGW_FORCE_MULTIRANK_PATH') == '1' and dist.is_initialized())
    if not multi:
        return
    had = [p.grad is not None for p in params]
    live = _LIVE.get(model)
    # liveness is agreed at the first reduction and RE-agreed whenever a rank sees a gradient on a parameter the agreement left
    # out (a node type or relation absent from every rank's first batch that a later batch reaches): one 1-element MAX
    # all-reduce per step keeps the decision to re-agree itself collective -- every rank runs the same sequence
    # (round 5: the "somebody saw a new live parameter" flag rides as one more element of the gradient bucket and is read back
    #  WITHOUT stalling the step -- an asynchronous copy, looked at by the next call: no extra collective, no host sync per step;
    #  the re-agreement happens one step after the event, on every rank at the same step since the flag is the reduced one)
    fresh = live is None or len(live) != len(params)
    pend = _FLAG.get(model)
    if not fresh and pend is not None:
        pend[1].synchronize()                      # (the copy of the PREVIOUS step's flag: long finished)
        fresh = bool(float(pend[0][0]) > 0.0)
    mine_new = (not fresh) and any(h and not l for h, l in zip(had, live))
    # (the re-agreement runs one step AFTER the step that raised the flag and votes with that later step's gradients: the parameter
    #  that raised it is remembered here -- a relation that shows up on non-consecutive batches would otherwise never become live)
    seen = _SEEN.get(model)
    if seen is None or len(seen) != len(params):
        seen = [False] * len(params)
    seen = _SEEN[model] = [s or h for s, h in zip(seen, had)]
    if fresh:
        t = torch.tensor([seen[i] or (live is not None and len(live) == len(params) and live[i]) for i in range(len(params))],
                         dtype=torch.int32, device=params[0].device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        live = _LIVE[model] = [bool(v) for v in t.cpu().tolist()]
        _SEEN[model] = [False] * len(params)
    sel = [p for p, l in zip(params, live) if l]
    for p, l in zip(params, live):
        if not l:
            p.grad = None            # (never stepped on a local, unreduced gradient)
    if not sel:
        return
    flat = torch.cat([p.grad.reshape(-1) if p.grad is not None else p.new_zeros(p.numel()) for p in sel] +
                     [sel[0].new_full((1,), 1.0 if mine_new else 0.0)])
    allreduce_flat(flat, world)
    host = pend[0] if pend is not None else (torch.zeros(1, dtype=flat.dtype).pin_memory() if flat.is_cuda else torch.zeros(1, dtype=flat.dtype))
    host.copy_(flat[-1:], non_blocking=True)
    ev = torch.cuda.Event() if flat.is_cuda else None
    if ev is not None:
        ev.record()
    _FLAG[model] = (host, ev if ev is not None else _Done())
    off = 0
    for p in sel:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n


def check_same_on_all_ranks(value: int, what: str):
    """Raise on EVERY rank if ``value`` differs between ranks (a mismatch in a later collective's size would hang instead)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([value, -value], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t[0]) != -int(t[1]):
        raise RuntimeError(f'{what} differs between ranks (max {int(t[0])}, min {-int(t[1])}): the ranks would issue collectives of '
                           'different sizes')


def broadcast_params(model, src: int = 0):
    """Rank ``src``'s parameters (and persistent floating-point buffers) to every rank, one flat bucket per dtype.  The
    non-persistent int32 / int64 index tables of the relation packs are structural -- identical on every rank by
    construction -- and are NOT sent (flattening them with the float parameters would round-trip them through fp32)."""
    skip = set()
    for mod in model.modules():
        for name in getattr(mod, '_non_persistent_buffers_set', ()):
            b = mod._buffers.get(name)
            if b is not None:
                skip.add(id(b))
    tensors = [p.data for p in model.parameters()] + [b.data for b in model.buffers() if id(b) not in skip]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for group in by_dtype.values():
        flat = torch._utils._flatten_dense_tensors(group)
        dist.broadcast(flat, src)
        for t, f in zip(group, torch._utils._unflatten_dense_tensors(flat, group)):
            t.copy_(f)
