"""Which Python lines issue the small framework kernels of a training step (torch.profiler, eager step)."""
import sys, contextlib, collections, numpy as np, torch
sys.path.insert(0, '.')  # run from the repo root
from torch.profiler import profile, ProfilerActivity
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data
from kgwas_amd.sampler import NeighborLoader
with contextlib.redirect_stdout(sys.stderr):
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_bench_0')
run = KGWAS(data, device='cuda:0', seed=1); run.initialize_model()
ld_w = run._ld_weight_vector(); ids = np.asarray(data.train_input_nodes[1])[:512 * 8]
from kgwas_amd.optim import FusedAdam
opt = FusedAdam(run.model.parameters(), lr=1e-4, weight_decay=5e-4)
it = iter(NeighborLoader(data.data, [-1, -1], ('SNP', ids), batch_size=512, drop_last=True, device='cuda:0'))
for _ in range(3):
    run.train_step(next(it), opt, ld_w, 1)
torch.cuda.synchronize()
b = next(it)
from torch.autograd.profiler import record_function
from kgwas_amd import ops, model as kmodel
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        with record_function('KGW:' + name):
            return f(*a, **k)
    setattr(obj, name, g)
for n in ('rel_vectors', 'gat_aggregate', 'layer_transform', 'mlp_tail', 'linear_relu'):
    wrap(ops, n)
for n in ('_embed_all', '_fused_layers', '_embed'):
    wrap(kmodel.HeteroGNN, n)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run.train_step(b, opt, ld_w, 1)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type.name == 'CPU']
ranges = [e for e in evs if e.name.startswith('KGW:') or e.name.startswith('autograd::engine::evaluate_function')]
agg = collections.Counter(); tim = collections.Counter()
for ev in evs:
    if not ev.name.startswith('aten::') or not ev.kernels:
        continue
    t0, t1 = ev.time_range.start, ev.time_range.end
    inner = None
    for r in ranges:
        if r.time_range.start <= t0 and r.time_range.end >= t1:
            if inner is None or r.time_range.start >= inner.time_range.start:
                inner = r
    where = inner.name.replace('autograd::engine::evaluate_function: ', 'bwd ') if inner else '(top level)'
    key = (ev.name, where[:60])
    agg[key] += len(ev.kernels); tim[key] += sum(k.duration for k in ev.kernels)
for key, n in sorted(agg.items(), key=lambda kv: -tim[kv[0]])[:70]:
    print(f'{tim[key]:8.1f} us  x{n:3d}  {key[0]:28s} {key[1]}')
