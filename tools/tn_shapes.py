"""Shapes and stand-alone times of the weight-gradient products (kgw_tn_gemm*) of one benchmark training step."""
import sys, contextlib, collections, numpy as np, torch
sys.path.insert(0, '.')
from kgwas_amd import ops
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data
from kgwas_amd.sampler import NeighborLoader
from kgwas_amd.optim import FusedAdam
with contextlib.redirect_stdout(sys.stderr):
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_bench_0')
run = KGWAS(data, device='cuda:0', seed=1); run.initialize_model()
ld_w = run._ld_weight_vector(); ids = np.asarray(data.train_input_nodes[1])[:512 * 8]
opt = FusedAdam(run.model.parameters(), lr=1e-4, weight_decay=5e-4)
it = iter(NeighborLoader(data.data, [-1, -1], ('SNP', ids), batch_size=512, drop_last=True, device='cuda:0'))
for _ in range(3):
    b = next(it); run.train_step(b, opt, ld_w, 1)
print('nodes of a batch:', dict(b.n_nodes))
torch.cuda.synchronize()
log = collections.OrderedDict()
def timed(name, f, key):
    def g(*a, **k):
        kk = (name,) + key(*a, **k)
        r = f(*a, **k)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record(); f(*a, **k); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
        log.setdefault(kk, []).append(min(ts))
        return r
    return g
ops.tn_gemm = timed('tn_gemm', ops.tn_gemm, lambda A, B, *a, **k: (tuple(A.shape), tuple(B.shape)))
ops._tn_gemm_group = timed('tn_group', ops._tn_gemm_group, lambda jobs: tuple((tuple(A.shape), tuple(B.shape)) for A, B, _, _ in jobs))
run.train_step(next(it), opt, ld_w, 1)
tot = 0
for k, v in log.items():
    print('%8.1f us  %s' % (np.mean(v), k)); tot += np.mean(v)
print('total us/step', tot)
