import sys, contextlib, numpy as np, torch, collections
sys.path.insert(0,'.')
from kgwas_amd import ops
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data
from kgwas_amd.sampler import NeighborLoader
with contextlib.redirect_stdout(sys.stderr):
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_bench_0')
run = KGWAS(data, device='cuda:0', seed=1); run.initialize_model()
ld_w = run._ld_weight_vector(); ids = np.asarray(data.train_input_nodes[1])[:512*8]
opt = torch.optim.Adam(run.model.parameters(), lr=1e-4, weight_decay=5e-4)
it = iter(NeighborLoader(data.data, [-1,-1], ('SNP', ids), batch_size=512, drop_last=True, device='cuda:0'))
for _ in range(3): run.train_step(next(it), opt, ld_w, 1)
torch.cuda.synchronize()
log = collections.OrderedDict()
def wrap(mod, name, keyf):
    f = getattr(mod, name)
    def g(*a, **k):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record(); r = f(*a, **k); e1.record(); torch.cuda.synchronize()
        log.setdefault((name,)+keyf(*a,**k), []).append(e0.elapsed_time(e1)*1e3)
        return r
    setattr(mod, name, g)
wrap(ops, 'linear', lambda X,W,bias=None,relu=False,mask=None,w_kn=False: (tuple(X.shape), tuple(W.shape), w_kn, mask is not None))
wrap(ops, 'tn_gemm', lambda A,B,*a,**k: (tuple(A.shape), tuple(B.shape)))
import torch.nn.functional as F
_mm = torch.Tensor.__matmul__
for _ in range(2): run.train_step(next(it), opt, ld_w, 1)
tot=0
for k,v in log.items():
    print(f'{np.mean(v):8.1f} us x{len(v)//2}/step  {k}'); tot+=sum(v)/2
print('total us/step', tot)
