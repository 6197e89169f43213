import sys, time, numpy as np, torch, contextlib
sys.path.insert(0, '.')
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data
from kgwas_amd.sampler import NeighborLoader
with contextlib.redirect_stdout(sys.stderr):
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, data_path='/tmp/kgwas_prof')
run = KGWAS(data, device='cuda:0', seed=1); run.initialize_model()
opt = torch.optim.Adam(run.model.parameters(), lr=1e-4, weight_decay=5e-4, capturable=True)
ld_w = run._ld_weight_vector()
ids = np.asarray(data.train_input_nodes[1])[:512*30]
it = iter(NeighborLoader(data.data, [-1,-1], ('SNP', ids), batch_size=512, drop_last=True, device='cuda:0'))
run.model.train()
for _ in range(5): run.train_step(next(it), opt, ld_w)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5): run.train_step(next(it), opt, ld_w)
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print('%-60s %6s %10s' % ('op', 'n/step', 'cuda us/step'))
for e in rows[:60]:
    ct = getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0))
    print('%-60s %6.1f %10.1f' % (e.key[:60], e.count/5, ct/5))
