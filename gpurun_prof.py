import cProfile, pstats, io, sys, time, numpy as np, torch, contextlib
sys.path.insert(0, '.')
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data
from kgwas_amd.sampler import NeighborLoader
with contextlib.redirect_stdout(sys.stderr):
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, data_path='/tmp/kgwas_prof')
run = KGWAS(data, device='cuda:0', seed=1); run.initialize_model()
opt = torch.optim.Adam(run.model.parameters(), lr=1e-4, weight_decay=5e-4)
ld_w = run._ld_weight_vector()
ids = np.asarray(data.train_input_nodes[1])[:512*80]
it = iter(NeighborLoader(data.data, [-1,-1], ('SNP', ids), batch_size=512, drop_last=True, device='cuda:0'))
run.model.train()
for _ in range(5): run.train_step(next(it), opt, ld_w)
torch.cuda.synchronize()
def timed(fn):
    torch.cuda.synchronize(); t=time.perf_counter(); r=fn(); t_cpu=(time.perf_counter()-t)*1e3; torch.cuda.synchronize(); return r, t_cpu, (time.perf_counter()-t)*1e3
for rep in range(3):
    b, c0, t_s = timed(lambda: next(it))
    opt.zero_grad(set_to_none=True)
    out, c1, t_f = timed(lambda: run.model(b.x_dict, b.edge_index_dict, 512))
    n_id = b.n_id('SNP')[:512].long(); y = b.dg.y['SNP'][n_id]; w = ld_w[n_id]
    loss = torch.mean(w * (out.reshape(-1) - y) ** 2)
    _, c2, t_b = timed(lambda: loss.backward())
    _, c3, t_o = timed(lambda: opt.step())
    print('ms cpu-enqueue/total: next %.2f/%.2f fwd %.2f/%.2f bwd %.2f/%.2f opt %.2f/%.2f' % (c0,t_s,c1,t_f,c2,t_b,c3,t_o), flush=True)
t=time.perf_counter()
for _ in range(30): run.train_step(next(it), opt, ld_w)
torch.cuda.synchronize(); print('30 steps wall ms/step', (time.perf_counter()-t)/30*1e3)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(10): run.train_step(next(it), opt, ld_w)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=40, max_name_column_width=50)[:9000])
