"""Proxy for splitting the step into graphs that run side by side: the gene feature MLP (forward + backward, kgw_gemm3 products)
as one captured graph, the SNP + GO feature MLPs as another, replayed alone and concurrently on two streams."""
import sys, contextlib, time, numpy as np, torch
sys.path.insert(0, '.')
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data
from kgwas_amd.sampler import NeighborLoader
with contextlib.redirect_stdout(sys.stderr):
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_bench_0')
run = KGWAS(data, device='cuda:0', seed=1); run.initialize_model()
model = run.model; model.train()
ids = np.asarray(data.train_input_nodes[1])[:512]
batch = next(iter(NeighborLoader(data.data, [-1, -1], ('SNP', ids), batch_size=512, device='cuda:0')))
xd = batch.x_dict
all_types = list(model.node_types)

def embed(types):
    model.node_types = types
    try:
        return model._embed_all(batch, xd, None, True)
    finally:
        model.node_types = all_types

sets = {'gene': ['Gene'], 'snp+go': [t for t in all_types if t != 'Gene']}
if len(sys.argv) > 1:
    sets = {k: v for k, v in sets.items() if k in sys.argv[1:]}
torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
params = {'gene': [p for n, p in model.gene_feat_mlp.named_parameters() if 'FC_output' not in n],
          'snp+go': [p for m in (model.snp_feat_mlp, model.go_feat_mlp) for n, p in m.named_parameters() if 'FC_output' not in n]}
grads, keep = {}, {}
for k, ts in sets.items():
    h = embed(ts)
    grads[k] = {t: torch.randn_like(v) for t, v in h.items()}
    del h          # (a live autograd graph keeps the parameters' AccumulateGrad nodes on THIS stream: fatal inside a capture on another)
torch.cuda.synchronize()

def work(k):
    h = embed(sets[k])
    keep[k] = torch.autograd.grad(list(h.values()), params[k], [grads[k][t] for t in h], allow_unused=True)

graphs, streams = {}, {}
for k in sets:
    s = torch.cuda.Stream(); streams[k] = s
    with torch.cuda.stream(s):
        work(k); work(k); torch.cuda.synchronize(); print('capturing', k, flush=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            work(k)
        graphs[k] = g

def timeit(keys, n=100):
    def once():
        for k in keys:
            with torch.cuda.stream(streams[k]):
                graphs[k].replay()
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6

if len(sets) < 2:
    print({k: timeit([k]) for k in sets}); sys.exit(0)
a, b = timeit(['gene']), timeit(['snp+go'])
ab = timeit(['gene', 'snp+go'])
print('gene MLP fwd+bwd alone %.1f us   SNP+GO MLPs fwd+bwd alone %.1f us   sum %.1f   side by side %.1f us' % (a, b, a + b, ab))
