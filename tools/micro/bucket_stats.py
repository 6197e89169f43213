import sys, contextlib, numpy as np, torch
sys.path.insert(0, '.')
from kgwas_amd.kgwas_data import KGWAS_Data
from kgwas_amd.sampler import NeighborLoader
with contextlib.redirect_stdout(sys.stderr):
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_bench_0')
ids = np.asarray(data.train_input_nodes[1])[:512]
batch = next(iter(NeighborLoader(data.data, [-1, -1], ('SNP', ids), batch_size=512, device='cuda:0')))
m, buf, sc = batch.meta, batch.buf, batch.dg.schema
for l in (1, 2):
    TR = int(m.t_base[l - 1][sc.NT]); tp = buf.t_ptr[l - 1][:TR + 1].cpu().numpy()
    print('layer', l, 'rows', TR, 'entries', tp[-1], 'n_chunks', int(m.n_chunks[l-1]), 'longest row', int(np.diff(tp).max()))
    for sh in (8, 9, 10):
        edges = np.arange(0, TR + (1 << sh), 1 << sh).clip(max=TR)
        cnt = np.diff(tp[edges])
        print('  sh', sh, 'buckets', len(cnt), 'mean %.0f' % cnt.mean(), 'max', cnt.max(), 'top5', np.sort(cnt)[-5:], '>2000:', int((cnt > 2000).sum()), ' sum of top-1024-th..', )
