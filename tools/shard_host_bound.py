import os, sys, time, contextlib
os.environ['KGW_FORCE_MULTIRANK_PATH']='1'
os.environ.setdefault('MASTER_ADDR','127.0.0.1'); os.environ.setdefault('MASTER_PORT','29577'); os.environ['RANK']='0'; os.environ['WORLD_SIZE']='1'
sys.path.insert(0,'.')
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda:0'))
from kgwas_amd.kgwas import KGWAS
from kgwas_amd.kgwas_data import KGWAS_Data
from kgwas_amd.shard import ShardedTrainer
with contextlib.redirect_stdout(sys.stderr):
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_bench_0')
run = KGWAS(data, device='cuda:0', seed=1); run.initialize_model(); run.model.train()
ids = np.asarray(data.train_input_nodes[1])[:512*60]
for ov in ('0','1'):
    os.environ['KGW_SHARD_OVERLAP_SAMPLING']=ov
    st = ShardedTrainer(run, ('SNP', ids), 512, use_graph=True)
    for i in range(5): st.step(i)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for i in range(5,55): st.step(i)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print('overlap', ov, 'host enqueue per step %.3f ms, total per step %.3f ms' % ((t1-t0)/50*1e3, (t2-t0)/50*1e3), 'segments', len(st.comp_seg[0].items), len(st.samp_seg[0].items))
dist.destroy_process_group()
