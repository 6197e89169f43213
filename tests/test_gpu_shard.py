"""-m gpu, world_size 2 and 4 (all ranks on cuda:0, gloo collectives): the SNP-SHARDED mode on the HIP path
(kgwas_amd/shard.py; SURVEY.md 8e-ii, BASELINE.json north_star / configs[3]) against the single-process run of the same
batches with the same weights:

* every rank predicts exactly the seeds it owns, and those predictions equal the single-process predictions;
* the SUM over ranks of the parameter gradients equals the single-process gradient of the 512-style batch loss --
  for every parameter, including the replicated gene / GO MLPs and the relations whose softmax is split across ranks;
* after the flat all-reduce + Adam the ranks hold bit-identical parameters, equal to the single-process step;
* the merged attention statistics equal the single-process ones (partial online-softmax merge, hub gene included).
The graphs: SynthKG at 1 % scale (hub genes above KGW_CHUNK = 128 in-edges, so multi-chunk partial states are merged
too) and the hand-made corner-case graph (empty relation, duplicate edges, zero-degree seeds)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

BS, STEPS, N_GRAD_BATCHES = 64, 2, 6


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_run(which, seed=11, random_bias=True, no_relu=False, device='cuda:0'):
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.kgwas_data import KGWAS_Data
    if which == 'small':
        data = KGWAS_Data.from_synthetic(scale=0.01, seed=1, feat_dims={'Gene': 96}, data_path=f'/tmp/kgwas_gpushard_{os.getpid()}')
    else:
        from tests.conftest import make_edge_case_graph
        g, d = make_edge_case_graph()
        data = KGWAS_Data(f'/tmp/kgwas_gpushard_edge_{os.getpid()}')
        data.data = g
        data.snp_init_dim_size, data.gene_init_dim_size, data.go_init_dim_size = d['SNP'], d['Gene'], 16
        n = int(g['SNP'].x.shape[0])
        rng = np.random.default_rng(5)
        data.all_ids = np.arange(n)
        data.ldsc_weight = 0.5 + rng.random(n)
        data.train_input_nodes = ('SNP', rng.permutation(n))
    run = KGWAS(data, device=device, seed=seed)
    run.initialize_model(no_relu=no_relu)
    if random_bias:
        with torch.no_grad():                               # non-zero relation biases: exercise their path
            g_ = torch.Generator(device='cpu').manual_seed(3)
            for pack in list(run.model.live_packs) + list(run.model.dead_packs):
                pack.bias.copy_(torch.randn(pack.bias.shape, generator=g_) * 0.1)
    return data, run


def _ids(data):
    return np.asarray(data.train_input_nodes[1])[:BS * max(STEPS + 1, N_GRAD_BATCHES)]


def _worker(rank, world, port, out_dir, which, backend='gloo'):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dev = 'cuda:0'
    if backend == 'nccl-per-device':            # the real thing: one rank per GPU over RCCL / xGMI (boxes with >= world devices)
        dev = f'cuda:{rank}'
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(dev))
    elif backend == 'nccl':                     # RCCL on a 1-GPU box: ONE rank that still issues every collective
        torch.cuda.set_device(0)
        os.environ['KGW_FORCE_MULTIRANK_PATH'] = '1'
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda:0'))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from kgwas_amd import dist as kdist
        from kgwas_amd.shard import ShardedTrainer
        data, run = _make_run(which, device=dev)
        kdist.broadcast_params(run.model)
        run.model.train()
        st = ShardedTrainer(run, ('SNP', _ids(data)), BS, lr=1e-3, weight_decay=5e-4)
        # step 0 without the optimiser: predictions, loss share, gradients, merged softmax statistics
        batch, part, pred = st.forward_backward(0)
        own = (batch.n_id('SNP')[:batch.batch_size].long() + st.lo).cpu()
        grads = {k: (None if v is None else v.detach().cpu().clone()) for k, v in run.model.named_reference_tensors(grad=True).items()}
        rec = {'own': own, 'pred': pred.detach().cpu(), 'part': part.cpu(), 'grads': grads, 'lo': st.lo, 'hi': st.hi,
               'genes_l1': batch.n_id('Gene')[:int(batch.meta.n_rows[0][batch.dg.schema.type_id['Gene']])].cpu()}
        # all-reduced gradients of a few more batches at the SAME parameters (isolates the exchange from Adam's
        # amplification of fp32 noise)
        rec['summed'] = []
        for i in range(1, N_GRAD_BATCHES):
            st.forward_backward(i)
            st.allreduce_grads()
            if rank == 0:
                rec['summed'].append({k: (None if v is None else v.detach().cpu().clone())
                                      for k, v in run.model.named_reference_tensors(grad=True).items()})
        # then full training steps
        for i in range(STEPS):
            st.step(i)
        rec['params'] = {k: v.detach().cpu() for k, v in run.model.named_reference_tensors().items()}
        rec['pred_all'] = st.predict(_ids(data)[:3 * BS // 2]).cpu()
        rec['bytes_moved'] = st.xchg.bytes_moved
        torch.save(rec, os.path.join(out_dir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('which,world,backend', [('small', 2, 'gloo'), ('small', 4, 'gloo'), ('edge', 2, 'gloo'), ('small', 1, 'nccl'),
                                                 ('small', 2, 'nccl-per-device'), ('edge', 2, 'nccl-per-device')])
def test_sharded_mode_equals_single_process(tmp_path, which, world, backend):
    """(the 'nccl' case: the exchange's collectives -- MIN all-reduce of the frontier flags, all-gather of the partial softmax
    states, SUM all-reduce of dZ rows / gradients / predictions -- issued over RCCL by the one rank a 1-GPU box can host;
    'nccl-per-device' (VERDICT r5 item 6): one rank PER GPU over RCCL, the same assertions -- skipped on a box with fewer GPUs
    than ranks)"""
    if backend == 'nccl-per-device' and torch.cuda.device_count() < world:
        pytest.skip(f'needs {world} GPUs, this box has {torch.cuda.device_count()}')
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path), which, backend), nprocs=world, join=True, start_method='spawn')
    recs = [torch.load(os.path.join(tmp_path, f'rank{r}.pt'), weights_only=False) for r in range(world)]

    from kgwas_amd.optim import FusedAdam
    from kgwas_amd.sampler import NeighborLoader
    data, run = _make_run(which)
    run.model.train()
    ids = _ids(data)
    ld_w = run._ld_weight_vector()

    def single(i):
        batch = next(iter(NeighborLoader(data.data, [-1, -1], ('SNP', ids[i * BS:(i + 1) * BS]), batch_size=BS, device='cuda:0')))
        run.model.zero_grad(set_to_none=True)
        loss, pred = run.model.forward_loss(batch.x_dict, batch.edge_index_dict, BS, batch.n_id('SNP'), batch.dg.y['SNP'], ld_w)
        loss.backward()
        return batch, loss.detach().cpu(), pred.detach().cpu()

    batch, loss, pred = single(0)
    # 1. ownership: the ranks' seed sets partition the batch, in batch order inside a rank
    seeds0 = ids[:BS]
    got = np.concatenate([r['own'].numpy() for r in recs])
    assert sorted(got.tolist()) == sorted(seeds0.tolist())
    pos = {int(s): k for k, s in enumerate(seeds0)}
    for r in recs:
        assert all(r['lo'] <= int(s) < r['hi'] for s in r['own'])
        idx = torch.tensor([pos[int(s)] for s in r['own']])
        assert torch.allclose(r['pred'].double(), pred[idx].double(), rtol=1e-4, atol=1e-5), 'sharded prediction differs'
        # every rank expanded the same hop-1 genes, in the same order
        assert torch.equal(r['genes_l1'], recs[0]['genes_l1'])
    # 2. loss shares add up to the batch loss
    total = sum(float(r['part']) for r in recs)
    assert abs(total - float(loss)) <= 1e-5 * abs(float(loss)) + 1e-8, (total, float(loss))
    # 3. SUM over ranks of the gradients == single-process gradients
    ref = {k: (None if v is None else v.detach().cpu()) for k, v in run.model.named_reference_tensors(grad=True).items()}
    n_live = 0
    for k, g in ref.items():
        parts = [r['grads'][k] for r in recs]
        if g is None:
            assert all(p is None for p in parts), k
            continue
        assert all(p is not None for p in parts), k
        s = sum(p.double() for p in parts)
        scale = float(g.double().abs().max())
        err = float((s - g.double()).abs().max())
        assert err <= 2e-4 * scale + 1e-6, (k, err, scale)
        n_live += 1
    assert n_live > 30
    for i, summed in enumerate(recs[0]['summed'], start=1):
        single(i)
        refi = {k: (None if v is None else v.detach().cpu()) for k, v in run.model.named_reference_tensors(grad=True).items()}
        for k, g in refi.items():
            if g is None:
                assert summed[k] is None, k
                continue
            scale = float(g.double().abs().max())
            err = float((summed[k].double() - g.double()).abs().max())
            assert err <= 2e-4 * scale + 1e-6, (i, k, err, scale)
    # 4. training steps: ranks bit-identical, equal to the single-process steps
    for k in recs[0]['params']:
        for r in recs[1:]:
            assert torch.equal(recs[0]['params'][k], r['params'][k]), f'{k}: ranks diverged'
    opt = FusedAdam(run.model.parameters(), lr=1e-3, weight_decay=5e-4)
    for i in range(STEPS):
        single(i)
        opt.step()
    refp = {k: v.detach().cpu() for k, v in run.model.named_reference_tensors().items()}
    for k in refp:
        d = float((recs[0]['params'][k].double() - refp[k].double()).abs().max())
        scale = float(refp[k].double().abs().max())
        assert d <= 5e-5 * max(scale, 1e-6) + 3e-6, (k, d, scale)
    # 5. sharded inference over an id list that is not a multiple of the batch size
    run.model.eval()
    with torch.no_grad():
        want = []
        q = ids[:3 * BS // 2]
        for a in range(0, len(q), BS):
            b = next(iter(NeighborLoader(data.data, [-1, -1], ('SNP', q[a:a + BS]), batch_size=len(q[a:a + BS]), device='cuda:0')))
            want.append(run.model(b.x_dict, b.edge_index_dict, len(q[a:a + BS])).reshape(-1).cpu())
        want = torch.cat(want)
    for r in recs:
        assert torch.allclose(r['pred_all'].double(), want.double(), rtol=1e-4, atol=1e-5)
    assert recs[0]['bytes_moved'] > 0


def _train_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        # (linear read-out: a ReLU read-out that dies during the epoch would make every prediction 0 and the Pearson r undefined)
        data, run = _make_run('small', seed=21, random_bias=False, no_relu=True)
        run.train(batch_size=BS, epoch=1, save_best_model=False, save_name=f'shard{rank}', parallelism='shard')
        torch.save({'val': run.val_metrics, 'test': run.test_metrics, 'pred': np.asarray(run.kgwas_res['pred'].values),
                    'kgwas_p': np.asarray(run.kgwas_res['KGWAS_P'].values)}, os.path.join(out_dir, f'train{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('variant', ['pinned_order_fp32_pipe', 'pinned_order_shipped_pipe', 'shipped'])
def test_kgwas_train_in_sharded_mode_matches_single_process_training(tmp_path, monkeypatch, variant):
    """KGWAS.train(parallelism='shard') over two ranks vs KGWAS.train on one process: same data, same initial weights, one epoch, up
    to fp32 summation order.

    This toy problem has no signal (validation Pearson ~0.02) and several relation weights whose true gradient is zero:
    Adam divides their rounding noise by eps, so ANY change of summation order random-walks them (measured: 7e-9 after
    step 0, 5e-5 after step 4, 4e-3 after the epoch) and moves the validation MSE by ~0.1 %.  Three variants:
    * ``pinned_order_fp32_pipe``: summation order pinned on both sides -- the backward's 8-rows-per-wavefront path off (it groups a
      rank's LOCAL source rows, so it orders the a_src term differently in the sharded and the single-process layout) and the
      128-wide weight-gradient products on the fp32 matrix pipe (the bf16 x 3 pipe takes 16 rows per MFMA step: a rank's rows sit
      in other groups of 16 than the single process's) -- held to north_star's bar: validation Pearson within 1e-3, MSE within
      1e-3, predictions within tolerance;
    * ``pinned_order_shipped_pipe``: the same with the SHIPPED pipe of those products (bf16 x 3).  Gradients
      of a step then differ from the fp32 pipe's by 3e-7 (measured), and that alone random-walks the zero-gradient weights like any
      other change of order: the bounds are what that noise was measured to do, with a factor ~2.5 of head-room;
    * ``shipped``: every default (short-row path on), held to what the noise allows (the path itself is compared with the general
      one in test_gpu_aggregate.py)."""
    from kgwas_amd import _lib, ops
    short_rows = variant == 'shipped'
    split = variant != 'pinned_order_fp32_pipe'
    monkeypatch.setenv('KGW_SHORT_ROWS', '1' if short_rows else '0')          # the spawned ranks
    monkeypatch.setattr(ops, '_SHORT_ROWS', short_rows)                        # this process
    monkeypatch.setenv('KGW_TN_SPLIT', '1' if split else '0')
    was = _lib.lib().kgw_tn_split(1 if split else 0)
    try:
        _sharded_vs_single(tmp_path, variant)
    finally:
        _lib.lib().kgw_tn_split(was)


def _sharded_vs_single(tmp_path, variant):
    world = 2
    port = _free_port()
    mp.start_processes(_train_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    r0 = torch.load(os.path.join(tmp_path, 'train0.pt'), weights_only=False)
    r1 = torch.load(os.path.join(tmp_path, 'train1.pt'), weights_only=False)
    assert np.array_equal(r0['pred'], r1['pred']) and r0['val']['pearsonr'] == r1['val']['pearsonr']
    data, run = _make_run('small', seed=21, random_bias=False, no_relu=True)
    run.train(batch_size=BS, epoch=1, save_best_model=False, save_name='single', use_graph=False)
    assert np.isfinite(run.val_metrics['pearsonr'])
    ref = np.asarray(run.kgwas_res['pred'].values)
    d_r = abs(float(r0['val']['pearsonr']) - float(run.val_metrics['pearsonr']))
    d_val = abs(float(r0['val']['mse']) - float(run.val_metrics['mse'])) / float(run.val_metrics['mse'])
    d_test = abs(float(r0['test']['mse']) - float(run.test_metrics['mse'])) / float(run.test_metrics['mse'])
    d_pred = float(np.abs(r0['pred'] - ref).max())
    print(f'[sharded vs single, {variant}] |d Pearson| {d_r:.2e}  rel d val MSE {d_val:.2e}  rel d test MSE {d_test:.2e}  '
          f'max |d pred| {d_pred:.2e}  corr {np.corrcoef(r0["pred"], ref)[0, 1]:.6f}')
    if variant == 'shipped':
        assert d_val < 1e-2 and d_test < 1e-2
        assert np.corrcoef(r0['pred'], ref)[0, 1] > 0.999
        return
    if variant == 'pinned_order_shipped_pipe':
        # measured (round 6): |d Pearson| 7.1e-3 (at a Pearson of ~0.02 on ~800 validation SNPs), validation MSE 1.15e-3, test MSE
        # 3.0e-4 relative, max |d pred| 5.1e-2, correlation of the predictions 0.99941 -- the same figures as the `shipped` variant:
        # on this graph the pipe of the weight-gradient products is the whole difference
        assert d_r < 2e-2 and d_val < 3e-3 and d_test < 1e-3
        assert np.corrcoef(r0['pred'], ref)[0, 1] > 0.999
        return
    assert d_r < 1e-3 and d_val < 1e-3 and d_test < 1e-3
    assert np.allclose(r0['pred'], ref, rtol=2e-3, atol=2e-4)


def _golden_run():
    """The committed tiny case (tests/golden/gat_case.py) as a KGWAS run: its graph, its fixed parameters, its LD weights."""
    from collections import OrderedDict
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.kgwas_data import KGWAS_Data
    from tests.golden import gat_case as gc
    from tests.golden_io import case_graph
    g, w_all = case_graph()
    data = KGWAS_Data(f'/tmp/kgwas_gpushard_golden_{os.getpid()}')
    data.data = g
    data.snp_init_dim_size, data.gene_init_dim_size, data.go_init_dim_size = gc.DIMS['SNP'], gc.DIMS['Gene'], gc.DIMS['GO']
    data.all_ids = np.arange(gc.NODES['SNP'])
    data.ldsc_weight = w_all.numpy()
    data.train_input_nodes = ('SNP', gc.SEEDS.copy())
    run = KGWAS(data, device='cuda:0', seed=1)
    run.initialize_model()
    sd = OrderedDict((k, torch.from_numpy(v)) for k, v in gc.parameters(list(g.edge_types)).items())
    run.model.load_state_dict(sd, strict=True)
    return data, run


def _golden_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from kgwas_amd.shard import ShardedTrainer
        from tests.golden import gat_case as gc
        data, run = _golden_run()
        run.model.train()
        st = ShardedTrainer(run, ('SNP', gc.SEEDS.copy()), gc.BATCH, lr=1e-3, weight_decay=5e-4)
        batch, part, pred = st.forward_backward(0)
        own = (batch.n_id('SNP')[:batch.batch_size].long() + st.lo).cpu()
        real = st.loss_scale[0] > 0
        st.allreduce_grads()
        grads = {k: (None if v is None else v.detach().cpu().clone()) for k, v in run.model.named_reference_tensors(grad=True).items()}
        torch.save({'own': own if real else own[:0], 'pred': pred.detach().cpu() if real else pred.detach().cpu()[:0], 'part': part.cpu(),
                    'grads': grads, 'collectives': dict(st.xchg.collectives)}, os.path.join(out_dir, f'gold{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_sharded_mode_reproduces_the_committed_golden_vectors(tmp_path, world):
    """The SNP-sharded HIP path held to tests/golden/gat_small.npz directly (not to another mode of the same build): every rank's
    predictions of the seeds it owns, the sum of the loss shares, and the all-reduced gradient of every parameter.  With 4 ranks
    over 400 SNPs one rank owns seeds on one side of the hub gene only and the partial softmax states of three relations
    (hub row of 300 in-edges split over the ranks, an empty relation, duplicate edges) are merged."""
    from tests.golden import gat_case as gc
    from tests.golden_io import golden
    port = _free_port()
    mp.start_processes(_golden_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    recs = [torch.load(os.path.join(tmp_path, f'gold{r}.pt'), weights_only=False) for r in range(world)]
    G = golden()
    pos = {int(s): k for k, s in enumerate(gc.SEEDS)}
    seen = []
    for r in recs:
        idx = [pos[int(s)] for s in r['own']]
        seen += idx
        want = torch.from_numpy(G['pred'])[idx]
        assert torch.allclose(r['pred'].double(), want, rtol=1e-4, atol=1e-5), (r['pred'], want)
    assert sorted(seen) == list(range(gc.BATCH))
    total = sum(float(r['part']) for r in recs)
    assert abs(total - float(G['loss'])) <= 1e-5 * abs(float(G['loss'])) + 1e-7
    none = set(G['grad_none'].tolist())
    stride = int(G['grad_stride'])
    n = 0
    for r in recs:                                   # after the all-reduce every rank holds the full-batch gradient
        for name, g in r['grads'].items():
            if name in none or g is None:
                assert g is None or float(g.abs().max()) == 0.0 or name not in none, name
                continue
            g = g.double().numpy()
            if f'g_{name}' in G.files:
                ref = G[f'g_{name}']
                assert np.allclose(g.reshape(ref.shape), ref, rtol=1e-4, atol=2e-4 * max(np.abs(ref).max(), 1e-6) + 1e-7), name
            elif f'gs_{name}' in G.files:
                ref = G[f'gs_{name}']
                assert np.allclose(g.reshape(-1)[::stride], ref, rtol=1e-4, atol=2e-4 * max(np.abs(ref).max(), 1e-6) + 1e-7), name
            else:
                continue
            n += 1
    assert n > 60 * world
    assert any('all_gather' in k for k in recs[0]['collectives']) and any('gradients' in k for k in recs[0]['collectives'])


def _noseed_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from kgwas_amd.shard import ShardedTrainer
        from tests.golden import gat_case as gc
        data, run = _golden_run()
        run.model.train()
        seeds = gc.SEEDS[gc.SEEDS < 200].copy()                 # every seed belongs to rank 0 of 2
        st = ShardedTrainer(run, ('SNP', seeds), len(seeds), lr=1e-3, weight_decay=5e-4)
        assert st.loss_scale[0] == (1.0 if rank == 0 else 0.0)
        batch, part, pred = st.forward_backward(0)
        st.allreduce_grads()
        grads = {k: (None if v is None else v.detach().cpu().clone()) for k, v in run.model.named_reference_tensors(grad=True).items()}
        st.opt.step()
        torch.save({'part': part.cpu(), 'pred': pred.detach().cpu(), 'grads': grads,
                    'params': {k: v.detach().cpu() for k, v in run.model.named_reference_tensors().items()},
                    'pred_all': st.predict(seeds).cpu()}, os.path.join(out_dir, f'ns{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_rank_that_owns_no_seed_of_a_batch_still_takes_part(tmp_path):
    """A batch whose seeds all lie in rank 0's id range: rank 1 expands a node of its own with loss weight 0, joins every
    collective, contributes its SNP-side partial softmax states -- and the step equals the single-process step."""
    from kgwas_amd.sampler import NeighborLoader
    from tests.golden import gat_case as gc
    port = _free_port()
    mp.start_processes(_noseed_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method='spawn')
    r0, r1 = (torch.load(os.path.join(tmp_path, f'ns{r}.pt'), weights_only=False) for r in range(2))
    data, run = _golden_run()
    run.model.train()
    seeds = gc.SEEDS[gc.SEEDS < 200].copy()
    n = len(seeds)
    batch = next(iter(NeighborLoader(data.data, [-1, -1], ('SNP', seeds), batch_size=n, device='cuda:0')))
    loss, pred = run.model.forward_loss(batch.x_dict, batch.edge_index_dict, n, batch.n_id('SNP'), batch.dg.y['SNP'], run._ld_weight_vector())
    loss.backward()
    assert float(r1['part']) == 0.0 and abs(float(r0['part']) - float(loss)) <= 1e-5 * abs(float(loss))
    assert torch.allclose(r0['pred'].double(), pred.detach().cpu().double(), rtol=1e-4, atol=1e-5)
    ref = run.model.named_reference_tensors(grad=True)
    for k, g in ref.items():
        if g is None:
            continue
        g = g.detach().cpu().double()
        for r in (r0, r1):
            assert r['grads'][k] is not None, k
            assert float((r['grads'][k].double() - g).abs().max()) <= 2e-4 * float(g.abs().max()) + 1e-6, k
    for k in r0['params']:
        assert torch.equal(r0['params'][k], r1['params'][k]), k
    assert torch.equal(r0['pred_all'], r1['pred_all']) and r0['pred_all'].shape == (n,)


def _wide_shard_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['KGW_SHARD_GENE_LAYER'] = '1'
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from kgwas_amd.shard import ShardedTrainer
        from tests.test_gpu_dist import _wide_run
        data, run, ids = _wide_run()
        data.train_input_nodes = ('SNP', ids)
        run.model.train()
        st = ShardedTrainer(run, ('SNP', ids[:512 * 2]), 512, lr=1e-3, weight_decay=5e-4)
        assert st.gene_shard is not None and st.gene_shard.inline
        st.forward_backward(0)
        st.allreduce_grads()
        grads = {k: (None if v is None else v.detach().cpu().clone()) for k, v in run.model.named_reference_tensors(grad=True).items()}
        st.opt.step()
        st.step(1)
        params = {k: v.detach().cpu() for k, v in run.model.named_reference_tensors().items()}
        coll = st.collectives()
        # the same two steps in the CAPTURED form (staged gene-layer shard: stages at the trainer's level, between graph segments)
        data2, run2, _ = _wide_run()
        run2.model.train()
        st2 = ShardedTrainer(run2, ('SNP', ids[:512 * 2]), 512, lr=1e-3, weight_decay=5e-4, use_graph=True)
        assert st2.use_graph and st2.gene_shard is not None and not st2.gene_shard.inline
        st2.step(0); st2.step(1); st2.check()
        params_g = {k: v.detach().cpu() for k, v in run2.model.named_reference_tensors().items()}
        torch.save({'grads': grads, 'params': params, 'params_graph': params_g, 'collectives': coll,
                    'collectives_graph': st2.collectives()}, os.path.join(out_dir, f'ws{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_sharded_mode_with_the_gene_layer_split_by_rows_equals_single_process(tmp_path):
    """SNP-sharded mode on the benchmark-shaped case (resident first gene layer on kgw_gemm3) with that layer ALSO split by gene
    rows over the ranks (inline all-gather / reduce-scatter inside the autograd node): all-reduced gradients of batch 0 equal the
    single-process gradients, the ranks stay bit-identical through two steps."""
    from tests.test_gpu_dist import _wide_run
    world = 2
    mp.start_processes(_wide_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    r0, r1 = (torch.load(os.path.join(tmp_path, f'ws{r}.pt'), weights_only=False) for r in range(world))
    for k in r0['params']:
        assert torch.equal(r0['params'][k], r1['params'][k]), k
        assert torch.equal(r0['params_graph'][k], r1['params_graph'][k]), k
        d = float((r0['params_graph'][k].double() - r0['params'][k].double()).abs().max())
        assert d <= 5e-5 * max(float(r0['params'][k].double().abs().max()), 1e-6) + 3e-6, (k, d)      # captured == uncaptured
    assert any('gene layer' in k for k in r0['collectives']) and any('gene layer' in k for k in r0['collectives_graph'])
    from kgwas_amd.sampler import NeighborLoader
    data, run, ids = _wide_run()
    run.model.train()
    batch = next(iter(NeighborLoader(data.data, [-1, -1], ('SNP', ids[:512]), batch_size=512, device='cuda:0')))
    loss, _ = run.model.forward_loss(batch.x_dict, batch.edge_index_dict, 512, batch.n_id('SNP'), batch.dg.y['SNP'], run._ld_weight_vector())
    loss.backward()
    for k, g in run.model.named_reference_tensors(grad=True).items():
        if g is None:
            continue
        g = g.detach().cpu().double()
        err = float((r0['grads'][k].double() - g).abs().max())
        assert err <= 2e-4 * float(g.abs().max()) + 1e-6, (k, err)


def _graph_worker(rank, world, port, out_dir, which):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from kgwas_amd import dist as kdist
        from kgwas_amd.shard import ShardedTrainer
        out = {}
        for use_graph in (False, True):
            data, run = _make_run(which)
            kdist.broadcast_params(run.model)
            run.model.train()
            ids = _ids(data)[:BS * 4]
            if which == 'edge':                                   # a batch whose seeds all lie in rank 0's range
                ids = np.concatenate([ids[:BS], np.sort(ids)[:BS], ids[BS:2 * BS]])
            st = ShardedTrainer(run, ('SNP', ids), BS, lr=1e-3, weight_decay=5e-4, use_graph=use_graph)
            assert st.use_graph == use_graph
            losses = []
            for i in range(st.n_batches):
                st.step(i)
                losses.append(float(st.last_loss))
            st.check() if use_graph else None
            out[use_graph] = {'params': {k: v.detach().cpu() for k, v in run.model.named_reference_tensors().items()},
                              'losses': losses, 'pred': st.predict(ids[:BS + 7]).cpu(),
                              'segments': (len(st.comp_seg[0].items) + len(st.samp_seg[0].items)) if use_graph else 0, 'collectives': st.collectives()}
        torch.save(out, os.path.join(out_dir, f'g{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('which', ['small', 'edge'])
def test_captured_sharded_step_equals_the_uncaptured_one(tmp_path, which):
    """ShardedTrainer(use_graph=True): static layout (a rank's share of a batch padded with zero-weight pad nodes), the staged
    exchange, the step replayed from graph segments with the collectives between them -- against the uncaptured step on the
    same batches: per-rank loss shares, parameters after every batch's Adam step, sharded inference.  'edge': one batch has no
    seed in rank 1's range (all of its seeds are pads there)."""
    world = 2
    mp.start_processes(_graph_worker, args=(world, _free_port(), str(tmp_path), which), nprocs=world, join=True, start_method='spawn')
    recs = [torch.load(os.path.join(tmp_path, f'g{r}.pt'), weights_only=False) for r in range(world)]
    for r in recs:
        e, g = r[False], r[True]
        assert g['segments'] >= 7                                  # graphs and collectives alternate
        for a, b in zip(e['losses'], g['losses']):
            assert abs(a - b) <= 1e-5 * abs(a) + 1e-7, (e['losses'], g['losses'])
        for k in e['params']:
            d = float((e['params'][k].double() - g['params'][k].double()).abs().max())
            scale = float(e['params'][k].double().abs().max())
            assert d <= 5e-5 * max(scale, 1e-6) + 3e-6, (k, d, scale)
        assert torch.allclose(e['pred'], g['pred'], rtol=1e-4, atol=1e-5)
        assert set(e['collectives']) == set(g['collectives'])
    for k in recs[0][True]['params']:
        assert torch.equal(recs[0][True]['params'][k], recs[1][True]['params'][k]), k
