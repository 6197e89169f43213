"""-m gpu: the numerics the small cases cannot reach (VERDICT r3 "missing" 3-5):

* HUB ROWS -- a gene with 75 000 SNP in-edges and one with 10 000: the softmax of kgwas/conv.py:223 over ALL in-edges of a row
  is where fp32 summation order bites (the kernels split such a row into ~590 chunks of 128 edges, merge online-softmax partials
  in a second kernel, and in the backward add 75 000 per-edge terms per column).  Held to the float64 oracle: every attention
  weight of the hub rows, the rows' layer output through the prediction, every parameter gradient.
* the FULL-SIZE benchmark graph (BASELINE.json configs[1]: 784 256 SNPs / 20 032 genes / 20.6 M edges, 5 120-wide gene features)
  for a handful of seeds: the oracle's unpruned pass over their 2-hop subgraph takes seconds; real degree distributions,
  kgw_gemm3 at its real shape.
* the configs[4] feature widths (70 / 57 742 / 128) at a reduced gene count: the zero-padded-K route of kgw_gemm3 at its real
  width against the oracle.

TOLERANCES, stated (fp32 kernels vs a float64 oracle):
  attention weight          |a - a*| <= 1e-4 a* + 1e-9      (relative: a hub row's weights are ~1e-5 each)
  prediction, activations   |x - x*| <= 1e-5 + 1e-4 |x*| + 1e-5 max|x*|
  parameter gradients       |g - g*| <= 1e-4 |g*| + max(1e-5, 1e-4 max|g*|) + 1e-5 max|g*|       (hub case)
  (rtol 1e-4 / atol 1e-5 on activations and gradients "summation-order differences over up-to-1e5-degree rows")
  full-size graph           the same for prediction / loss; parameter gradients  ||g - g*|| <= 1e-3 ||g*||  (norm-wise, every
                            tensor; the MEDIAN tensor within 1e-5; the test prints the largest) and
                            |g - g*| <= 1e-4 |g*| + max(1e-5, 1e-3 max|g*|) element-wise: with 16 seeds a weight gradient is a sum over ~1e5
                            sampled rows of terms of both signs that cancels to a few 1e-2 -- an fp32 sum carries u * sum|terms|,
                            not u * |result| (measured: 22 of 16 384 elements of one 128 x 128 gradient 6e-4 of its maximum off)
"""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle.gat_oracle import weighted_mse
from tests.helpers import assert_close, batch_cpu, grads_by_name, oracle_from_product, params_by_name

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5
HUB, HUB2 = 75_000, 10_000


def make_hub_graph():
    from kgwas_amd.graph import HeteroGraph, add_self_loops, to_undirected
    rng = np.random.default_rng(21)
    n = OrderedDict([('SNP', 90_000), ('Gene', 24), ('CellularComponent', 5), ('BiologicalProcess', 4), ('MolecularFunction', 3)])
    e = OrderedDict()
    hub = np.stack([np.arange(HUB), np.zeros(HUB, dtype=np.int64)])                                    # gene 0: 75 000 in-edges
    few = np.stack([rng.choice(np.arange(HUB, 90_000), 60, replace=False), rng.integers(2, 24, 60)])
    e[('SNP', 'ABC', 'Gene')] = np.concatenate([hub, few], axis=1)
    hub2 = np.stack([np.arange(HUB, HUB + HUB2), np.ones(HUB2, dtype=np.int64)])                       # gene 1: 10 000 in-edges
    rest = np.arange(HUB + HUB2, 90_000)
    e[('SNP', 'TSS', 'Gene')] = np.concatenate([hub2, np.stack([rest, rest % 22 + 2])], axis=1)
    e[('Gene', 'G2G', 'Gene')] = np.array([[0, 1, 2, 3, 0, 5, 7], [1, 2, 0, 4, 6, 0, 1]])
    e[('Gene', 'G-CC', 'CellularComponent')] = np.stack([rng.integers(0, 24, 20), rng.integers(0, 5, 20)])
    e[('Gene', 'G-BP', 'BiologicalProcess')] = np.stack([rng.integers(0, 24, 12), rng.integers(0, 4, 12)])
    e[('Gene', 'G-MF', 'MolecularFunction')] = np.array([[0, 1, 11], [0, 1, 2]])
    g = torch.Generator().manual_seed(5)
    data = HeteroGraph()
    dims = {'SNP': 20, 'Gene': 24}
    for t, k in n.items():
        data[t].x = torch.rand(k, dims.get(t, 16), generator=g)
    und = add_self_loops(to_undirected(e, n), n)
    for et, ei in und.items():
        data[et].edge_index = torch.from_numpy(np.ascontiguousarray(ei))
    data['SNP'].y = torch.rand(n['SNP'], generator=g)
    return data, (dims['SNP'], dims['Gene'], 16)


def _model(data, dims, seed=3):
    from kgwas_amd.model import HeteroGNN
    torch.manual_seed(seed)
    m = HeteroGNN(data, 128, 1, 2, 'GAT', 'sum', dims[0], dims[1], dims[2], 1).cuda()
    with torch.no_grad():
        for pack in list(m.live_packs) + list(m.dead_packs):
            pack.bias.normal_(0, 0.1)
            # attention vectors and MLP output large enough that the 75 000 logits of a hub row are far from uniform: the row
            # maximum, the exponent range and the rescaling of the online softmax's running sums all get exercised
            pack.att_src.mul_(12.0)
            pack.att_dst.mul_(12.0)
        m.snp_feat_mlp.FC_output.weight.mul_(4.0)
    return m


def _oracle_layer1_attention(oracle, x, ei):
    xd = dict(x)
    xd['SNP'] = oracle.snp_feat_mlp(xd['SNP'])
    xd['Gene'] = oracle.gene_feat_mlp(xd['Gene'])
    for t in ('CellularComponent', 'BiologicalProcess', 'MolecularFunction'):
        if t in xd:
            xd[t] = oracle.go_feat_mlp(xd[t])
    _, att = oracle.convs[0](xd, ei, return_attention_weights=True)
    return att


def test_hub_rows_match_the_float64_oracle():
    from kgwas_amd.sampler import NeighborLoader
    data, dims = make_hub_graph()
    model = _model(data, dims)
    # seeds: SNPs under the 75 000-hub, under the 10 000-hub and elsewhere => both hubs are hop-1 rows, all their SNPs hop 2
    seeds = np.array([5, 40_000, 74_999, HUB + 3, HUB + 9_999, 86_000, 88_500, 89_999] + list(range(100, 124)))
    bs = len(seeds)
    batch = next(iter(NeighborLoader(data, [-1, -1], ('SNP', seeds), batch_size=bs, device='cuda:0')))
    nid_g = batch.n_id('Gene').cpu().numpy()
    assert 0 in nid_g and 1 in nid_g and batch.n_nodes['SNP'] >= HUB + HUB2
    model.train()
    out = model(batch.x_dict, batch.edge_index_dict, bs)
    y = torch.rand(bs, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    w = torch.rand(bs, dtype=torch.float64, generator=torch.Generator().manual_seed(2)) + 0.5
    loss = weighted_mse(out, y.cuda(), w.cuda())
    loss.backward()

    oracle = oracle_from_product(model)
    x, ei = batch_cpu(batch)
    out_o = oracle(x, ei, bs)
    loss_o = weighted_mse(out_o, y, w)
    loss_o.backward()
    assert_close(out, out_o.detach(), RTOL, ATOL, 'pred')
    assert_close(loss.detach(), loss_o.detach(), RTOL, ATOL, 'loss')
    go = grads_by_name(oracle)
    n_live = 0
    for name, g in grads_by_name(model).items():
        ref = go[name]
        if g is None:
            assert ref is None or float(ref.abs().max()) == 0.0, name
            continue
        n_live += 1
        assert_close(g, ref, RTOL, max(ATOL, 1e-4 * float(ref.abs().max())), f'grad {name}')
    assert n_live > 10

    # ---- every attention weight of the two hub rows (layer 1) ----------------------------------------------------------------
    with torch.no_grad():
        att = model.hot_path_attention(batch)[0].cpu().numpy().astype(np.float64)
        att_o = _oracle_layer1_attention(oracle, x, ei)
    m, sc = batch.meta, batch.dg.schema
    seg_ptr = batch.buf.seg_ptr.cpu().numpy()
    col = batch.buf.col_local.cpu().numpy()
    nid_s = batch.n_id('SNP').cpu().numpy()
    checked = 0
    for rel, gene, deg in (('ABC', 0, HUB), ('TSS', 1, HUB2)):
        et = ('SNP', rel, 'Gene')
        r = sc.edge_types.index(et)
        e_o = ei[et].numpy()
        a_o = att_o[et].detach().reshape(-1).numpy()
        # oracle: (global SNP) -> alpha on the edges into this gene
        dst_local_o = int(np.nonzero(nid_g == gene)[0][0])
        sel = e_o[1] == dst_local_o
        assert int(sel.sum()) == deg
        want = dict(zip(nid_s[e_o[0][sel]].tolist(), a_o[sel].tolist()))
        # product: the gene is a hop-1 destination row; its segment in relation r
        d_i = sc.type_id['Gene']
        found = False
        for hop in range(2):
            a, b = int(m.seg_off[hop][r]), int(m.seg_off[hop][r + 1])
            row0 = int(m.node_off[d_i][hop])
            for sg in range(a, b):
                if nid_g[row0 + (sg - a)] != gene:
                    continue
                e0, e1 = int(seg_ptr[sg]), int(seg_ptr[sg + 1])
                assert e1 - e0 == deg
                got = att[e0:e1]
                ref = np.array([want[int(s)] for s in nid_s[col[e0:e1]]])
                err = np.abs(got - ref)
                assert np.all(err <= 1e-4 * ref + 1e-9), (rel, float((err / ref).max()))
                assert abs(got.sum() - 1.0) < 1e-5
                # the logits are not degenerate: the weights of one row spread over more than a factor of two
                assert ref.max() / ref.min() > 2.0, (rel, ref.max() / ref.min())
                checked += deg
                found = True
        assert found, rel
    assert checked == HUB + HUB2


@pytest.fixture(scope='module')
def full_c1():
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.kgwas_data import KGWAS_Data
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_synth_full_test_c1_fast_causal')
    run = KGWAS(data, device='cuda:0', seed=1)
    run.initialize_model()
    yield run
    data.data._extra.pop('_device_graphs', None)
    del run, data
    torch.cuda.empty_cache()


def _compare_with_oracle(run, ids, what, min_genes):
    from kgwas_amd import ops
    from kgwas_amd.sampler import NeighborLoader
    bs = len(ids)
    model = run.model
    with torch.no_grad():
        for pack in list(model.live_packs) + list(model.dead_packs):
            pack.bias.normal_(0, 0.1)
        model.lin.bias.fill_(0.5)          # (keep the read-out's ReLU of model.py:86 alive: a dead one zeroes every gradient)
    batch = next(iter(NeighborLoader(run.data.data, [-1, -1], ('SNP', np.asarray(ids)), batch_size=bs, device='cuda:0')))
    assert batch.n_nodes['Gene'] >= min_genes, batch.n_nodes
    ld_w = run._ld_weight_vector()
    model.train()
    for p in model.parameters():
        p.grad = None
    g3 = ops.ROUTES.get('kgw_gemm3', 0)
    lib0 = ops.LIBRARY_GEMM.calls
    loss, pred = model.forward_loss(batch.x_dict, batch.edge_index_dict, bs, batch.n_id('SNP'), batch.dg.y['SNP'], ld_w)
    loss.backward()
    torch.cuda.synchronize()
    assert ops.ROUTES.get('kgw_gemm3', 0) - g3 == 2, 'the wide gene layer (forward + weight gradient) must run on kgw_gemm3'
    assert ops.LIBRARY_GEMM.calls == lib0, 'no library GEMM at this size'
    oracle = oracle_from_product(model)
    x, ei = batch_cpu(batch)
    out_o = oracle(x, ei, bs)
    s = torch.as_tensor(np.asarray(ids))
    loss_o = weighted_mse(out_o, run.data.data['SNP'].y[s].double(), ld_w[s.cuda()].cpu())
    loss_o.backward()
    assert_close(pred, out_o.detach().reshape(-1), RTOL, ATOL, f'{what}: pred')
    assert abs(float(loss) - float(loss_o)) <= 1e-4 * abs(float(loss_o)) + 1e-7
    go = grads_by_name(oracle)
    # The yardstick for gradients that are SMALL DIFFERENCES OF LARGE TERMS: the same oracle in float32 (the reference's own
    # arithmetic, kgwas/conv.py on PyG in fp32).  d a_dst of a destination row is zero in exact arithmetic wherever the row's
    # logits sit on one branch of the leaky ReLU (a_dst shifts all of them alike and the softmax of conv.py:223 does not see a common
    # shift), so everything that hangs on it alone -- lin_dst / att_dst of the relations into the seed SNPs, and through the seeds'
    # layer-1 rows the lin_src of the same relations one layer down -- is cancellation residue: the float32 oracle is ~1e-3 off
    # float64 there itself, and which 1e-3 depends on the order of every sum upstream.
    oracle32 = oracle_from_product(model, dtype=torch.float32)
    x32, ei32 = batch_cpu(batch, dtype=torch.float32)
    loss32 = weighted_mse(oracle32(x32, ei32, bs), run.data.data['SNP'].y[s].float(), ld_w[s.cuda()].cpu())
    loss32.backward()
    go32 = grads_by_name(oracle32)
    n, nw = 0, []
    for name, g in grads_by_name(model).items():
        ref = go[name]
        if g is None:
            assert ref is None or float(ref.abs().max()) == 0.0, name
            continue
        assert_close(g, ref, RTOL, max(ATOL, 1e-3 * float(ref.abs().max())), f'{what}: grad {name}')
        # (a gradient that is zero in exact arithmetic -- d att_dst of a softmax over ONE destination's logits: adding a constant
        #  to them changes nothing -- comes out as 1e-19 in float64 and 1e-10 in fp32: only the absolute bound applies to it)
        if float(ref.norm()) > 1e-6:
            e32 = float((go32[name].double() - ref).norm() / ref.norm()) if go32.get(name) is not None else 0.0
            nw.append((float((g.double() - ref).norm() / ref.norm()), name, e32))
        n += 1
    assert n > 40
    nw.sort(reverse=True)
    print(f'[{what}] norm-wise gradient errors, largest first (this path / the float32 oracle): ' +
          ', '.join(f'{k} {e:.1e} / {e32:.1e}' for e, k, e32 in nw[:4]) + f'; median {nw[len(nw) // 2][0]:.1e} over {len(nw)} tensors')
    # every tensor within 1e-3 of float64 norm-wise, or -- the cancellation residues above -- within 2 x of what the reference
    # arithmetic itself is off (no other cap).  Round 4 measured this path at 1 - 7 x the float32 oracle's own error on these tensors
    # (8e-7 ... 2.6e-3 depending on the batch) and held it to 10 x, at most 2e-2: k_agg_bwd_dst took sum_j alpha_j d alpha_j of a row
    # from the stored fp32 aggregate, <dZ_i, z_i> -- one more rounding that does not cancel in d a_dst.  Round 5: the row sum d a_dst is
    # formed consistently with the edges' own d alpha (kgw_aggregate.hip, k_agg_bwd_dst: sum of a_e s_e (dalpha_e - c*) with
    # c* = sum a_e dalpha_e / sum a_e, in small differences), as autograd of the reference's softmax (conv.py:223) does.
    worst = max((e / max(e32, 1e-30), name, e, e32) for e, name, e32 in nw if e > 1e-3) if any(e > 1e-3 for e, _, _ in nw) else None
    print(f'[{what}] worst tensor above 1e-3 relative to the float32 oracle: {worst}')
    for e, name, e32 in nw:
        assert e <= max(1e-3, 2.0 * e32), (name, e, e32)
    assert nw[len(nw) // 2][0] <= 1e-5, nw[len(nw) // 2]
    gw, rw = grads_by_name(model)['gene_feat_mlp.FC_hidden.weight'].double(), go['gene_feat_mlp.FC_hidden.weight']
    assert float(rw.norm()) > 0 and int((pred > 0).sum()) >= bs // 2, 'a dead read-out would make this comparison empty'
    assert float((gw - rw).norm() / rw.norm()) < 1e-5, 'the wide layer relative to its own magnitude'
    k = min(8, bs)
    assert torch.equal(torch.topk(pred.cpu().double(), k).indices, torch.topk(out_o.detach().reshape(-1), k).indices)
    return batch


@pytest.mark.parametrize('first', [0, 300_000])
def test_full_size_graph_against_the_oracle_for_a_few_seeds(full_c1, first):
    """configs[1] at full size, 16 seeds of the reference's batch order: the unpruned float64 oracle on the product's own 2-hop
    subgraph (~all 20 032 genes as layer-1 sources, 5 120-wide features, real hub degrees) -- prediction, loss, every gradient."""
    ids = np.asarray(full_c1.data.train_input_nodes[1])
    ids = ids[first:first + 16]
    batch = _compare_with_oracle(full_c1, ids, f'full-size seeds {first}..', min_genes=10_017)
    # the degree distribution is the real one: some sampled destination row has thousands of in-edges
    sp = batch.buf.seg_ptr[:int(batch.meta.seg_end[batch.dg.n_hops - 1]) + 1].cpu().numpy()
    assert int(np.diff(sp).max()) >= 1000


def test_captured_training_steps_at_full_size_track_the_oracle(full_c1):
    """VERDICT r4 (missing #3): the CAPTURED training step itself -- GraphTrainStep: HIP graph, fused optimiser launch, next batch
    sampled by the side graph -- on full-size configs[1] at the benchmark's batch size, beside the float64 oracle trained from the
    same initial state on the same batches (kgwas/kgwas.py:129-151: forward, LD-weighted loss, backward, Adam with L2): per-step
    loss, the parameter update after the last step, and the ranking of the next batch's seeds."""
    import time
    from kgwas_amd.graph_step import GraphTrainStep
    from oracle.sampler_np import FullNeighborSamplerNP
    from kgwas_amd.kgwas import KGWAS
    run = KGWAS(full_c1.data, device='cuda:0', seed=5)     # (a model of its own: the fixture's has been re-biased by the tests above)
    run.initialize_model()
    model = run.model
    bs, n_steps, lr, wd = 512, 8, 1e-4, 5e-4                 # the reference's training configuration (kgwas/kgwas.py:85-87,116)
    ids = np.asarray(run.data.train_input_nodes[1])[:(n_steps + 1) * bs]
    gs = GraphTrainStep(run, ('SNP', ids), bs, lr=lr, weight_decay=wd)
    assert gs.fused_adam and gs.twin, 'the shipped single-GPU configuration: fused optimiser launch, side sampler'
    oracle = oracle_from_product(model, dtype=torch.float64)
    oracle0 = oracle_from_product(model, dtype=torch.float64)          # (stays at the initial state)
    p0 = params_by_name(model)
    model.train()
    losses = []
    for i in range(n_steps):
        losses.append(float(gs.step(i)))
    gs.check()
    g = run.data.data
    smp = FullNeighborSamplerNP(g.edge_index_dict, g.num_nodes_dict, 2)
    opt = torch.optim.Adam(oracle.parameters(), lr=lr, weight_decay=wd)
    y_all = g['SNP'].y.double()
    w_all = run._ld_weight_vector().cpu()
    t0 = time.time()
    losses_o = []
    for i in range(n_steps):
        n_id, ei = smp.sample('SNP', ids[i * bs:(i + 1) * bs])
        x = {t: g[t].x[n_id[t]].double() for t in g.node_types}
        opt.zero_grad()
        lo = weighted_mse(oracle(x, ei, bs), y_all[n_id['SNP'][:bs]], w_all[n_id['SNP'][:bs]])
        lo.backward()
        opt.step()
        losses_o.append(float(lo))
    print(f'[captured trajectory] {n_steps} oracle steps in {time.time() - t0:.0f} s; losses {losses} vs {losses_o}')
    for a, b in zip(losses, losses_o):
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, (losses, losses_o)
    # the parameter UPDATE, norm-wise (Adam turns fp32-noise gradients into +-lr steps: tests/test_gpu_model.py's trajectory bound)
    po = dict(oracle.named_parameters())
    num = den = 0.0
    for n, p in params_by_name(model).items():
        if n not in po:
            continue
        d_hip, d_ref = p - p0[n], po[n].detach() - p0[n]
        num += float((d_hip - d_ref).pow(2).sum()); den += float(d_ref.pow(2).sum())
        assert float((d_hip - d_ref).abs().max()) <= 2.5 * lr * n_steps, n
    assert den > 0 and (num / den) ** 0.5 < 2e-2, f'relative update error {(num / den) ** 0.5:.3e}'
    print(f'[captured trajectory] relative update error after {n_steps} steps: {(num / den) ** 0.5:.3e}')
    # the next batch through both: predictions and the order of the strongest seeds
    from kgwas_amd.sampler import NeighborLoader
    nxt = ids[n_steps * bs:(n_steps + 1) * bs]
    batch = next(iter(NeighborLoader(g, [-1, -1], ('SNP', nxt), batch_size=bs, device='cuda:0')))
    with torch.no_grad():
        pred = model(batch.x_dict, batch.edge_index_dict, bs).reshape(-1)
        n_id, ei = smp.sample('SNP', nxt)
        x = {t: g[t].x[n_id[t]].double() for t in g.node_types}
        pred_o = oracle(x, ei, bs).reshape(-1)
        pred_0 = oracle0(x, ei, bs).reshape(-1)
    # Adam turns gradient coordinates that are rounding noise into +-lr steps, differently in fp32 and fp64 (the 2e-2 of the update
    # above): predictions are held to a twentieth of what the eight steps moved them, on top of the forward tolerance
    moved = float((pred_o - pred_0).abs().max())
    err = float((pred.cpu().double() - pred_o).abs().max())
    print(f'[captured trajectory] next batch: the steps moved the predictions by up to {moved:.3e}, this path differs from the oracle by up to {err:.3e}')
    assert moved > 1e-3, 'the steps must have changed the predictions for this comparison to mean anything'
    assert_close(pred, pred_o, 1e-3, 1e-4 + 0.05 * moved, 'prediction after the captured steps')
    assert torch.equal(torch.topk(pred.cpu().double(), 8).indices, torch.topk(pred_o, 8).indices)


@pytest.mark.skipif(not int(__import__('os').environ.get('KGW_TRAJECTORY_STEPS', '0')), reason='artifact run: KGW_TRAJECTORY_STEPS=100')
def test_long_captured_trajectory_beside_the_oracle_as_an_artifact(full_c1):
    """VERDICT r5 item 7 (kgwas/kgwas.py:129-173): KGW_TRAJECTORY_STEPS captured steps of full-size configs[1] at batch 512 beside the
    float64 oracle from the same initial state -- per-step loss, the parameter-update error every 10 steps, and validation MSE /
    Pearson on a fixed subset of the validation SNPs before and after -- written to KGW_TRAJECTORY_OUT.  A hundred Adam steps are
    not eight: Adam normalises every coordinate by its own second moment, so the coordinates whose gradient is a cancellation
    residue (lin_dst / att_dst of the relations into the seeds) take +-lr steps whose SIGN is rounding noise -- in any
    arithmetic.  The yardstick is therefore the reference's own arithmetic: the SAME oracle run in float32 (what PyG computes in)
    beside the float64 one; this path is held to a small multiple of how far that run drifts.  Opt-in (~4 s of CPU per step)."""
    import os
    import time
    from scipy.stats import pearsonr
    from kgwas_amd.graph_step import GraphTrainStep
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.sampler import NeighborLoader
    from oracle.sampler_np import FullNeighborSamplerNP
    n_steps = int(os.environ['KGW_TRAJECTORY_STEPS'])
    n_val = int(os.environ.get('KGW_TRAJECTORY_VAL', '5120'))
    out_path = os.environ.get('KGW_TRAJECTORY_OUT', 'gpurun_out/trajectory.txt')
    run = KGWAS(full_c1.data, device='cuda:0', seed=5)
    run.initialize_model()
    model = run.model
    bs, lr, wd = 512, 1e-4, 5e-4
    ids = np.asarray(run.data.train_input_nodes[1])[:n_steps * bs]
    val_ids = np.asarray(run.data.val_input_nodes[1])[:n_val]
    g = run.data.data
    smp = FullNeighborSamplerNP(g.edge_index_dict, g.num_nodes_dict, 2)
    y_all = g['SNP'].y.double()
    w_all = run._ld_weight_vector().cpu()
    oracle = oracle_from_product(model, dtype=torch.float64)
    oracle32 = oracle_from_product(model, dtype=torch.float32)
    p0 = params_by_name(model)
    lines = [f'{n_steps} captured training steps (GraphTrainStep: HIP graph, fused optimiser launch, side sampler) of full-size configs[1] '
             f'(batch {bs}, Adam lr {lr}, weight decay {wd}: kgwas/kgwas.py:85-87,116) beside the float64 oracle from the same initial state;',
             'yardstick: the same oracle in float32 (the reference\'s arithmetic) beside the float64 one.', '']

    def val_hip():
        model.eval()
        preds = []
        with torch.no_grad():
            for b in NeighborLoader(g, [-1, -1], ('SNP', val_ids), batch_size=bs, device='cuda:0'):
                preds.append(model(b.x_dict, b.edge_index_dict, b.batch_size).reshape(-1).double().cpu())
        model.train()
        return torch.cat(preds)

    def val_oracle(orc, dt):
        preds = []
        with torch.no_grad():
            for k in range(0, len(val_ids), bs):
                seeds = val_ids[k:k + bs]
                n_id, ei = smp.sample('SNP', seeds)
                preds.append(orc({t: g[t].x[n_id[t]].to(dt) for t in g.node_types}, ei, len(seeds)).reshape(-1).double())
        return torch.cat(preds)

    def metrics(pred):
        truth = y_all[torch.from_numpy(val_ids)]
        return float(((pred - truth) ** 2).mean()), float(pearsonr(pred.numpy(), truth.numpy())[0])

    def val_line(tag):
        ph, po_, p32 = val_hip(), val_oracle(oracle, torch.float64), val_oracle(oracle32, torch.float32)
        (mh, rh), (mo, ro), (m3, r3) = metrics(ph), metrics(po_), metrics(p32)
        lines.append(f'validation subset ({len(val_ids)} SNPs) {tag}:')
        lines.append(f'    oracle float64   MSE {mo:.6f}  Pearson {ro:+.6f}')
        lines.append(f'    HIP path         MSE {mh:.6f}  Pearson {rh:+.6f}   |d MSE| {abs(mh - mo):.2e}  |d Pearson| {abs(rh - ro):.2e}  max |d pred| {float((ph - po_).abs().max()):.3e}')
        lines.append(f'    oracle float32   MSE {m3:.6f}  Pearson {r3:+.6f}   |d MSE| {abs(m3 - mo):.2e}  |d Pearson| {abs(r3 - ro):.2e}  max |d pred| {float((p32 - po_).abs().max()):.3e}')
        return ((abs(rh - ro), abs(mh - mo) / max(mo, 1e-12), float((ph - po_).abs().max())),
                (abs(r3 - ro), abs(m3 - mo) / max(mo, 1e-12), float((p32 - po_).abs().max())))
    val_line('before training')
    gs = GraphTrainStep(run, ('SNP', ids), bs, lr=lr, weight_decay=wd)
    model.train()
    losses, snaps = [], {}
    for i in range(n_steps):
        losses.append(gs.step(i).detach().clone())
        if (i + 1) % 10 == 0 or i + 1 == n_steps:
            torch.cuda.synchronize()
            snaps[i + 1] = {n: p.detach().clone().cpu().double() for n, p in params_by_name(model).items()}
    gs.check()
    losses = [float(l) for l in losses]
    p0 = {n: p.cpu().double() for n, p in p0.items()}
    opt = torch.optim.Adam(oracle.parameters(), lr=lr, weight_decay=wd)
    opt32 = torch.optim.Adam(oracle32.parameters(), lr=lr, weight_decay=wd)
    po, po32 = dict(oracle.named_parameters()), dict(oracle32.named_parameters())

    def upd_err(cur):
        num = den = 0.0
        for n, p in cur.items():
            if n in po:
                d_a, d_ref = p - p0[n], po[n].detach() - p0[n]
                num += float((d_a - d_ref).pow(2).sum()); den += float(d_ref.pow(2).sum())
        return (num / max(den, 1e-300)) ** 0.5
    t0 = time.time()
    lines.append('')
    lines.append('step   loss: oracle float64   HIP path (rel. diff)          oracle float32 (rel. diff)       [relative parameter-update error vs float64, norm-wise: HIP | float32 oracle]')
    worst = {'hip_loss': 0.0, 'f32_loss': 0.0, 'hip_upd': 0.0, 'f32_upd': 0.0}
    for i in range(n_steps):
        n_id, ei = smp.sample('SNP', ids[i * bs:(i + 1) * bs])
        sel = n_id['SNP'][:bs]
        x = {t: g[t].x[n_id[t]].double() for t in g.node_types}
        opt.zero_grad()
        lo = weighted_mse(oracle(x, ei, bs), y_all[sel], w_all[sel])
        lo.backward()
        opt.step()
        opt32.zero_grad()
        l3 = weighted_mse(oracle32({t: v.float() for t, v in x.items()}, ei, bs), y_all[sel], w_all[sel])       # (float64 loss on float32 predictions: kgwas.py:139-145)
        l3.backward()
        opt32.step()
        rh = abs(losses[i] - float(lo)) / max(abs(float(lo)), 1e-12)
        r3 = abs(float(l3) - float(lo)) / max(abs(float(lo)), 1e-12)
        worst['hip_loss'], worst['f32_loss'] = max(worst['hip_loss'], rh), max(worst['f32_loss'], r3)
        extra = ''
        if i + 1 in snaps:
            uh = upd_err(snaps[i + 1])
            u3 = upd_err({n: p.detach().double() for n, p in po32.items()})
            worst['hip_upd'], worst['f32_upd'] = max(worst['hip_upd'], uh), max(worst['f32_upd'], u3)
            extra = f'   update error {uh:.3e} | {u3:.3e}'
        lines.append(f'{i + 1:4d}   {float(lo):.10f}   {losses[i]:.10f} ({rh:.2e})   {float(l3):.10f} ({r3:.2e}){extra}')
    lines.append('')
    lines.append(f'({n_steps} steps of both oracles: {time.time() - t0:.0f} s of CPU)')
    lines.append(f'worst per-step loss difference from float64: HIP path {worst["hip_loss"]:.2e}, float32 oracle {worst["f32_loss"]:.2e} (relative)')
    lines.append(f'worst parameter-update error vs float64:     HIP path {worst["hip_upd"]:.3e}, float32 oracle {worst["f32_upd"]:.3e}')
    (dr_h, dm_h, dp_h), (dr_3, dm_3, dp_3) = val_line(f'after {n_steps} steps')
    lines.append(f'(Pearson on {len(val_ids)} SNPs has a sampling error of 1/sqrt(n) = {len(val_ids) ** -0.5:.1e}: on labels it correlates with at '
                 f'~0.00 the statistic cannot resolve the models\' drift; the predictions themselves can -- max |d pred| above)')
    os.makedirs(os.path.dirname(out_path) or '.', exist_ok=True)
    with open(out_path, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    print('\n'.join(lines[-12:]))
    # this path drifts from float64 no further than a small multiple of what the reference's own arithmetic does
    assert worst['hip_loss'] <= max(2e-4, 3.0 * worst['f32_loss']), worst
    assert worst['hip_upd'] <= max(2e-2, 3.0 * worst['f32_upd']), worst
    # predictions: no further from float64's than three times the float32 oracle's; the validation statistics within their own
    # resolution (measured, round 6: max |d pred| 0.19 against 0.12; |d Pearson| 5.5e-3 at values of -0.002 / +0.004; MSE 2.6e-3)
    assert dp_h <= max(1e-3, 3.0 * dp_3), (dp_h, dp_3)
    assert dr_h <= max(1e-3, len(val_ids) ** -0.5) and dm_h <= max(1e-3, 10.0 * dm_3), ((dr_h, dm_h), (dr_3, dm_3))


def test_full_mode_widths_against_the_oracle_at_a_reduced_gene_count():
    """configs[4] widths 70 / 57 742 / 128 (kgwas_data.py:167,244) on a quarter-scale graph (5 008 genes: the 57 742-wide product
    through kgw_gemm3's zero-padded-K route, 57 760 = 1 805 x 32) against the float64 oracle."""
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.kgwas_data import KGWAS_Data
    data = KGWAS_Data.from_synthetic(scale=0.25, seed=1, mode='full', gwas_kind='causal', data_path='/tmp/kgwas_synth_quarter_full')
    assert (data.snp_init_dim_size, data.gene_init_dim_size, data.go_init_dim_size) == (70, 57742, 128)
    run = KGWAS(data, device='cuda:0', seed=1)
    run.initialize_model()
    ids = np.asarray(data.train_input_nodes[1])[:64]
    _compare_with_oracle(run, ids, 'full-mode widths', min_genes=2_505)
    data.data._extra.pop('_device_graphs', None)
