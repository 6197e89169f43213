"""-m gpu: the path at BASELINE.json's FULL size (configs[1]: SynthKG-fast, 784 256 SNPs / 20 032 genes / ~20.6 M
directed edges, 512-seed batches), where the CPU oracle would take minutes per batch: checked through
size-independent properties of the domain instead --

* full-neighbour sampling: every sampled destination row carries its WHOLE in-neighbourhood (segment sizes == CSR
  degrees), local ids are a bijection onto the sampled global ids, seeds come first;
* softmax: the attention weights of every non-empty (destination row, relation) segment sum to 1, in both layers;
* minibatch invariance (survey fact 6): a seed's prediction does not depend on which other seeds share its batch;
* linearity / determinism of the backward pass: grad(2 * loss) == 2 * grad(loss) bit for bit (power-of-two scale),
  two runs give bitwise identical gradients;
* the captured HIP-graph step reproduces the eager step's losses on the same batches.
"""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu

BS = 512


@pytest.fixture(scope='module')
def full_run():
    from kgwas_amd.kgwas import KGWAS
    from kgwas_amd.kgwas_data import KGWAS_Data
    data = KGWAS_Data.from_synthetic(scale=1.0, seed=1, mode='fast', gwas_kind='causal', data_path='/tmp/kgwas_synth_full_test')
    run = KGWAS(data, device='cuda:0', seed=1)
    run.initialize_model()
    n = {t: int(x.shape[0]) for t, x in ((t, data.data[t].x) for t in data.data.node_types)}
    assert n['SNP'] == 784256 and n['Gene'] == 20032
    return run


def _loader(run, ids, bs=BS):
    from kgwas_amd.sampler import NeighborLoader
    return NeighborLoader(run.data.data, [-1, -1], ('SNP', np.asarray(ids)), batch_size=bs, drop_last=True, device='cuda:0')


def test_full_neighbourhoods_and_relabelling(full_run):
    run = full_run
    ids = np.asarray(run.data.train_input_nodes[1][:BS])
    batch = next(iter(_loader(run, ids)))
    dg = batch.dg
    sc = dg.schema
    g = run.data.data
    # local <-> global ids: unique per type, seeds first and in order
    nid = {t: batch.n_id(t).cpu().numpy() for t in sc.node_types}
    for t, v in nid.items():
        assert len(np.unique(v)) == len(v), t
    assert np.array_equal(nid['SNP'][:BS], ids)
    ei = batch.edge_index_dict
    total = 0
    for r, et in enumerate(sc.edge_types):
        e = ei[et].cpu().numpy()
        total += e.shape[1]
        if e.shape[1] == 0:
            continue
        s_t, d_t = et[0], et[2]
        assert e[0].max() < len(nid[s_t]) and e[1].max() < len(nid[d_t])
        # every destination row that was expanded carries its whole in-neighbourhood
        full_dst = np.asarray(g[et].edge_index[1])
        indeg = np.bincount(full_dst, minlength=len(np.asarray(g[d_t].x)))
        dst_local, cnt = np.unique(e[1], return_counts=True)
        assert np.array_equal(cnt, indeg[nid[d_t][dst_local]]), et
    assert total == batch.n_edges_sampled
    # layer 2 aggregates only into what the read-out needs, layer 1 into what layer 2 reads: pruning never adds edges
    assert batch.n_edges_per_layer[1] <= batch.n_edges_per_layer[0] <= total


def test_attention_weights_of_every_segment_sum_to_one(full_run):
    run = full_run
    ids = np.asarray(run.data.train_input_nodes[1][BS:2 * BS])
    batch = next(iter(_loader(run, ids)))
    run.model.eval()
    attention = run.model.hot_path_attention(batch)
    seg_ptr = batch.buf.seg_ptr.long()
    for l, alpha in enumerate(attention, start=1):
        E = int(alpha.numel())
        assert E == batch.n_edges_per_layer[l - 1] and E > 0
        # layer l aggregates the segments of hops 0 .. L - l (hop-major segment and edge order)
        m, L, NR = batch.meta, batch.dg.num_layers, len(batch.dg.schema.edge_types)
        nseg = int(m.seg_off[L - l][NR])
        sp = seg_ptr[:nseg + 1]
        assert int(sp[0]) == 0 and int(sp[-1]) == E, (l, nseg, int(sp[0]), int(sp[-1]), E)
        seg = torch.bucketize(torch.arange(E, device=alpha.device), sp[1:], right=True)
        sums = torch.zeros(nseg, dtype=torch.float64, device=alpha.device).index_add_(0, seg, alpha.double())
        nonempty = (sp[1:] - sp[:-1]) > 0
        assert bool(nonempty.any())
        err = (sums[nonempty] - 1.0).abs().max().item()
        assert err < 1e-5, (l, err)
        assert float(alpha.min()) >= 0.0 and float(alpha.max()) <= 1.0 + 1e-6


def test_prediction_of_a_seed_does_not_depend_on_its_batch(full_run):
    run = full_run
    tr = np.asarray(run.data.train_input_nodes[1])
    A, B, C = tr[:256], tr[1000:1256], tr[300000:300256]
    run.model.eval()
    preds = []
    with torch.no_grad():
        for other in (B, C):
            batch = next(iter(_loader(run, np.concatenate([A, other]))))
            preds.append(run.model(batch.x_dict, batch.edge_index_dict, BS)[:256].flatten().double().cpu())
    assert_close(preds[0], preds[1], 1e-5, 1e-6, 'seed prediction across batches', rel_to_max=1e-5)
    assert float(preds[0].abs().max()) > 0.0


def test_backward_is_linear_and_bitwise_reproducible(full_run):
    run = full_run
    ids = np.asarray(run.data.train_input_nodes[1][2 * BS:3 * BS])
    ld_w = run._ld_weight_vector()
    run.model.train()

    def grads(scale):
        batch = next(iter(_loader(run, ids)))
        for p in run.model.parameters():
            p.grad = None
        loss, _ = run.model.forward_loss(batch.x_dict, batch.edge_index_dict, BS, batch.n_id('SNP'), batch.dg.y['SNP'], ld_w)
        (loss * scale).backward()
        return float(loss), {n: p.grad.clone() for n, p in run.model.named_parameters() if p.grad is not None}

    l1, g1 = grads(1.0)
    l1b, g1b = grads(1.0)
    l2, g2 = grads(2.0)
    assert l1 == l1b == l2 and np.isfinite(l1) and len(g1) > 20
    for n in g1:
        assert torch.equal(g1[n], g1b[n]), f'{n}: not reproducible'
        assert torch.equal(g2[n], 2.0 * g1[n]), f'{n}: backward is not linear in the loss scale'
        assert bool(torch.isfinite(g1[n]).all())
    assert any(float(g.abs().max()) > 0 for g in g1.values())


def test_graph_step_reproduces_eager_losses_at_full_size(full_run):
    from kgwas_amd.graph_step import GraphTrainStep
    from kgwas_amd.kgwas import KGWAS
    run = full_run
    ids = np.asarray(run.data.train_input_nodes[1][:4 * BS])
    run_g = KGWAS(run.data, device='cuda:0', seed=2)
    run_g.initialize_model()
    run_g.model.load_state_dict(run.model.state_dict())
    run_e = KGWAS(run.data, device='cuda:0', seed=3)
    run_e.initialize_model()
    run_e.model.load_state_dict(run.model.state_dict())
    gs = GraphTrainStep(run_g, ('SNP', ids), BS, lr=1e-4, weight_decay=5e-4)
    opt = torch.optim.Adam(run_e.model.parameters(), lr=1e-4, weight_decay=5e-4)
    ld_w = run_e._ld_weight_vector()
    run_e.model.train()
    it = iter(_loader(run_e, ids))
    for i in range(4):
        lg = float(gs.step(i))
        le = float(run_e.train_step(next(it), opt, ld_w))
        assert np.isfinite(lg) and abs(lg - le) <= 1e-4 * abs(le) + 1e-7, (i, lg, le)
    gs.check()
