"""CPU: the data side of BASELINE.json's configurations -- what changes between them is the label source (null / causal
simulation, sub-sampled cohort, full cohort; kgwas/kgwas_data.py:275-389), the edge thinning of load_kg
(sample_edges / sample_ratio, :261-268) and the feature widths; the graph + model path is the same.  configs[0]
("sample_ratio=0.01 KG + null-simulation GWAS seed=1, reference on CPU, plumbing") runs here end to end on the CPU
restatement; the HIP path runs every configuration in tests/test_gpu_fullsize.py."""
import numpy as np
import pytest
import torch


@pytest.fixture(scope='module')
def thinned_null():
    from kgwas_amd.kgwas_data import KGWAS_Data
    return KGWAS_Data.from_synthetic(scale=1.0, seed=1, gwas_kind='null', sample_edges=True, sample_ratio=0.01,
                                     data_path='/tmp/kgwas_cfg0')


def test_config0_edge_thinning_and_null_labels(thinned_null):
    from kgwas_amd.synth import make_synth_edges
    d = thinned_null
    g = d.data
    full, nc = make_synth_edges(1.0, 1)
    n_gene = nc['Gene']
    for et, ei in full.items():
        s, rel, t = et
        keep = int(ei.shape[1] * 0.01)                                  # kgwas_data.py:263
        if s != t:
            assert g[et].edge_index.shape[1] == keep, et
            assert g[(t, 'rev_' + rel, s)].edge_index.shape[1] == keep    # ToUndirected mirror
        else:
            e = g[et].edge_index.numpy()
            loops = int((e[0] == e[1]).sum())
            assert loops >= n_gene and e.shape[1] <= 2 * keep + n_gene    # symmetrised + coalesced, + N self-loops
    # null simulation: chi-square(1) labels, no signal (kgwas_data.py:275-294 'null')
    assert abs(float(np.mean(d.y)) - 1.0) < 0.02 and d.sample_size == 5000
    assert len(d.train_input_nodes[1]) == 489839 and len(d.val_input_nodes[1]) == 25781 and len(d.test_input_nodes[1]) == 27138
    assert len(d.train_input_nodes[1]) // 512 == 956                    # demo/kgwas_101.ipynb:352


def test_config0_training_steps_on_the_cpu_restatement(thinned_null):
    from oracle.gat_oracle import HeteroGNNOracle, weighted_mse
    from oracle.sampler_np import FullNeighborSamplerNP
    d = thinned_null
    g = d.data
    smp = FullNeighborSamplerNP(g.edge_index_dict, g.num_nodes_dict, 2)
    torch.manual_seed(1)
    model = HeteroGNNOracle(g.edge_types, 128, 1, 2, 'GAT', 'sum', d.snp_init_dim_size, d.gene_init_dim_size, d.go_init_dim_size, 1)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=5e-4)
    w_all = torch.zeros(g['SNP'].x.shape[0], dtype=torch.float64)
    w_all[torch.from_numpy(d.all_ids)] = torch.from_numpy(np.asarray(d.ldsc_weight))
    ids = np.asarray(d.train_input_nodes[1])
    losses, empty_types = [], 0
    for step in range(12):
        seeds = ids[step * 512:(step + 1) * 512]
        n_id, ei = smp.sample('SNP', seeds)
        empty_types += sum(1 for v in n_id.values() if len(v) == 0)
        x = {k: g[k].x[v] for k, v in n_id.items()}
        opt.zero_grad()
        out = model(x, ei, 512)
        s = torch.as_tensor(n_id['SNP'][:512])
        loss = weighted_mse(out, g['SNP'].y[s], w_all[s])
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < 10.0


@pytest.mark.parametrize('kind,n', [('subsample', 10000), ('full_cohort', 387113)])
def test_cohort_size_reaches_the_ld_weights(kind, n):
    """configs[2] / [3]: sample_size drives the LD-score regression weights (kgwas_data.py:399-428); the weights of the
    synthetic cohort equal the reference helper's formula at that N (pinned by tests/golden/ref_helpers.npz elsewhere)."""
    from kgwas_amd.kgwas_data import KGWAS_Data
    from kgwas_amd.utils import ldsc_regression_weights
    d = KGWAS_Data.from_synthetic(scale=0.01, seed=1, gwas_kind=kind, feat_dims={'Gene': 32}, data_path=f'/tmp/kgwas_cfg_{kind}')
    assert d.sample_size == n and float(np.mean(d.lr_uni.N)) == float(n)
    ld, wld = d._synth_ld
    w = ldsc_regression_weights(ld, wld, float(n), 15000000, 0.5)
    assert np.allclose(d.ldsc_weight, w / w.mean(), rtol=1e-12)
    assert np.isfinite(d.y).all() and float(np.mean(d.y)) > 1.0           # causal architecture: inflated chi-square


def test_residual_labels_follow_the_reference_definition():
    """kgwas/kgwas_data.py:448-500: chi-square of BETA / SE (NaN -> 0) minus a straight-line fit on an LD score -- weighted by the
    LDSC weights for 'residual-w-ld' / 'residual-ld', plain least squares for the '-ols' labels; the 'residual-ld*' labels fit on
    ld_score and predict with w_ld_score, as the reference does.  Checked against an independent weighted least-squares solve."""
    import pandas as pd
    from kgwas_amd.kgwas_data import KGWAS_Data
    kg = KGWAS_Data.from_synthetic(scale=0.002, seed=2, data_path='/tmp/kgwas_synth_resid', split=False)
    rng = np.random.default_rng(0)
    n = len(kg.lr_uni)
    base = kg.lr_uni.drop(columns=['chi'])
    base['BETA'] = rng.standard_normal(n)
    base['SE'] = rng.uniform(0.5, 2.0, n)
    base.loc[base.index[:3], 'SE'] = np.nan                      # NaN labels become 0 before the fit (kgwas_data.py:450)
    for label in ('residual-w-ld', 'residual-ld', 'residual-ld-ols', 'residual-ld-ols-abs'):
        kg.lr_uni = base.copy()
        kg.process_gwas_file(label=label)
        lr = kg.lr_uni
        y0 = np.nan_to_num((base['BETA'] / base['SE']).values ** 2, nan=0.0)
        fit_x = (lr.w_ld_score if label == 'residual-w-ld' else lr.ld_score).values
        w = np.asarray(kg.ldsc_weight) if label in ('residual-w-ld', 'residual-ld') else np.ones(n)
        X = np.stack([np.ones(n), fit_x], 1) * np.sqrt(w)[:, None]
        a, b = np.linalg.lstsq(X, y0 * np.sqrt(w), rcond=None)[0]
        want = y0 - (a + b * lr.w_ld_score.values)
        if label.endswith('abs'):
            want = np.abs(want)
        assert np.allclose(kg.y, want, rtol=1e-9, atol=1e-9), label
        assert np.isfinite(kg.y).all()
    with pytest.raises(NotImplementedError):
        kg.lr_uni = base.copy()
        kg.process_gwas_file(label='no-such-label')
