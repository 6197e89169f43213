"""``KGWAS_Data`` -- the reference's graph + label container (kgwas/kgwas_data.py:19-558), API kept.

In scope here: the graph-build semantics the hot path consumes -- COO lists per
relation, ToUndirected, AddSelfLoops (kgwas_data.py:259-272), labels ``y`` (-1 = unlabelled) and
``n_id`` (kgwas_data.py:532-539), LD-score regression weights (kgwas_data.py:391-428), the
train/val/test split (kgwas_data.py:522-530).  File formats are the reference's (pickled dicts,
fastGWA tables); nothing is downloaded (no network) -- missing files raise.  ``from_synthetic``
builds the same object from the SynthKG generator for tests and the benchmark.

Differences by design: features are assembled with one vectorised pass instead of a Python loop per
node (kgwas_data.py:144-145,181-182,234-235); random fill-ins / GO features come from a seeded
generator (the reference draws them unseeded); edge sub-sampling
(``sample_edges``) uses a seeded permutation.
"""
from __future__ import annotations

import os
import pickle
from collections import OrderedDict

import numpy as np
import torch

from .graph import HeteroGraph, add_self_loops, to_undirected
from .utils import ldsc_regression_weights, load_dict

GO_TYPES = ['CellularComponent', 'BiologicalProcess', 'MolecularFunction']

_SNP_EMB = {'random': (None, 128), 'baselineLD': ('cell_kg/node_emb/variant_emb/baselineld_feat.pkl', 70),
            'SLDSC': ('cell_kg/node_emb/variant_emb/sldsc_feat.pkl', 165),
            'enformer': ('cell_kg/node_emb/variant_emb/enformer_feat.pkl', 20)}
_GENE_EMB = {'random': (None, 128), 'esm': ('cell_kg/node_emb/gene_emb/esm_feat.pkl', 5120),
             'pops': ('cell_kg/node_emb/gene_emb/pops_feat.pkl', 57742),
             'pops_expression': ('cell_kg/node_emb/gene_emb/pops_expression_feat.pkl', 40546)}
_GO_EMB = {'random': (None, 128), 'biogpt': ('cell_kg/node_emb/program_emb/biogpt_feat.pkl', 1600)}
# traits fitted with PLINK's logistic model when sub-sampled to <= 3000 individuals (kgwas_data.py:375,434)
_BINARY_PHENOS = ('body_BALDING1', 'cancer_BREAST', 'disease_ALLERGY_ECZEMA_DIAGNOSED', 'disease_HYPOTHYROIDISM_SELF_REP',
                  'other_MORNINGPERSON', 'pigment_SUNBURN')


def _feature_matrix(node_map, feat, dim, gen):
    """[len(node_map), dim] fp32: feat[node_map[i]] where present, U(0,1) otherwise."""
    n = len(node_map)
    x = torch.rand(n, dim, generator=gen, dtype=torch.float32)
    if feat is not None:
        rows = [i for i in range(n) if node_map[i] in feat]
        if rows:
            vals = np.asarray([feat[node_map[i]] for i in rows], dtype=np.float32)
            x[torch.as_tensor(rows)] = torch.from_numpy(vals)
    return x


def _weighted_line_fit(x, y, w):
    """(intercept, slope) minimising sum_i w_i (y_i - a - b x_i)^2 -- what sm.WLS(y, add_constant(x), weights=w).fit().params
    returns (kgwas/kgwas_data.py:456-460); w = 1: sm.OLS.  Solved on centred data (the normal equations of the raw columns are
    badly conditioned when x ~ 1..200 over 5e5 rows)."""
    x, y, w = (np.asarray(v, dtype=np.float64) for v in (x, y, w))
    sw = w.sum()
    mx, my = (w * x).sum() / sw, (w * y).sum() / sw
    dx = x - mx
    b = (w * dx * (y - my)).sum() / (w * dx * dx).sum()
    return my - b * mx, b


class KGWAS_Data:
    def __init__(self, data_path='./data/'):
        self.data_path = data_path
        os.makedirs(data_path, exist_ok=True)

    # ---------------------------------------------------------------------------------------------
    def load_kg(self, snp_init_emb='enformer', go_init_emb='random', gene_init_emb='esm', sample_edges=False,
                sample_ratio=1, seed=1, cache=True):
        """kgwas/kgwas_data.py:112-273.  ``cache`` (extra): keep / reuse the converted graph under
        ``<data_path>/cell_kg/kgwas_amd_cache/<options>`` (kgwas_amd/ingest.py): later loads map contiguous arrays
        instead of unpickling dicts and sorting 21 M edges."""
        dp = self.data_path
        opts = {'snp_init_emb': snp_init_emb, 'go_init_emb': go_init_emb, 'gene_init_emb': gene_init_emb,
                'sample_edges': bool(sample_edges), 'sample_ratio': float(sample_ratio), 'seed': int(seed)}
        cache_dir = os.path.join(dp, 'cell_kg', 'kgwas_amd_cache',
                                 '_'.join(str(v) for v in opts.values()).replace('.', 'p'))
        net = os.path.join(dp, 'cell_kg/network')
        src_files = [os.path.join(net, f) for f in ('node_idx2id.pkl', 'edge_index.pkl', 'node_id2idx.pkl')]
        for table, name in ((_SNP_EMB, snp_init_emb), (_GENE_EMB, gene_init_emb), (_GO_EMB, go_init_emb)):
            if name in table and table[name][0]:
                src_files.append(os.path.join(dp, table[name][0]))
        if cache:
            from . import ingest
            meta = ingest.read_meta(cache_dir)
            # (a cache converted from other files -- size / mtime of the pickles changed -- is stale: rebuild it)
            if meta is not None and meta.get('options') == opts and \
                    meta.get('sources') == ingest.source_fingerprint(src_files):
                print('--loading KG (cached)---')
                ingest.load(self, cache_dir)
                return
        for f in ('node_idx2id.pkl', 'edge_index.pkl', 'node_id2idx.pkl'):
            if not os.path.exists(os.path.join(net, f)):
                raise FileNotFoundError(f'{os.path.join(net, f)} missing (no network access to download the '
                                        'KGWAS core data; use KGWAS_Data.from_synthetic for a synthetic KG)')
        print('--loading KG---')
        self.idx2id = load_dict(os.path.join(net, 'node_idx2id.pkl'))
        edge_index_all = load_dict(os.path.join(net, 'edge_index.pkl'))
        self.id2idx = load_dict(os.path.join(net, 'node_id2idx.pkl'))
        gen = torch.Generator().manual_seed(seed)
        data = HeteroGraph()

        def emb(table, name, kind):
            if name not in table:
                raise NotImplementedError(f'{kind} embedding {name!r}')
            path, dim = table[name]
            return (load_dict(os.path.join(dp, path)) if path else None), dim

        feat, self.snp_init_dim_size = emb(_SNP_EMB, snp_init_emb, 'SNP')
        data['SNP'].x = _feature_matrix(self.idx2id['SNP'], feat, self.snp_init_dim_size, gen)
        feat, self.gene_init_dim_size = emb(_GENE_EMB, gene_init_emb, 'gene')
        data['Gene'].x = _feature_matrix(self.idx2id['Gene'], feat, self.gene_init_dim_size, gen)
        feat, self.go_init_dim_size = emb(_GO_EMB, go_init_emb, 'GO')
        for t in GO_TYPES:
            data[t].x = _feature_matrix(self.idx2id[t], feat, self.go_init_dim_size, gen)

        edges = OrderedDict()
        for et, ei in edge_index_all.items():                       # kgwas_data.py:259-270
            ei = np.asarray(ei, dtype=np.int64).reshape(2, -1)
            if sample_edges:
                k = int(ei.shape[1] * sample_ratio)
                perm = torch.randperm(ei.shape[1], generator=gen)[:k].numpy()
                ei = ei[:, perm]
            edges[tuple(et)] = ei
        self._finish_graph(data, edges)
        if cache:
            try:
                ingest.convert(self, cache_dir, opts, ingest.source_fingerprint(src_files))
            except OSError as e:                     # read-only data directory: run without a cache
                print(f'KG cache not written: {e}')

    def _finish_graph(self, data: HeteroGraph, edges):
        nn_ = data.num_nodes_dict
        und = add_self_loops(to_undirected(edges, nn_), nn_)      # kgwas_data.py:271-272
        for et, ei in und.items():
            data[et].edge_index = torch.from_numpy(np.ascontiguousarray(ei))
        self.data = data

    @classmethod
    def from_synthetic(cls, scale=1.0, seed=1, mode='fast', data_path='/tmp/kgwas_synth', n_labelled=None,
                       gwas_kind='causal', sample_size=None, feat_dims=None, split=True, snp_scale=1.0,
                       sample_edges=False, sample_ratio=1.0, node_counts=None):
        """SynthKG + synthetic summary statistics through the same pipeline as the real files.
        ``gwas_kind`` mirrors the reference's four label sources (BASELINE.json configs): 'causal' / 'null' = the
        simulations of load_simulation_gwas (N = 5000, kgwas_data.py:275-294), 'subsample' = load_gwas_subsample
        (N = ``sample_size``, default 10000, :367-389), 'full_cohort' = load_full_gwas (N = 387113, :341-365).
        ``sample_edges`` / ``sample_ratio``: load_kg's edge thinning (:261-268) -- int(E * ratio) edges of every ORIGINAL
        relation, seeded permutation, before ToUndirected + AddSelfLoops."""
        import pandas as pd
        from .synth import FEAT_DIMS, make_synth_edges, make_synth_gwas
        self = cls(data_path)
        if sample_size is None:
            sample_size = {'subsample': 10000, 'full_cohort': 387113}.get(gwas_kind, 5000)
        edges, nc = make_synth_edges(scale, seed, node_counts=node_counts, snp_scale=snp_scale)
        if sample_edges:
            gen_e = torch.Generator().manual_seed(seed)
            for et in list(edges.keys()):
                ei = edges[et]
                k = int(ei.shape[1] * sample_ratio)
                perm = torch.randperm(ei.shape[1], generator=gen_e)[:k].numpy()
                edges[et] = ei[:, perm]
        dims = dict(FEAT_DIMS[mode])
        if feat_dims:
            dims.update(feat_dims)
        gen = torch.Generator().manual_seed(seed)
        data = HeteroGraph()
        for t, n in nc.items():
            f = dims['SNP'] if t == 'SNP' else dims['Gene'] if t == 'Gene' else dims['GO']
            data[t].x = torch.rand(n, f, generator=gen, dtype=torch.float32)
        self.snp_init_dim_size, self.gene_init_dim_size, self.go_init_dim_size = dims['SNP'], dims['Gene'], dims['GO']
        self.idx2id = {t: _IdentityMap(n, t) for t, n in nc.items()}
        self.id2idx = {t: _IdentityMap(n, t, inverse=True) for t, n in nc.items()}
        self._finish_graph(data, edges)
        n_snp = nc['SNP']
        if n_labelled is None:
            n_labelled = int(round(n_snp * 542_758 / 784_256))      # demo/kgwas_101.ipynb:57-59
        g = make_synth_gwas(n_snp, n_labelled, seed, gwas_kind, sample_size)
        self.lr_uni = pd.DataFrame({'#CHROM': 1, 'ID': [f'rs{i}' for i in g['ids']], 'P': g['P'], 'N': g['N'],
                                    'chi': g['y']})
        self._synth_ld = (g['ld_score'], g['w_ld_score'])
        self.idx2id['SNP'] = _IdentityMap(n_snp, 'rs')
        self.id2idx['SNP'] = _IdentityMap(n_snp, 'rs', inverse=True)
        self.sample_size = sample_size
        self.pheno = {'causal': 'simulation', 'null': 'simulation'}.get(gwas_kind, 'synthetic_' + gwas_kind)
        self.seed = seed
        self.process_gwas_file()
        if split:
            self.prepare_split()
        return self

    # --- GWAS loaders (kgwas_data.py:275-389) -------------------------------------------------------
    def load_simulation_gwas(self, simulation_type, seed):
        import pandas as pd
        dp = self.data_path
        small_cohort, num_causal_hits, heritability = 5000, 20000, 0.3
        self.sample_size = small_cohort
        name = {'causal_link': f'simulation_gwas/causal_link_simulation/{num_causal_hits}_{seed}_{heritability}_graph_funct_v2_ggi.fastGWA',
                'causal': f'simulation_gwas/causal_simulation/{num_causal_hits}_{seed}_{heritability}_{small_cohort}_graph_funct_v2.fastGWA',
                'null': f'simulation_gwas/null_simulation/{num_causal_hits}_{seed}_{heritability}_{small_cohort}.fastGWA'}[simulation_type]
        lr_uni = pd.read_csv(os.path.join(dp, name), sep='\t')
        if ('SNP' in lr_uni.columns.values) and ('ID' in lr_uni.columns.values):
            self.lr_uni = lr_uni.rename(columns={'CHR': '#CHROM'})
        else:
            self.lr_uni = lr_uni.rename(columns={'CHR': '#CHROM', 'SNP': 'ID'})
        self.seed = seed
        self.pheno = 'simulation'

    def load_external_gwas(self, path=None, seed=42, example_file=False):
        import pandas as pd
        if example_file:
            path = os.path.join(self.data_path, 'biochemistry_Creatinine_fastgwa_full_10000_1.fastGWA')
            if not os.path.exists(path):
                raise FileNotFoundError(path + ' (example file must be downloaded beforehand; no network)')
        if path is None:
            raise ValueError('A valid path must be provided or example_file must be set to True.')
        lr_uni = pd.read_csv(path, sep=None, engine='python')
        for col, msg in (('CHR', 'CHR chromosome'), ('SNP', 'SNP column'), ('P', 'P column'),
                         ('N', 'N column number of sample size')):
            if col not in lr_uni.columns.values:
                raise ValueError(f'{msg} not in the file!')
        lr_uni = lr_uni.rename(columns={'CHR': '#CHROM', 'SNP': 'ID'})
        n_old = len(lr_uni)
        lr_uni = lr_uni[lr_uni.ID.isin(set(self.idx2id['SNP'].values()))]
        print('Number of SNPs in the KG:', len(self.idx2id['SNP']))
        print('Number of SNPs in the GWAS:', n_old)
        print('Number of SNPs in the KG variant set:', len(lr_uni))
        self.lr_uni = lr_uni
        self.sample_size = lr_uni.N.values[0]
        self.pheno = 'EXTERNAL'
        self.seed = seed

    def load_full_gwas(self, pheno, seed=42):
        import pandas as pd
        self.pheno = pheno
        lr_uni = pd.read_csv(os.path.join(self.data_path, 'full_gwas', f'{pheno}_with_rel_fastgwa.fastGWA'), sep='\t')
        self.lr_uni = lr_uni.rename(columns={'CHR': '#CHROM', 'SNP': 'ID'})
        self.seed = seed
        self.sample_size = 387113

    def load_gwas_subsample(self, pheno, sample_size, seed):
        import pandas as pd
        self.sample_size, self.pheno = sample_size, pheno
        binary = pheno in _BINARY_PHENOS
        sub = os.path.join(self.data_path, 'subsample_gwas')
        if sample_size > 3000:
            lr_uni = pd.read_csv(os.path.join(sub, f'{pheno}_fastgwa_full_{sample_size}_{seed}.fastGWA'), sep='\t')
            lr_uni = lr_uni.rename(columns={'CHR': '#CHROM', 'SNP': 'ID'})
        else:
            suffix = '.PHENO1.glm.logistic.hybrid' if binary else '.PHENO1.glm.linear'
            lr_uni = pd.read_csv(os.path.join(sub, f'{pheno}_plink_{sample_size}_{seed}{suffix}'), sep='\t')
        self.lr_uni = lr_uni
        self.seed = seed

    # --- labels + LD weights (kgwas_data.py:391-500) ----------------------------------------------
    def process_gwas_file(self, label='chi'):
        import pandas as pd
        lr_uni = self.lr_uni
        if getattr(self, '_synth_ld', None) is not None:
            lr_uni['ld_score'], lr_uni['w_ld_score'] = self._synth_ld
        else:
            dp = self.data_path
            ld_scores = pd.read_csv(os.path.join(dp, 'ld_score/filter_genotyped_ldscores.csv'))
            w_ld_scores = pd.read_csv(os.path.join(dp, 'ld_score/ldscores_from_data.csv'))
            ld_map = pd.Series(ld_scores.iloc[:, 1].values, index=ld_scores.iloc[:, 0].values)
            wld_map = pd.Series(w_ld_scores.iloc[:, 1].values, index=w_ld_scores.iloc[:, 0].values)
            ld_map = ld_map[~ld_map.index.duplicated(keep='last')]
            wld_map = wld_map[~wld_map.index.duplicated(keep='last')]
            # SNPs without an LD score get the minimum (kgwas_data.py:410-417); w_ld excludes the SNP itself -> +1
            lr_uni['ld_score'] = lr_uni.ID.map(ld_map).fillna(ld_map.min()).values
            lr_uni['w_ld_score'] = 1 + lr_uni.ID.map(wld_map).fillna(wld_map.min()).values
        m = 15000000
        n = self.sample_size if 'N' not in lr_uni.columns.values else np.mean(lr_uni.N)
        h_g_2 = 0.5
        w = ldsc_regression_weights(lr_uni['ld_score'].values, lr_uni['w_ld_score'].values, n, m, h_g_2)
        w = w / np.mean(w)
        print('ldsc_weight mean: ', np.mean(w))
        self.ldsc_weight = w
        self.rs_id_to_ldsc_weight = dict(zip(lr_uni.ID.values, w))
        if label == 'chi':
            # branch order of kgwas_data.py:431-446: pre-computed chi; Z_STAT for the binary traits run through PLINK's
            # logistic model (sample_size <= 3000); BETA/SE; else the P column -- NaN labels become 0 in every branch
            if 'chi' in lr_uni.columns.values:
                lr_uni['y'] = lr_uni['chi'].values
            elif getattr(self, 'pheno', None) in _BINARY_PHENOS and getattr(self, 'sample_size', 1 << 30) <= 3000:
                lr_uni['y'] = lr_uni['Z_STAT'].values ** 2
                lr_uni['y'] = lr_uni.y.fillna(0)
            elif ('BETA' in lr_uni.columns.values) and ('SE' in lr_uni.columns.values):
                lr_uni['y'] = ((lr_uni['BETA'] / lr_uni['SE']).values ** 2)
                lr_uni['y'] = lr_uni.y.fillna(0)
            else:
                from scipy.stats import chi2
                lr_uni['y'] = chi2.ppf(1 - lr_uni['P'].values, 1)
                lr_uni['y'] = lr_uni.y.fillna(0)
        elif label in ('residual-w-ld', 'residual-ld', 'residual-ld-ols', 'residual-ld-ols-abs'):
            # kgwas_data.py:448-500: chi-square from BETA / SE, NaN -> 0, then the residual of a straight-line fit on an LD
            # score -- weighted by the LDSC weights (sm.WLS) for the first two, unweighted (sm.OLS) for the '-ols' ones.
            # As in the reference: 'residual-w-ld' fits on w_ld_score; the three 'residual-ld*' labels FIT on ld_score but
            # PREDICT with w_ld_score (:475,487,499) -- kept, a label is only comparable to the reference's if it is the same
            # number.  statsmodels is not needed for a two-parameter least-squares fit.
            lr_uni['y'] = (lr_uni['BETA'] / lr_uni['SE']).values ** 2
            lr_uni['y'] = lr_uni.y.fillna(0)
            y = lr_uni.y.values.astype(np.float64)
            fit_x = (lr_uni.w_ld_score if label == 'residual-w-ld' else lr_uni.ld_score).values.astype(np.float64)
            wt = np.asarray(w, dtype=np.float64) if label in ('residual-w-ld', 'residual-ld') else np.ones_like(y)
            if label in ('residual-w-ld', 'residual-ld'):
                lr_uni['ld_weight'] = wt
            a, b = _weighted_line_fit(fit_x, y, wt)
            y_pred = a + b * lr_uni.w_ld_score.values
            lr_uni['y'] = np.abs(y - y_pred) if label == 'residual-ld-ols-abs' else y - y_pred
        else:
            raise NotImplementedError(f"label {label!r}: the reference defines 'chi', 'residual-w-ld', 'residual-ld', "
                                      "'residual-ld-ols', 'residual-ld-ols-abs' (kgwas/kgwas_data.py:429-500)")
        id2idx = self.id2idx['SNP']
        self.all_ids = np.array([id2idx[i] for i in lr_uni.ID.values], dtype=np.int64)
        self.y = lr_uni.y.values
        self.lr_uni = lr_uni

    def prepare_split(self, test_set_fraction_data=0.05):
        """kgwas_data.py:522-545: sklearn train_test_split twice (5 % test, then 5 % of the rest val)."""
        from sklearn.model_selection import train_test_split
        if not np.isfinite(np.asarray(self.y, dtype=np.float64)).all():
            raise ValueError('non-finite GWAS labels: one NaN label turns the loss, and through Adam every parameter, into NaN')
        train_val_ids, test_ids, y_train_val, y_test = train_test_split(
            self.all_ids, self.y, test_size=test_set_fraction_data, random_state=self.seed)
        train_ids, val_ids, y_train, y_val = train_test_split(
            train_val_ids, y_train_val, test_size=0.05, random_state=self.seed)
        self.train_input_nodes = ('SNP', train_ids)
        self.val_input_nodes = ('SNP', val_ids)
        self.test_input_nodes = ('SNP', test_ids)
        y_snp = torch.zeros(self.data['SNP'].x.shape[0]) - 1
        y_snp[train_ids] = torch.tensor(y_train).float()
        y_snp[val_ids] = torch.tensor(y_val).float()
        y_snp[test_ids] = torch.tensor(y_test).float()
        self.data['SNP'].y = y_snp
        for t in self.data.node_types:
            self.data[t].n_id = torch.arange(self.data[t].x.shape[0])
        self.data.train_mask = train_ids
        self.data.val_mask = val_ids
        self.data.test_mask = test_ids
        self.data.all_mask = self.all_ids
        self.data._extra.pop('_device_graphs', None)   # labels changed: rebuild resident copies lazily

    def get_pheno_list(self):
        return {'large_cohort': [],
                '21_indep_traits': ['body_BALDING1', 'disease_ALLERGY_ECZEMA_DIAGNOSED',
                                    'disease_HYPOTHYROIDISM_SELF_REP', 'pigment_SUNBURN', '21001', '50', '30080',
                                    '30070', '30010', '30000', 'biochemistry_AlkalinePhosphatase',
                                    'biochemistry_AspartateAminotransferase', 'biochemistry_Cholesterol',
                                    'biochemistry_Creatinine', 'biochemistry_IGF1', 'biochemistry_Phosphate',
                                    'biochemistry_Testosterone_Male', 'biochemistry_TotalBilirubin',
                                    'biochemistry_TotalProtein', 'biochemistry_VitaminD', 'bmd_HEEL_TSCOREz']}


class _IdentityMap:
    """idx <-> id map of a synthetic graph without materialising 784 k Python strings:
    forward: i -> f'{prefix}{i}', inverse: f'{prefix}{i}' -> i."""

    def __init__(self, n, prefix, inverse=False):
        self.n, self.prefix, self.inverse = n, prefix, inverse

    def __len__(self):
        return self.n

    def __getitem__(self, k):
        if self.inverse:
            return int(k[len(self.prefix):])
        return f'{self.prefix}{int(k)}'

    def __contains__(self, k):
        try:
            i = self[k] if self.inverse else int(k)
            return 0 <= int(i) < self.n
        except (ValueError, TypeError):
            return False

    def values(self):
        if self.inverse:
            return range(self.n)
        return (f'{self.prefix}{i}' for i in range(self.n))
