"""Harness helpers with the reference's names and contracts (kgwas/utils.py:20-45,203-233,397-434)."""
from __future__ import annotations

import os
import pickle
import sys

import numpy as np
import torch


def print_sys(s):
    """kgwas/utils.py:227-233."""
    print(s, flush=True, file=sys.stderr)


def save_dict(path, obj):
    with open(path, 'wb') as f:
        pickle.dump(obj, f)


def load_dict(path):
    with open(path, 'rb') as f:
        return pickle.load(f)


_HOST_LIB = None


def _host_lib():
    """libkgwas_host.so (kgwas_amd/csrc/host/kgw_tsv.cpp; plain host C++), or False when it has not been built."""
    global _HOST_LIB
    if _HOST_LIB is None:
        import ctypes as C
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libkgwas_host.so')
        try:
            L = C.CDLL(path)
            L.kgw_write_tsv.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char]
            L.kgw_write_tsv.restype = C.c_int
            _HOST_LIB = L
        except OSError:
            _HOST_LIB = False
    return _HOST_LIB


def write_tsv(df, path):
    """``df.to_csv(path, index=False, sep='\\t')`` (kgwas/kgwas.py:205-212) -- the same bytes, written by kgw_write_tsv from the
    columns' own buffers: pandas takes 3 - 7 s for the ~0.54 M-row result table of a run, this ~0.2 s.  Anything the native writer
    does not cover (other dtypes, a string that needs quoting, the library not built) goes through pandas itself."""
    import ctypes as C
    L = _host_lib()
    n, cols = len(df), list(df.columns)
    ok = bool(L) and n > 0 and len(cols) > 0 and all(isinstance(c, str) and not (set(c) & set('\t"\n\r')) and c for c in cols)
    # (duplicate column names make df[c] two-dimensional; extension dtypes -- nullable Int64, categoricals, ... -- have their own
    #  text form: both go through pandas)
    ok = ok and df.columns.is_unique and all(isinstance(df[c].dtype, np.dtype) for c in cols)
    keep, kinds, ptrs, offs = [], [], [], []
    if ok:
        for c in cols:
            v = df[c].to_numpy()
            off = None
            if v.dtype == np.float64:
                kind, buf = 0, np.ascontiguousarray(v)
            elif v.dtype == np.float32:
                kind, buf = 1, np.ascontiguousarray(v)
            elif v.dtype.kind in 'iu' and v.dtype.itemsize <= 8 and not (v.dtype == np.uint64 and v.size and int(v.max()) >= 2 ** 63):
                kind, buf = 2, np.ascontiguousarray(v, dtype=np.int64)
            elif v.dtype == np.bool_:
                kind, buf = 3, np.ascontiguousarray(v).view(np.uint8)
            elif v.dtype == object:
                lst = v.tolist()
                if not all(type(x) is str for x in lst):
                    ok = False
                    break
                joined = ''.join(lst)
                raw = joined.encode('utf-8')
                if len(raw) != len(joined):              # (non-ASCII text: byte lengths differ from character counts)
                    ok = False
                    break
                off = np.zeros(n + 1, dtype=np.int64)
                np.cumsum(np.fromiter(map(len, lst), dtype=np.int64, count=n), out=off[1:])
                kind, buf = 4, np.frombuffer(raw, dtype=np.uint8)
                keep.append(raw)
            else:
                ok = False
                break
            keep.append(buf)
            kinds.append(kind); ptrs.append(buf.ctypes.data); offs.append(off)
    if ok:
        K = (C.c_int32 * len(cols))(*kinds)
        P = (C.c_void_p * len(cols))(*ptrs)
        O = (C.c_void_p * len(cols))(*[o.ctypes.data if o is not None else None for o in offs])
        rc = L.kgw_write_tsv(os.fsencode(path), '\t'.join(cols).encode('utf-8'), n, len(cols), K, P, O, b'\t')
        if rc == 0:
            return
        if rc == -3:
            try:
                os.remove(path)                          # (no truncated table left behind)
            except OSError:
                pass
            raise OSError(f'could not write {path}')
    df.to_csv(path, index=False, sep='\t')


def evaluate_minibatch_clean(loader, model, device):
    """kgwas/utils.py:20-39: eval-mode forward over a loader; returns {'pred', 'truth'} numpy arrays.
    Differences by design: wrapped in no_grad (the reference builds and frees autograd graphs), predictions stay on
    the GPU until the end (one D2H copy instead of one per batch), and -- for this package's own NeighborLoader on a
    ROCm device -- the per-batch forward is a captured HIP graph (kgwas_amd/graph_step.py::GraphEvalStep; set
    KGW_EVAL_EAGER=1 to issue the launches one by one instead: same kernels, same values)."""
    model.eval()
    from .sampler import NeighborLoader
    if isinstance(loader, NeighborLoader) and torch.device(device).type == 'cuda' and \
            os.environ.get('KGW_EVAL_EAGER', '0') != '1' and len(loader) > 0:
        cache = loader.__dict__.setdefault('_graph_eval', {})
        ge = cache.get(id(model))
        if ge is None:
            try:
                from .graph_step import GraphEvalStep
                # (a drop_last loader -- the reference's val loader, kgwas.py:102-103 -- never sees its tail)
                ids = loader.ids[:len(loader) * loader.batch_size] if loader.drop_last else loader.ids
                ge = GraphEvalStep(model, loader.data, loader.num_layers, (loader.input_type, ids), loader.batch_size, device)
                ge.eval_ids = ids
            except ValueError:
                ge = False                       # too few nodes for a padded static layout: eager path below
            cache.clear()
            cache[id(model)] = ge
        if ge:
            pred = ge.run()
            truth = loader.dg.y[loader.input_type][ge.eval_ids]
            return {'pred': pred.float().cpu().numpy(), 'truth': truth.float().cpu().numpy()}
    preds, truths = [], []
    with torch.no_grad():
        for batch in loader:
            batch = batch.to(device)
            bs = batch['SNP'].batch_size
            out = model(batch.x_dict, batch.edge_index_dict, bs)
            preds.append(out.reshape(-1))
            truths.append(batch['SNP'].y[:bs])
    if not preds:
        return {'pred': np.zeros(0, np.float32), 'truth': np.zeros(0, np.float32)}
    return {'pred': torch.cat(preds).float().cpu().numpy(), 'truth': torch.cat(truths).float().cpu().numpy()}


def compute_metrics(results, binary=False, coverage=None, uncertainty_reg=1, loss_fct=None):
    """kgwas/utils.py:41-45: sklearn mean_squared_error + scipy pearsonr."""
    from scipy.stats import pearsonr
    from sklearn.metrics import mean_squared_error
    return {'mse': mean_squared_error(results['pred'], results['truth']),
            'pearsonr': pearsonr(results['pred'], results['truth'])[0]}


def save_model(model, config, path_dir):
    """kgwas/utils.py:203-207: model.pt (state_dict) + config.pkl."""
    os.makedirs(path_dir, exist_ok=True)
    torch.save(model.state_dict(), os.path.join(path_dir, 'model.pt'))
    save_dict(os.path.join(path_dir, 'config.pkl'), config)


def load_pretrained(path, model):
    """kgwas/utils.py:209-222 (strips a DataParallel ``module.`` prefix)."""
    state_dict = torch.load(os.path.join(path, 'model.pt'), map_location=torch.device('cpu'), weights_only=False)
    if next(iter(state_dict))[:7] == 'module.':
        from collections import OrderedDict
        state_dict = OrderedDict((k[7:], v) for k, v in state_dict.items())
    model.load_state_dict(state_dict)
    return model


def ldsc_regression_weights(ld, w_ld, N, M, hsq, intercept=None, ii=None):
    """LD-score regression weights: the ten-line formula of kgwas/utils.py:397-434, TRANSCRIBED (same name, arguments and
    order of operations, so the values match the reference's bit for bit -- tests/golden/ref_helpers.npz):
    1 / (2 (intercept + hsq N ld / M)^2 w_ld) with ld, w_ld floored at 1 and hsq clipped to [0, 1]."""
    M = float(M)
    if intercept is None:
        intercept = 1
    hsq = min(max(hsq, 0.0), 1.0)
    ld = np.fmax(ld, 1.0)
    w_ld = np.fmax(w_ld, 1.0)
    c = hsq * N / M
    het_w = 1.0 / (2 * np.square(intercept + np.multiply(c, ld)))
    oc_w = 1.0 / w_ld
    return np.multiply(het_w, oc_w)


def get_network_weight(run, data):
    """kgwas/utils.py:437-494: raw attention weight of every edge of the knowledge graph at every layer, from the
    best model, as a DataFrame [h_idx, t_idx, weight, h_type, rel_type, t_type, layer] (duplicates of
    (h_idx, t_idx, rel_type, layer) dropped).  The reference runs its modified HeteroConv on the CPU; here the
    whole-graph pass runs on the fused kernels (HeteroGNN.raw_attention_full_graph)."""
    import pandas as pd
    model = getattr(run, 'best_model', None) or run.model
    print('Retrieving weights...')
    was_training = model.training
    model.eval()
    layers = model.raw_attention_full_graph(data.data, run.device)
    model.train(was_training)
    print('Aggregating across node types...')
    frames = []
    node_types = list(data.data.node_types)
    for k, att in enumerate(layers):
        for node_type in node_types:                                     # utils.py:469-474 iteration order
            for et, (ei, w) in att.items():
                if et[2] != node_type:
                    continue
                ei = ei.cpu().numpy()
                df = pd.DataFrame({'h_idx': ei[0].astype(np.float64), 't_idx': ei[1].astype(np.float64),
                                   'weight': w.cpu().numpy().astype(np.float64)})
                df['h_type'], df['rel_type'], df['t_type'], df['layer'] = et[0], et[1], et[2], f'l{k + 1}'
                frames.append(df)
    cols = ['h_idx', 't_idx', 'weight', 'h_type', 'rel_type', 't_type', 'layer']
    df_all = pd.concat(frames)[cols] if frames else pd.DataFrame(columns=cols)
    return df_all.drop_duplicates(['h_idx', 't_idx', 'rel_type', 'layer'])


# ------------------------------------------------------------------------------------------------------
# Disease-critical network + variant interpretation (kgwas/utils.py:496-724; SURVEY.md 8 row f-2, second half)
# ------------------------------------------------------------------------------------------------------
_NET_COLS = ['h_idx', 't_idx', 'importance', 'h_type', 't_type', 'rel_type']


def _importance_by_relation(edges, reference_edges):
    """z-score of every edge weight inside its relation, with the relation's mean and (n - 1) standard deviation taken over
    ``reference_edges`` (the reference's rel2mean / rel2std tables, kgwas/utils.py:584-589: pandas' groupby std).  Relations
    without a row in ``reference_edges`` drop out, as the reference's inner merges make them."""
    stats = reference_edges.groupby('rel_type').weight.agg(['mean', 'std'])
    kept = edges[edges.rel_type.isin(stats.index)]
    z = (kept.weight.to_numpy() - kept.rel_type.map(stats['mean']).to_numpy()) / kept.rel_type.map(stats['std']).to_numpy()
    return kept.assign(importance=z)


def _strongest_relation_per_pair(scored):
    """One row per (h_idx, t_idx): the relation / layer whose importance is the pair's maximum (every row equal to the
    maximum is kept, like the reference's merge back on the value, kgwas/utils.py:591-593), pairs in ascending order."""
    if not len(scored):
        return scored[_NET_COLS]
    top = scored.groupby(['h_idx', 't_idx']).importance.transform('max')
    imp = scored.importance
    keep = (imp == top) | (imp.isna() & top.isna())
    out = scored[keep.to_numpy()].sort_values(['h_idx', 't_idx'], kind='stable')
    return out[_NET_COLS].reset_index(drop=True)


def _names(idx, table):
    return [table[i] for i in idx.to_numpy()]


def _top_k_by_tail(frame, k):
    """{t_id: row positions of its k most important rows, most important first}; NaN importances rank first, which is
    where the reference's ascending sort read backwards puts them (kgwas/utils.py:499,505)."""
    order = frame.sort_values('importance', ascending=False, na_position='first', kind='stable')
    order = order.groupby('t_id', sort=False).head(k)
    pos = {}
    for p, t in zip(order['_pos'].to_numpy(), order['t_id'].to_numpy()):
        pos.setdefault(t, []).append(p)
    return pos


def generate_viz(run, df_network, data_path, variant_threshold=5e-8, magma_path=None, magma_threshold=0.05,
                 program_threshold=0.05, K_neighbors=3, num_cpus=1):
    """kgwas/utils.py:523-724 without the MAGMA / GSEA filter: from the attention table of ``get_network_weight`` build
      * the disease-critical network: Gene->SNP edges at the GWAS hits (V2G), Gene->Gene (G2G) and BiologicalProcess->Gene (G2P)
        edges, TSS relations left out, every weight z-scored inside its relation over the selected edges, one row per node pair
        (its strongest relation), with readable ids;
      * the variant interpretation: for every hit SNP that has a gene edge, its ``K_neighbors`` strongest genes and, around each of
        them, the strongest gene, gene-program and SNP neighbours (importance = z-score with the statistics of the selected edges).
    Returns (df_variant_interpretation, disease_critical_network) with the reference's columns.
    Checked against the frames the reference's own function returns (tests/golden/viz_network.npz).

    Differences by design: hits are ``P < variant_threshold`` (the reference hard-codes 5e-8, the default, and ignores the
    argument, utils.py:547); the per-SNP loop is table look-ups instead of a multiprocessing pool (``num_cpus`` is accepted and
    unused); misc_data/go2name.pkl is optional (terms keep their ids without it); a hit SNP without any gene edge is left out
    of the interpretation, which is what the reference's bare ``except`` amounts to (utils.py:521).
    ``magma_path`` (gene-level MAGMA output filtered by Bonferroni + GSEA prerank, utils.py:553-579) needs an external binary's
    output, statsmodels and gseapy: not built."""
    import pandas as pd
    if magma_path is not None:
        raise NotImplementedError('the MAGMA / GSEA filter (kgwas/utils.py:553-579: external MAGMA output, statsmodels, gseapy) is '
                                  'not built; pass magma_path=None for the unfiltered network')
    gwas = run.kgwas_res
    idx2id, id2idx = run.data.idx2id, run.data.id2idx
    print('Start generating disease critical network...')
    go2name = {}
    p = os.path.join(data_path, 'misc_data', 'go2name.pkl')
    if os.path.exists(p):
        go2name = load_dict(p)

    def program_name(i):
        g = idx2id['BiologicalProcess'][i]
        return go2name[g].capitalize() if g in go2name else g

    net = df_network[~df_network.rel_type.isin(['TSS', 'rev_TSS'])]
    gene_to_snp = net[(net.t_type == 'SNP') & (net.h_type == 'Gene')]
    gene_to_gene = net[(net.t_type == 'Gene') & (net.h_type == 'Gene')]
    program_to_gene = net[(net.t_type == 'Gene') & (net.h_type == 'BiologicalProcess')]
    snp_to_gene = net[(net.h_type == 'SNP') & (net.t_type == 'Gene')]
    if 'SNP' not in gwas.columns.values:
        gwas.loc[:, 'SNP'] = gwas['ID']
    hit_snps = gwas[gwas.P < variant_threshold].SNP.values
    hit_idx = np.array([id2idx['SNP'][s] for s in hit_snps], dtype=np.float64)
    print('No filters... Using all genes and gene programs...')

    # --- the network over the hits ------------------------------------------------------------------------------------
    at_hits = gene_to_snp[gene_to_snp.t_idx.isin(hit_idx)]
    v2g_hit = _strongest_relation_per_pair(_importance_by_relation(at_hits, at_hits))
    v2g_hit['rel_type'] = [r[4:] for r in v2g_hit.rel_type]                          # 'rev_ABC' -> 'ABC'
    v2g_hit['Category'] = 'V2G'
    v2g_hit['h_id'], v2g_hit['t_id'] = _names(v2g_hit.h_idx, idx2id['Gene']), _names(v2g_hit.t_idx, idx2id['SNP'])
    g2g_hit = _strongest_relation_per_pair(_importance_by_relation(gene_to_gene, gene_to_gene))
    g2g_hit['rel_type'] = [r.split('-')[1] for r in g2g_hit.rel_type]                 # 'Gene-Literature-Gene' -> 'Literature'
    g2g_hit['Category'] = 'G2G'
    g2g_hit['h_id'], g2g_hit['t_id'] = _names(g2g_hit.h_idx, idx2id['Gene']), _names(g2g_hit.t_idx, idx2id['Gene'])
    g2p_hit = _strongest_relation_per_pair(_importance_by_relation(program_to_gene, program_to_gene))
    g2p_hit['rel_type'] = [r.split('-')[1] for r in g2p_hit.rel_type]
    g2p_hit['Category'] = 'G2P'
    g2p_hit['h_id'] = [program_name(i) for i in g2p_hit.h_idx.to_numpy()]
    g2p_hit['t_id'] = _names(g2p_hit.t_idx, idx2id['Gene'])
    disease_critical_network = pd.concat((v2g_hit, g2g_hit, g2p_hit)).reset_index(drop=True)
    print('Disease critical network finished generating...')
    print('Generating variant interpretation networks...')

    # --- neighbourhoods of the hits: every edge scored, statistics still those of the selected edges ------------------------
    v2g = _strongest_relation_per_pair(_importance_by_relation(gene_to_snp, at_hits))
    v2g['h_id'], v2g['t_id'] = _names(v2g.h_idx, idx2id['Gene']), _names(v2g.t_idx, idx2id['SNP'])
    g2g = _strongest_relation_per_pair(_importance_by_relation(gene_to_gene, gene_to_gene))
    g2g['h_id'], g2g['t_id'] = _names(g2g.h_idx, idx2id['Gene']), _names(g2g.t_idx, idx2id['Gene'])
    g2g = g2g[g2g.h_idx != g2g.t_idx].reset_index(drop=True)                          # self loops say nothing
    g2p = _strongest_relation_per_pair(_importance_by_relation(program_to_gene, program_to_gene))
    g2p['h_id'] = [program_name(i) for i in g2p.h_idx.to_numpy()]
    g2p['t_id'] = _names(g2p.t_idx, idx2id['Gene'])
    g2v = _strongest_relation_per_pair(_importance_by_relation(snp_to_gene, snp_to_gene[snp_to_gene.h_idx.isin(hit_idx)]))
    g2v['h_id'], g2v['t_id'] = _names(g2v.h_idx, idx2id['SNP']), _names(g2v.t_idx, idx2id['Gene'])
    print('Number of hit snps: ', len(hit_snps))
    tables = []
    for t in (v2g, g2g, g2p, g2v):
        t = t.copy()
        t['_pos'] = np.arange(len(t))
        tables.append(t)
    v2g, g2g, g2p, g2v = tables
    v2g['rel_type_short'] = [r[4:] for r in v2g.rel_type]
    for t in (g2g, g2p):
        t['rel_type_short'] = [r.split('-')[1] for r in t.rel_type]
    g2v['rel_type_short'] = g2v.rel_type
    best = [_top_k_by_tail(t, K_neighbors) for t in (v2g, g2g, g2p, g2v)]
    parts = []
    for snp in hit_snps:
        genes_at = best[0].get(snp)
        if not genes_at:
            continue
        genes = v2g['h_id'].to_numpy()[genes_at]
        rows = [v2g.iloc[genes_at]]
        for tab, pos in ((g2g, best[1]), (g2p, best[2]), (g2v, best[3])):
            take = [q for g in genes for q in pos.get(g, ())]
            rows.append(tab.iloc[take])
        local = pd.concat(rows)
        local = local.assign(rel_type=local['rel_type_short'], QUERY_SNP=snp).drop(columns=['_pos', 'rel_type_short'])
        parts.append(local)
    df_variant_interpretation = pd.concat(parts) if parts else pd.DataFrame()
    return df_variant_interpretation, disease_critical_network
