"""SynthKG: synthetic stand-in for the KGWAS knowledge graph + GWAS labels.

The real graph (cell_kg/network/*.pkl on Harvard Dataverse) is not available offline, so the
benchmark / tests use a seeded generator that reproduces what the reference's notebooks reveal:
  * node counts: SNP 784 256 (demo/kgwas_101.ipynb:57), Gene >= 20 032, BiologicalProcess
    >= 17 248, MolecularFunction >= 4 512, CellularComponent >= 1 192
    (demo/disease_critical_network.ipynb cells 3-4);
  * relation names seen in those notebooks; ~21.4 M directed edges per layer after
    ToUndirected + AddSelfLoops (disease_critical_network.ipynb:265);
  * fast-mode feature widths SNP 20 / Gene 5120 / GO 128 (kgwas_data.py:183,236,191),
    full-mode SNP 70 / Gene 57 742 (kgwas_data.py:167,244);
  * SNP->Gene edges are locality structured (SNP index ~ genomic order) so genome-ordered seed
    batches touch ~3x smaller subgraphs than random ones (kgwas_101.ipynb:353-357).
Per-relation edge counts are NOT published; the table below is a plausible split that hits the
published total.  Everything is parameterised (``scale`` shrinks nodes and edges together).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import numpy as np
import torch

from .graph import HeteroGraph, add_self_loops, to_undirected

NODE_COUNTS = OrderedDict([
    ('SNP', 784_256), ('Gene', 20_032), ('CellularComponent', 1_192),
    ('BiologicalProcess', 17_248), ('MolecularFunction', 4_512),
])

# original (pre-ToUndirected) directed edge counts; total after transforms ~= 21.4 M
V2G = OrderedDict([('ABC', 600_000), ('TSS', 784_256), ('Exon', 150_000), ('PCHi-C', 900_000),
                   ('VEP', 100_000), ('eQTL', 500_000)])
G2G = OrderedDict([('Gene-PhysicalAssociation-Gene', 1_200_000), ('Gene-DosageLethality-Gene', 300_000),
                   ('Gene-Literature-Gene', 3_000_000), ('Gene-Signaling-Gene', 600_000),
                   ('Gene-Reaction-Gene', 1_200_000)])
G2GO = OrderedDict([
    ('Gene-Associates-BiologicalProcess', ('BiologicalProcess', 600_000)),
    ('Gene-NotAssociates-BiologicalProcess', ('BiologicalProcess', 20_000)),
    ('Gene-Colocalizes-CellularComponent', ('CellularComponent', 300_000)),
    ('Gene-NotColocalizes-CellularComponent', ('CellularComponent', 5_000)),
    ('Gene-Contributes-MolecularFunction', ('MolecularFunction', 350_000)),
    ('Gene-NotContributes-MolecularFunction', ('MolecularFunction', 5_000)),
])

FEAT_DIMS = {'fast': {'SNP': 20, 'Gene': 5120, 'GO': 128},
             'full': {'SNP': 70, 'Gene': 57_742, 'GO': 128}}


def _heavy_tail_weights(rng, n, sigma):
    w = rng.lognormal(mean=0.0, sigma=sigma, size=n)
    return w / w.sum()


def make_synth_edges(scale: float = 1.0, seed: int = 1, node_counts=None, snp_scale: float = 1.0):
    """Original directed COO lists, dict (src, rel, dst) -> int64[2, E], plus node counts.  ``snp_scale`` multiplies
    the SNP count and the SNP->Gene edge counts on top of ``scale`` (a denser genotyping array over the same genes:
    BASELINE.json's ~10 M-SNP full-cohort case is snp_scale = 12.75)."""
    rng = np.random.default_rng(seed)
    nc = OrderedDict((k, max(8, int(round(v * scale * (snp_scale if k == 'SNP' else 1.0)))))
                     for k, v in (node_counts or NODE_COUNTS).items())
    n_snp, n_gene = nc['SNP'], nc['Gene']
    edges: "OrderedDict[tuple, np.ndarray]" = OrderedDict()

    # genes sit at sorted genomic centres; each has a heavy-tailed cis window
    centre = np.sort(rng.integers(0, n_snp, size=n_gene))
    win = np.clip(rng.lognormal(mean=np.log(60.0), sigma=1.0, size=n_gene),
                  2, max(4, n_snp // 50)).astype(np.int64)
    gene_pop = _heavy_tail_weights(rng, n_gene, 1.0)
    for rel, e_full in V2G.items():
        e = max(4, int(round(e_full * scale * snp_scale)))
        if rel == 'TSS':   # every SNP -> nearest gene centre (exactly one edge per SNP)
            snp = np.arange(n_snp, dtype=np.int64)
            pos = np.searchsorted(centre, snp)
            lo = np.clip(pos - 1, 0, n_gene - 1)
            hi = np.clip(pos, 0, n_gene - 1)
            gene = np.where(np.abs(centre[lo] - snp) <= np.abs(centre[hi] - snp), lo, hi)
        else:
            gene = rng.choice(n_gene, size=e, p=gene_pop)
            off = (rng.standard_normal(e) * win[gene]).astype(np.int64)
            snp = np.clip(centre[gene] + off, 0, n_snp - 1)
        edges[('SNP', rel, 'Gene')] = np.stack([snp, gene.astype(np.int64)])

    for rel, e_full in G2G.items():
        e = min(max(4, int(round(e_full * scale))), n_gene * n_gene // 4)
        w = _heavy_tail_weights(rng, n_gene, 1.2)
        a = rng.choice(n_gene, size=e, p=w)
        b = rng.choice(n_gene, size=e, p=w)
        edges[('Gene', rel, 'Gene')] = np.stack([a, b]).astype(np.int64)

    for rel, (go_t, e_full) in G2GO.items():
        n_go = nc[go_t]
        e = max(4, int(round(e_full * scale)))
        e = min(e, n_gene * n_go // 2)
        wg = _heavy_tail_weights(rng, n_go, 1.5)
        g = rng.integers(0, n_gene, size=e)
        t = rng.choice(n_go, size=e, p=wg)
        edges[('Gene', rel, go_t)] = np.stack([g, t]).astype(np.int64)
    return edges, nc


def make_synth_kg(scale: float = 1.0, seed: int = 1, mode: str = 'fast', feat_dims=None,
                  node_counts=None) -> HeteroGraph:
    """Full pipeline of KGWAS_Data.load_kg (kgwas_data.py:112-273) on synthetic inputs:
    features U(0,1) per type, COO lists, ToUndirected, AddSelfLoops."""
    edges, nc = make_synth_edges(scale, seed, node_counts)
    dims = dict(FEAT_DIMS[mode])
    if feat_dims:
        dims.update(feat_dims)
    g = torch.Generator().manual_seed(seed)
    data = HeteroGraph()
    for t, n in nc.items():
        f = dims['SNP'] if t == 'SNP' else dims['Gene'] if t == 'Gene' else dims['GO']
        data[t].x = torch.rand(n, f, generator=g, dtype=torch.float32)
    und = add_self_loops(to_undirected(edges, nc), nc)
    for et, ei in und.items():
        data[et].edge_index = torch.from_numpy(np.ascontiguousarray(ei))
    return data


def make_synth_gwas(n_snp: int, n_labelled: int, seed: int = 1, kind: str = 'causal',
                    sample_size: int = 5000, n_causal: int = 20_000):
    """Synthetic summary statistics with the columns the reference consumes
    (kgwas_data.py:275-294,391-447): labelled SNP ids, chi-square labels y, P, N, LD scores."""
    rng = np.random.default_rng(seed)
    n_labelled = min(n_labelled, n_snp)
    ids = np.sort(rng.choice(n_snp, size=n_labelled, replace=False))
    z = rng.standard_normal(n_labelled)
    if kind not in ('causal', 'null', 'subsample', 'full_cohort'):
        raise ValueError(f'gwas kind {kind!r}')
    if kind != 'null':         # 'subsample' / 'full_cohort': the same causal architecture seen at another cohort size
        k = min(n_causal, n_labelled)
        causal = rng.choice(n_labelled, size=k, replace=False)
        z[causal] += rng.standard_normal(k) * np.sqrt(0.3 * sample_size / max(k, 1)) * 3.0
    y = z ** 2
    from scipy.stats import chi2
    p = chi2.sf(y, 1)
    ld = rng.uniform(1, 200, n_labelled)
    w_ld = 1.0 + rng.uniform(0, 10, n_labelled)
    return {'ids': ids, 'y': y.astype(np.float64), 'P': p, 'N': float(sample_size),
            'ld_score': ld, 'w_ld_score': w_ld}
