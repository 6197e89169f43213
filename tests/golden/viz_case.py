"""Inputs of the disease-critical-network golden (tests/golden/viz_network.npz): a small attention table shaped like the
output of get_network_weight (kgwas/utils.py:437-494) and the GWAS frame / id maps generate_viz reads (kgwas/utils.py:523-724).
Integer hashing only, so the inputs are identical everywhere; the committed file holds what the REFERENCE returned for them."""
import numpy as np
import pandas as pd

from tests.golden.gat_case import hash01, hash_int

N = {'SNP': 300, 'Gene': 40, 'BiologicalProcess': 14, 'CellularComponent': 4, 'MolecularFunction': 4}
K_NEIGHBORS = 3

# (head type, relation, tail type, number of edges); rev_* = the ToUndirected twins, TSS / rev_TSS are dropped by generate_viz
RELS = [('SNP', 'ABC', 'Gene', 260), ('Gene', 'rev_ABC', 'SNP', 260),
        ('SNP', 'eQTL', 'Gene', 180), ('Gene', 'rev_eQTL', 'SNP', 180),
        ('SNP', 'TSS', 'Gene', 300), ('Gene', 'rev_TSS', 'SNP', 300),
        ('SNP', 'PCHi-C', 'Gene', 3), ('Gene', 'rev_PCHi-C', 'SNP', 3),          # a relation with at most one hit edge: std = NaN
        ('Gene', 'Gene-Literature-Gene', 'Gene', 220), ('Gene', 'Gene-Reaction-Gene', 'Gene', 120),
        ('Gene', 'Gene-Associates-BiologicalProcess', 'BiologicalProcess', 90),
        ('BiologicalProcess', 'rev_Gene-Associates-BiologicalProcess', 'Gene', 90),
        ('Gene', 'Gene-NotAssociates-BiologicalProcess', 'BiologicalProcess', 50),
        ('BiologicalProcess', 'rev_Gene-NotAssociates-BiologicalProcess', 'Gene', 50),
        ('Gene', 'Gene-Colocalizes-CellularComponent', 'CellularComponent', 20),
        ('CellularComponent', 'rev_Gene-Colocalizes-CellularComponent', 'Gene', 20)]


def network():
    """The frame get_network_weight returns: float h_idx / t_idx / weight (np.vstack of ints and floats, utils.py:472), both
    layers, duplicates of (h_idx, t_idx, rel_type, layer) already dropped; self loops on the Gene-Gene relations."""
    frames = []
    salt = 7000
    for layer in ('l1', 'l2'):
        for h, r, t, m in RELS:
            salt += 3
            base = r[4:] if r.startswith('rev_') else r
            k = 100 + sum(ord(c) for c in base)                      # a relation and its twin join the same node pairs
            a, b = hash_int(m, k, N['SNP' if 'SNP' in (h, t) else h]), hash_int(m, k + 1, N['Gene'] if 'SNP' in (h, t) else N[t if h == 'Gene' else h])
            if base == 'PCHi-C':                                     # one edge at a hit SNP, two elsewhere
                hs = hit_snps()
                non = [i for i in range(N['SNP']) if i not in set(hs.tolist())][:2]
                a, b = np.array([hs[1]] + non), np.array([5, 6, 7])
            if h == 'SNP':
                hi, ti = a, b
            elif t == 'SNP':
                hi, ti = b, a
            elif h == 'Gene' and t == 'Gene':
                hi, ti = hash_int(m, k, N['Gene']), hash_int(m, k + 1, N['Gene'])
                loops = np.arange(N['Gene'])
                hi, ti = np.concatenate([hi, loops]), np.concatenate([ti, loops])        # AddSelfLoops (kgwas_data.py:272)
            elif h == 'Gene':
                hi, ti = hash_int(m, k, N['Gene']), hash_int(m, k + 1, N[t])
            else:
                hi, ti = hash_int(m, k + 1, N[h]), hash_int(m, k, N['Gene'])
            w = (hash01(len(hi), salt) - 0.4) * (1.0 + 3.0 * hash01(len(hi), salt + 1))       # raw LeakyReLU logits: any sign
            df = pd.DataFrame({'h_idx': hi.astype(np.float64), 't_idx': ti.astype(np.float64), 'weight': w.astype(np.float32).astype(np.float64)})
            df['h_type'], df['rel_type'], df['t_type'], df['layer'] = h, r, t, layer
            frames.append(df)
    return pd.concat(frames).drop_duplicates(['h_idx', 't_idx', 'rel_type', 'layer'])


def hit_snps():
    return np.sort(np.argsort(hash01(N['SNP'], 9002), kind='stable')[:25])


def gwas():
    """run.kgwas_res as generate_viz uses it: ID, P (hits are P < 5e-8, utils.py:547), N."""
    n = N['SNP']
    p = hash01(n, 9001) * 0.9 + 1e-3
    hits = hit_snps()
    p[hits] = 1e-9 * (1.0 + hash01(25, 9003))
    p[hits[0]] = 5e-8                                                  # exactly the threshold: NOT a hit (strict <)
    return pd.DataFrame({'ID': [f'rs{i}' for i in range(n)], 'P': p, 'N': 5000})


def id_maps():
    idx2id = {'SNP': {i: f'rs{i}' for i in range(N['SNP'])}, 'Gene': {i: f'GENE{i}' for i in range(N['Gene'])},
              'BiologicalProcess': {i: f'GO:{1000 + i}' for i in range(N['BiologicalProcess'])},
              'CellularComponent': {i: f'GO:{2000 + i}' for i in range(N['CellularComponent'])},
              'MolecularFunction': {i: f'GO:{3000 + i}' for i in range(N['MolecularFunction'])}}
    id2idx = {t: {v: k for k, v in m.items()} for t, m in idx2id.items()}
    return idx2id, id2idx


def go2name():
    return {f'GO:{1000 + i}': f'process number {i}' for i in range(0, N['BiologicalProcess'], 2)}      # half of the terms are named
