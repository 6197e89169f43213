"""The BENCHMARK-SHAPED case behind tests/golden/gat_wide.npz: a graph large enough that every product of the training
step takes the route it takes on the full KG -- above all the first gene Linear over the RESIDENT wide feature matrix
(kgwas/model.py:13,19 on kgwas_data.py:236) and its weight gradient on kgw_gemm3 (k_g3_gemm), which the small cases
never reach (their gene matrices are 24 - 640 wide).

Deliberately awkward sizes: 4 613 genes (not a multiple of 32: the real KG's gene count need not be one) with 1 050-wide
features (not a multiple of 32 either: the 57 742-wide mode='full' features are not).  Everything is the exact integer
hashing of tests/golden/gat_case.py, so the inputs are bit-identical on every machine; the committed file holds outputs.

Shape of a 512-seed batch (2 hops): ~1.4 k hop-1 genes, every gene and > 4 096 GO terms at hop 2, > 16 384 SNP rows --
the row counts at which kgwas_amd's own MFMA kernels take every Linear (tests run with the library fallback forbidden).
"""
from collections import OrderedDict

import numpy as np

from tests.golden.gat_case import hash01, hash_int

NODES = OrderedDict([('SNP', 60000), ('Gene', 4613), ('CellularComponent', 400), ('BiologicalProcess', 3100),
                     ('MolecularFunction', 1100)])
DIMS = {'SNP': 20, 'Gene': 1050, 'GO': 128}
HIDDEN = 128
NUM_LAYERS = 2
BATCH = 512
N_STEPS = 3                     # Adam steps of the committed trajectory (batches: SEEDS[k * BATCH:(k + 1) * BATCH])
LR, WEIGHT_DECAY = 1e-3, 5e-4
GO_TYPES = ('CellularComponent', 'BiologicalProcess', 'MolecularFunction')


def seeds():
    """N_STEPS batches of 512 distinct SNP ids, genome-ordered inside a batch like the reference's loader (no shuffle)."""
    n = NODES['SNP']
    perm = np.argsort(hash01(n, 77), kind='stable')[:N_STEPS * BATCH].astype(np.int64)
    return np.concatenate([np.sort(perm[k * BATCH:(k + 1) * BATCH]) for k in range(N_STEPS)])


def original_edges():
    """What edge_index.pkl would hold (before ToUndirected + AddSelfLoops, kgwas_data.py:271-272)."""
    ns, ng = NODES['SNP'], NODES['Gene']
    e = OrderedDict()
    snp = np.arange(ns, dtype=np.int64)
    e[('SNP', 'TSS', 'Gene')] = np.stack([snp, snp * ng // ns])                                # one gene per SNP, in genome order
    g = hash_int(90000, 11, ng)
    off = (hash_int(90000, 12, 600) - 300).astype(np.int64)
    e[('SNP', 'ABC', 'Gene')] = np.stack([np.clip(g * ns // ng + off, 0, ns - 1), g])          # cis windows around a gene
    hub = np.stack([hash_int(4000, 13, ns), np.full(4000, 17, dtype=np.int64)])                # gene 17: a 4 000-edge hub row
    few = np.stack([hash_int(30000, 14, ns), hash_int(30000, 15, ng)])
    e[('SNP', 'eQTL', 'Gene')] = np.concatenate([hub, few], axis=1)
    pop = (hash01(ng, 16) ** 3 * ng).astype(np.int64)                                          # heavy-tailed gene popularity
    a = pop[hash_int(120000, 17, ng)]
    b = hash_int(120000, 18, ng)
    e[('Gene', 'Gene-Literature-Gene', 'Gene')] = np.stack([a, b])
    e[('Gene', 'Gene-Reaction-Gene', 'Gene')] = np.stack([hash_int(40000, 19, ng), pop[hash_int(40000, 20, ng)]])
    for k, (rel, t, m) in enumerate((('Gene-Colocalizes-CellularComponent', 'CellularComponent', 12000),
                                     ('Gene-Associates-BiologicalProcess', 'BiologicalProcess', 60000),
                                     ('Gene-Contributes-MolecularFunction', 'MolecularFunction', 25000))):
        nt = NODES[t]
        cover = np.stack([hash_int(nt, 30 + 2 * k, ng), np.arange(nt, dtype=np.int64)])        # every term has a gene
        rnd = np.stack([hash_int(m, 31 + 2 * k, ng), (hash01(m, 60 + k) ** 2 * nt).astype(np.int64)])
        e[('Gene', rel, t)] = np.concatenate([cover, rnd], axis=1)
    return e


def features():
    x = OrderedDict()
    for k, (t, n) in enumerate(NODES.items()):
        d = DIMS['SNP'] if t == 'SNP' else DIMS['Gene'] if t == 'Gene' else DIMS['GO']
        x[t] = hash01(n * d, 400 + k).reshape(n, d).astype(np.float32)
    return x


def labels_and_weights():
    y = (4.0 * hash01(NODES['SNP'], 500)).astype(np.float32)
    w = 0.5 + hash01(NODES['SNP'], 501)                       # float64 LD weights
    return y, w


def parameters(edge_types):
    """{reference state_dict name: float32 array}; uniform with the glorot / nn.Linear bounds, NON-zero biases."""
    P = OrderedDict()
    salt = [3000]

    def uni(shape, a):
        salt[0] += 1
        n = int(np.prod(shape))
        return ((2.0 * hash01(n, salt[0]) - 1.0) * a).reshape(shape).astype(np.float32)

    C = HIDDEN
    for l in range(NUM_LAYERS):
        for et in edge_types:
            pre = f'convs.{l}.convs.{"__".join(et)}.'
            P[pre + 'att_src'] = uni((1, 1, C), np.sqrt(6.0 / (1 + C)))
            P[pre + 'att_dst'] = uni((1, 1, C), np.sqrt(6.0 / (1 + C)))
            P[pre + 'bias'] = uni((C,), 0.1)
            P[pre + 'lin_src.weight'] = uni((C, C), np.sqrt(6.0 / (2 * C)))
            if et[0] != et[2]:
                P[pre + 'lin_dst.weight'] = uni((C, C), np.sqrt(6.0 / (2 * C)))
    for name, d in (('snp_feat_mlp', DIMS['SNP']), ('go_feat_mlp', DIMS['GO']), ('gene_feat_mlp', DIMS['Gene'])):
        for lin, k in (('FC_hidden', d), ('FC_hidden2', C), ('FC_output', C)):
            P[f'{name}.{lin}.weight'] = uni((C, k), 1.0 / np.sqrt(k))
            P[f'{name}.{lin}.bias'] = uni((C,), 1.0 / np.sqrt(k))
    P['lin.weight'] = uni((1, C), 1.0 / np.sqrt(C))
    P['lin.bias'] = np.array([0.25], dtype=np.float32)
    return P
