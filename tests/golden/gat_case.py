"""The tiny heterogeneous graph + fixed weights behind tests/golden/gat_small.npz.

Everything here is exact integer arithmetic (a 64-bit mix hash), so the inputs are bit-identical on every machine
and numpy / torch version: the committed file holds OUTPUTS (attention, activations, prediction, loss, gradients,
sampled node / edge sets, transformed edge lists), this module regenerates the INPUTS they belong to.

Graph (corner cases of SURVEY.md 4): 5 node types; three SNP->Gene relations (one with a hub row of 300 in-edges
> KGW_CHUNK = 128 plus duplicate edges, one regular, one EMPTY), one Gene-Gene relation with pre-existing self-loops,
three Gene->GO relations; SNPs without any edge (zero-degree seeds).  ``original_edges()`` is what the reference's
``edge_index.pkl`` would hold; the reference then applies ToUndirected + AddSelfLoops (kgwas_data.py:271-272).
"""
from collections import OrderedDict

import numpy as np

NODES = OrderedDict([('SNP', 400), ('Gene', 10), ('CellularComponent', 4), ('BiologicalProcess', 3),
                     ('MolecularFunction', 3)])
DIMS = {'SNP': 20, 'Gene': 24, 'GO': 16}
HIDDEN = 128
NUM_LAYERS = 2
BATCH = 16
SEEDS = np.array([0, 5, 301, 302, 7, 150, 299, 3, 399, 398, 6, 9, 12, 200, 100, 50], dtype=np.int64)
GO_TYPES = ('CellularComponent', 'BiologicalProcess', 'MolecularFunction')


def hash01(n, salt):
    """n float64 values in [0, 1): splitmix64 of (index, salt); exact integer arithmetic."""
    with np.errstate(over='ignore'):
        v = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(salt) * np.uint64(0xBF58476D1CE4E5B9)
        v ^= v >> np.uint64(30); v *= np.uint64(0xBF58476D1CE4E5B9)
        v ^= v >> np.uint64(27); v *= np.uint64(0x94D049BB133111EB)
        v ^= v >> np.uint64(31)
    return (v >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def hash_int(n, salt, hi):
    return np.minimum((hash01(n, salt) * hi).astype(np.int64), hi - 1)


def original_edges():
    e = OrderedDict()
    hub = np.stack([np.arange(300), np.zeros(300, dtype=np.int64)])                       # gene 0: 300 in-edges
    few = np.stack([hash_int(40, 101, 400), 1 + hash_int(40, 102, 9)])
    dup = np.array([[5, 5, 5], [3, 3, 3]], dtype=np.int64)                                  # duplicate edges
    e[('SNP', 'ABC', 'Gene')] = np.concatenate([hub, few, dup], axis=1)
    tss = np.arange(0, 400, 3, dtype=np.int64)
    e[('SNP', 'TSS', 'Gene')] = np.stack([tss, tss % 10])
    e[('SNP', 'EMPTY', 'Gene')] = np.zeros((2, 0), dtype=np.int64)
    e[('Gene', 'G2G', 'Gene')] = np.array([[0, 1, 2, 2, 3, 0, 7], [1, 2, 0, 2, 3, 5, 8]], dtype=np.int64)
    e[('Gene', 'G-CC', 'CellularComponent')] = np.stack([hash_int(14, 103, 10), hash_int(14, 104, 4)])
    e[('Gene', 'G-BP', 'BiologicalProcess')] = np.stack([hash_int(9, 105, 10), hash_int(9, 106, 3)])
    e[('Gene', 'G-MF', 'MolecularFunction')] = np.array([[9], [2]], dtype=np.int64)
    return e


def features():
    x = OrderedDict()
    for k, (t, n) in enumerate(NODES.items()):
        d = DIMS['SNP'] if t == 'SNP' else DIMS['Gene'] if t == 'Gene' else DIMS['GO']
        x[t] = hash01(n * d, 200 + k).reshape(n, d).astype(np.float32)
    return x


def labels_and_weights():
    y = (4.0 * hash01(NODES['SNP'], 300)).astype(np.float32)
    w = 0.5 + hash01(NODES['SNP'], 301)                       # float64 LD weights
    return y, w


def parameters(edge_types):
    """{reference state_dict name: float32 array}; uniform with the glorot / nn.Linear bounds, NON-zero biases."""
    P = OrderedDict()
    salt = [1000]

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
    P['lin.bias'] = np.array([0.25], dtype=np.float32)      # keeps most seed predictions on the active side of the ReLU
    return P
