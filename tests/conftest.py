import os
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    torch.set_num_threads(min(16, torch.get_num_threads()))   # CPU oracle: more threads only add fork/join cost
    config.addinivalue_line('markers', 'gpu: needs a ROCm GPU (MI355X); run with -m gpu')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def small_kg():
    """SynthKG at 1 % scale (7.8 k SNPs, 200 genes, hubs above KGW_CHUNK) with narrow gene features."""
    from kgwas_amd.kgwas_data import KGWAS_Data
    return KGWAS_Data.from_synthetic(scale=0.01, seed=1, feat_dims={'Gene': 96}, data_path='/tmp/kgwas_synth_small')


@pytest.fixture(scope='session')
def tiny_kg():
    from kgwas_amd.kgwas_data import KGWAS_Data
    return KGWAS_Data.from_synthetic(scale=0.002, seed=3, feat_dims={'Gene': 40}, data_path='/tmp/kgwas_synth_tiny')


def make_edge_case_graph():
    """Hand-made hetero graph with the corner cases of SURVEY.md 4: degree-0 rows, degree-1 rows, a hub
    with > 4 chunks, an EMPTY relation, duplicate edges, existing self-loops, a GO-style leaf type."""
    from kgwas_amd.graph import HeteroGraph, add_self_loops, to_undirected
    rng = np.random.default_rng(7)
    n = OrderedDict([('SNP', 3000), ('Gene', 12), ('CellularComponent', 5), ('BiologicalProcess', 4),
                     ('MolecularFunction', 3)])
    e = OrderedDict()
    hub = np.stack([np.arange(1500), np.zeros(1500, dtype=np.int64)])               # gene 0: 1500 in-edges
    few = np.stack([rng.integers(0, 3000, 40), rng.integers(1, 12, 40)])
    dup = np.array([[5, 5, 5], [3, 3, 3]])                                           # duplicate edges
    e[('SNP', 'ABC', 'Gene')] = np.concatenate([hub, few, dup], axis=1)
    e[('SNP', 'TSS', 'Gene')] = np.stack([np.arange(0, 3000, 7), np.arange(0, 3000, 7) % 12])
    e[('SNP', 'EMPTY', 'Gene')] = np.zeros((2, 0), dtype=np.int64)                   # empty relation
    e[('Gene', 'G2G', 'Gene')] = np.array([[0, 1, 2, 2, 3, 0], [1, 2, 0, 2, 3, 5]])  # has self-loops 2,3
    e[('Gene', 'G-CC', 'CellularComponent')] = np.stack([rng.integers(0, 12, 20), rng.integers(0, 5, 20)])
    e[('Gene', 'G-BP', 'BiologicalProcess')] = np.stack([rng.integers(0, 12, 9), rng.integers(0, 4, 9)])
    e[('Gene', 'G-MF', 'MolecularFunction')] = np.array([[11], [2]])
    g = torch.Generator().manual_seed(11)
    data = HeteroGraph()
    dims = {'SNP': 20, 'Gene': 24}
    for t, k in n.items():
        data[t].x = torch.rand(k, dims.get(t, 16), generator=g)
    und = add_self_loops(to_undirected(e, n), n)
    for et, ei in und.items():
        data[et].edge_index = torch.from_numpy(np.ascontiguousarray(ei))
    data['SNP'].y = torch.rand(3000, generator=g)
    return data, dims


@pytest.fixture(scope='session')
def edge_case_graph():
    return make_edge_case_graph()
