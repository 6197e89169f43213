"""Shared access to tests/golden/gat_small.npz and the inputs it belongs to (tests/golden/gat_case.py)."""
import functools
import os

import numpy as np
import torch

from tests.golden import gat_case as gc
from tests.golden.make_gat_golden import build_oracle, transformed_edges  # noqa: F401  (re-exported)

HERE = os.path.dirname(os.path.abspath(__file__))


@functools.lru_cache(maxsize=1)
def golden():
    return np.load(os.path.join(HERE, 'golden', 'gat_small.npz'), allow_pickle=False)


def sampled_inputs():
    """(x_dict float64, edge_index_dict, n_id, y[seeds] float64, w[seeds] float64) of the committed batch, in the
    oracle sampler's (PyG first-seen) local order."""
    from oracle.pyg_semantics import FullNeighborSampler
    und = transformed_edges()
    n_id, ei, _ = FullNeighborSampler(und, dict(gc.NODES), gc.NUM_LAYERS).sample('SNP', gc.SEEDS)
    feats = gc.features()
    x = {t: torch.from_numpy(feats[t])[n_id[t]].double() for t in gc.NODES}
    y_all, w_all = gc.labels_and_weights()
    s = torch.from_numpy(gc.SEEDS)
    return x, ei, n_id, torch.from_numpy(y_all)[s].double(), torch.from_numpy(w_all)[s]


def case_graph():
    """The case as a kgwas_amd HeteroGraph (features float32, transformed edges, labels) + LD weights."""
    from kgwas_amd.graph import HeteroGraph
    data = HeteroGraph()
    feats = gc.features()
    for t in gc.NODES:
        data[t].x = torch.from_numpy(feats[t])
    for et, ei in transformed_edges().items():
        data[et].edge_index = ei
    y_all, w_all = gc.labels_and_weights()
    data['SNP'].y = torch.from_numpy(y_all)
    return data, torch.from_numpy(w_all)
