#!/usr/bin/env python
"""Generate tests/golden/gat_wide.npz: committed outputs of the benchmark-shaped case in tests/golden/gat_wide_case.py --
the case whose gene feature matrix is wide and tall enough (4 613 x 1 050) that the product's first gene Linear and its
weight gradient run on kgw_gemm3 inside forward / backward / Adam.

The vectors come from the float64 CPU restatement (oracle/gat_oracle.py), which tests/golden/gat_small.npz pins against
the independently derived dense formulation (oracle/dense_gat.py); this file adds the shapes, not a new derivation:
  * step 0: prediction of the 512 seeds, loss, every parameter gradient (small tensors whole; matrices of >= 128 x 128
    entries as every 37th element + sum + norm);
  * N_STEPS Adam steps (lr 1e-3, weight decay 5e-4 as L2 -- kgwas/kgwas.py:116 with a larger step): loss of every step and
    the parameter update of gene_feat_mlp.FC_hidden.weight (strided) after the last one.

    python tests/golden/make_gat_wide_golden.py          (~1 min, ~6 GB)
"""
import os
import sys
import time
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.gat_oracle import HeteroGNNOracle, weighted_mse                                         # noqa: E402
from oracle.pyg_semantics import add_self_loops_hetero, to_undirected_hetero                          # noqa: E402
from oracle.sampler_np import FullNeighborSamplerNP                                                   # noqa: E402
from tests.golden import gat_wide_case as wc                                                          # noqa: E402

GRAD_STRIDE = 37


def transformed_edges():
    e0 = OrderedDict((k, torch.from_numpy(v)) for k, v in wc.original_edges().items())
    return add_self_loops_hetero(to_undirected_hetero(e0, dict(wc.NODES)), dict(wc.NODES))


def build_oracle(edge_types, dtype=torch.float64):
    o = HeteroGNNOracle(edge_types, wc.HIDDEN, 1, wc.NUM_LAYERS, 'GAT', 'sum', wc.DIMS['SNP'], wc.DIMS['Gene'],
                        wc.DIMS['GO'], 1, dtype=dtype)
    sd = OrderedDict((k, torch.from_numpy(v).to(dtype)) for k, v in wc.parameters(edge_types).items())
    o.load_state_dict(sd, strict=True)
    return o


def oracle_steps(n_steps, on_first_backward=None):
    """Run the restatement over the case's batches; returns (oracle, losses, per-step sampled node counts)."""
    und = transformed_edges()
    edge_types = list(und.keys())
    feats = {t: torch.from_numpy(v) for t, v in wc.features().items()}
    y_all, w_all = wc.labels_and_weights()
    y_all, w_all = torch.from_numpy(y_all).double(), torch.from_numpy(w_all)
    oracle = build_oracle(edge_types)
    opt = torch.optim.Adam(oracle.parameters(), lr=wc.LR, weight_decay=wc.WEIGHT_DECAY)
    smp = FullNeighborSamplerNP(und, dict(wc.NODES), wc.NUM_LAYERS)
    ids = wc.seeds()
    losses, counts = [], []
    for k in range(n_steps):
        s = ids[k * wc.BATCH:(k + 1) * wc.BATCH]
        n_id, ei = smp.sample('SNP', s)
        x = {t: feats[t][torch.as_tensor(n_id[t])].double() for t in wc.NODES}
        opt.zero_grad()
        pred = oracle(x, ei, wc.BATCH)
        st = torch.as_tensor(s)
        loss = weighted_mse(pred, y_all[st], w_all[st])
        loss.backward()
        if k == 0 and on_first_backward is not None:
            on_first_backward(oracle, pred.detach(), loss.detach(), n_id, ei)
        opt.step()
        losses.append(float(loss.detach()))
        counts.append({t: len(n_id[t]) for t in wc.NODES})
    return oracle, losses, counts


def main():
    t0 = time.time()
    out = OrderedDict()
    und = transformed_edges()
    out['edge_type_names'] = np.array(['|'.join(et) for et in und])
    out['input_checksum'] = np.array([float(sum(int(v.sum()) for v in und.values())),
                                      float(sum(float(v.astype(np.float64).sum()) for v in wc.features().values())),
                                      float(sum(float(v.astype(np.float64).sum()) for v in wc.parameters(list(und)).values()))])
    w0 = wc.parameters(list(und))['gene_feat_mlp.FC_hidden.weight'].astype(np.float64)

    def first(oracle, pred, loss, n_id, ei):
        out['pred'] = pred.numpy().reshape(-1)
        out['loss'] = np.float64(loss.item())
        out['n_edges'] = np.int64(sum(v.shape[1] for v in ei.values()))
        none = []
        for n, p in oracle.named_parameters():
            g = p.grad
            if g is None:
                none.append(n)
                continue
            g = g.numpy()
            if g.size >= 128 * 128:
                out[f'gs_{n}'] = g.reshape(-1)[::GRAD_STRIDE].copy()
                out[f'gn_{n}'] = np.array([g.sum(), np.sqrt((g ** 2).sum())])
            else:
                out[f'g_{n}'] = g.copy()
        out['grad_none'] = np.array(none)
        out['grad_stride'] = np.int64(GRAD_STRIDE)

    oracle, losses, counts = oracle_steps(wc.N_STEPS, first)
    out['losses'] = np.array(losses)
    out['sampled_nodes'] = np.array([[c[t] for t in wc.NODES] for c in counts], dtype=np.int64)
    w1 = dict(oracle.named_parameters())['gene_feat_mlp.FC_hidden.weight'].detach().numpy()
    out['dW_gene_first_strided'] = (w1 - w0).reshape(-1)[::GRAD_STRIDE].copy()
    path = os.path.join(HERE, 'gat_wide.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.2f MB' % (os.path.getsize(path) / 1e6), 'losses', losses, 'sampled', counts[0],
          'edges', int(out['n_edges']), '%.0f s' % (time.time() - t0))


if __name__ == '__main__':
    main()
