#!/usr/bin/env python
"""Generate tests/golden/ref_helpers.npz by IMPORTING the reference's own helpers from
/root/reference (works only in the build container; the GPU box never sees the reference).

The reference's hot-path modules (conv.py, model.py, kgwas.py) import torch_geometric and cannot be
imported here; utils.py / eval_utils.py can, once kgwas/__init__.py is bypassed.  The
fixture is DATA only: inputs and the outputs the reference functions returned for them.

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    pkg = types.ModuleType('kgwas')
    pkg.__path__ = ['/root/reference/kgwas']
    sys.modules['kgwas'] = pkg
    return importlib.import_module('kgwas.utils')


class _Store(dict):
    __getattr__ = dict.__getitem__


class _Batch:
    def __init__(self, x, y, bs):
        self.x_dict = {'SNP': x}
        self.edge_index_dict = {}
        self._s = _Store(batch_size=bs, y=y)

    def to(self, device):
        return self

    def __getitem__(self, k):
        return self._s


class _Model(torch.nn.Module):
    def forward(self, x_dict, edge_index_dict, bs):
        return (x_dict['SNP'][:bs] * torch.tensor([0.5, -1.0, 2.0])).sum(1, keepdim=True).relu()


def main():
    utils = import_reference()
    rng = np.random.default_rng(20250307)
    out = {}
    # 1. LD-score regression weights (kgwas/utils.py:397-434) as called at kgwas_data.py:424
    ld = rng.uniform(0.2, 250.0, 512)
    w_ld = 1.0 + rng.uniform(0.0, 12.0, 512)
    w_ld[:5] = [0.1, 0.5, 1.0, 2.0, 0.0]
    for tag, n in (('n5000', 5000.0), ('n387113', 387113.0)):
        out[f'ldsc_w_{tag}'] = np.asarray(utils.ldsc_regression_weights(ld, w_ld, n, 15000000, 0.5))
    out['ldsc_ld'], out['ldsc_wld'] = ld, w_ld
    out['ldsc_w_hsq_clip'] = np.asarray(utils.ldsc_regression_weights(ld, w_ld, 5000.0, 15000000, 1.7))
    # 2. metrics (kgwas/utils.py:41-45)
    pred = rng.standard_normal(1000).astype(np.float32)
    truth = (0.3 * pred + rng.standard_normal(1000)).astype(np.float32)
    m = utils.compute_metrics({'pred': pred, 'truth': truth}, False, -1, -1, None)
    out['met_pred'], out['met_truth'] = pred, truth
    out['met_mse'], out['met_pearsonr'] = np.float64(m['mse']), np.float64(m['pearsonr'])
    # 3. evaluation harness contract (kgwas/utils.py:20-39) with a duck-typed loader / model
    g = torch.Generator().manual_seed(3)
    batches = []
    xs, ys = [], []
    for bs in (8, 8, 5):
        x = torch.randn(bs + 6, 3, generator=g)      # 6 non-seed rows follow the seeds
        y = torch.randn(bs + 6, generator=g)
        batches.append(_Batch(x, y, bs)); xs.append(x.numpy()); ys.append(y.numpy())
    res = utils.evaluate_minibatch_clean(batches, _Model(), 'cpu')
    out['eval_pred'], out['eval_truth'] = res['pred'], res['truth']
    for i, (x, y) in enumerate(zip(xs, ys)):
        out[f'eval_x{i}'], out[f'eval_y{i}'] = x, y
    out['eval_bs'] = np.array([8, 8, 5])
    # 4. split sizes at reference scale (kgwas_data.py:525-526; demo/kgwas_101.ipynb:58-59,352-357)
    from sklearn.model_selection import train_test_split
    ids = np.arange(542758)
    tv, te = train_test_split(ids, test_size=0.05, random_state=1)
    tr, va = train_test_split(tv, test_size=0.05, random_state=1)
    out['split_sizes'] = np.array([len(tr), len(va), len(te)])
    out['split_train_head'] = tr[:8]
    # 5. prediction-weighted p-values (kgwas/eval_utils.py:539-596, 11-28) as driven by kgwas.py:194-203
    eu = importlib.import_module('kgwas.eval_utils')
    import pandas as pd
    n = 20000
    pr = rng.standard_normal(n)
    pv = rng.uniform(0, 1, n)
    causal = rng.choice(n, 1500, replace=False)
    pr[causal] += 2.0                                    # informative predictions: signal SNPs have small P
    pv[causal] = pv[causal] ** 4
    for tag, nb in (('b50', 50), ('b500', 500)):
        df = pd.DataFrame({'P': pv.copy(), 'abs_pred': np.abs(pr)})
        pw = eu.storey_ribshirani_integrate(df, column='abs_pred', num_bins=nb)
        out[f'sr_pw_{tag}'] = np.asarray(pw, dtype=np.float64)
        out[f'sr_scale_{tag}'] = np.float64(eu.find_closest_x(pd.DataFrame({'P': pv, 'P_weighted': pw})))
    out['sr_P'], out['sr_abs_pred'] = pv, np.abs(pr)
    np.savez_compressed(os.path.join(HERE, 'ref_helpers.npz'), **out)
    print('wrote', os.path.join(HERE, 'ref_helpers.npz'), {k: np.shape(v) for k, v in out.items()})


if __name__ == '__main__':
    main()
