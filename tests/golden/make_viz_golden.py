#!/usr/bin/env python
"""Generate tests/golden/viz_network.npz by RUNNING the reference's own generate_viz (kgwas/utils.py:523-724, the second half
of KGWAS.get_disease_critical_network, kgwas/kgwas.py:268-273) in the build container on the inputs of tests/golden/viz_case.py.
Works only where /root/reference exists; the fixture is data (the two frames it returned, column by column).

Two accommodations, neither touching the reference's arithmetic:
  * the reference was written for pandas < 2 (DataFrame.append, used by get_local_interpretation at utils.py:503-515, was removed
    in pandas 2.0; under pandas 2 every call raises inside its bare `except` and the interpretation comes back EMPTY).  The
    generator restores DataFrame.append with its pandas-1 meaning (concatenate, keep the index) for the duration of the call;
  * misc_data/gene_set_bp.pkl and go2name.pkl are read unconditionally (utils.py:531-533): written to a temporary directory.

    python tests/golden/make_viz_golden.py
"""
import os
import pickle
import sys
import tempfile
import warnings

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import viz_case as vc                      # noqa: E402
from tests.golden.make_golden import import_reference        # noqa: E402

COLS_NET = ['h_idx', 't_idx', 'importance', 'h_type', 't_type', 'rel_type', 'Category', 'h_id', 't_id']
COLS_VAR = ['h_idx', 't_idx', 'importance', 'h_type', 't_type', 'rel_type', 'h_id', 't_id', 'QUERY_SNP']


class _Run:
    pass


def frame_to_arrays(df, cols, tag, out):
    out[f'{tag}_n'] = np.array(len(df))
    for c in cols:
        v = df[c].to_numpy() if len(df) else np.array([])
        out[f'{tag}_{c}'] = v.astype(np.float64) if c in ('h_idx', 't_idx', 'importance') else v.astype(str)


def main():
    utils = import_reference()
    if not hasattr(pd.DataFrame, 'append'):
        pd.DataFrame.append = lambda self, other, **kw: pd.concat([self, other])      # pandas-1 DataFrame.append
    run = _Run()
    run.kgwas_res = vc.gwas()
    run.data = _Run()
    run.data.idx2id, run.data.id2idx = vc.id_maps()
    out = {}
    with tempfile.TemporaryDirectory() as d, warnings.catch_warnings():
        warnings.simplefilter('ignore')
        os.makedirs(os.path.join(d, 'misc_data'))
        with open(os.path.join(d, 'misc_data', 'gene_set_bp.pkl'), 'wb') as f:
            pickle.dump({}, f)
        with open(os.path.join(d, 'misc_data', 'go2name.pkl'), 'wb') as f:
            pickle.dump(vc.go2name(), f)
        df_var, df_net = utils.generate_viz(run, vc.network(), d, K_neighbors=vc.K_NEIGHBORS, num_cpus=1)
    print('disease critical network rows', len(df_net), dict(df_net.Category.value_counts()))
    print('variant interpretation rows', len(df_var), 'query SNPs', df_var.QUERY_SNP.nunique() if len(df_var) else 0)
    print('NaN importance rows: network', int(df_net.importance.isna().sum()), 'interpretation', int(df_var.importance.isna().sum()))
    frame_to_arrays(df_net, COLS_NET, 'net', out)
    frame_to_arrays(df_var, COLS_VAR, 'var', out)
    np.savez_compressed(os.path.join(HERE, 'viz_network.npz'), **out)
    print('wrote viz_network.npz', {k: v.shape for k, v in list(out.items())[:4]})


if __name__ == '__main__':
    main()
