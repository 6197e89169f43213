"""The native result-table writer (kgwas_amd/csrc/host/kgw_tsv.cpp, utils.write_tsv) against what it replaces:
``DataFrame.to_csv(path, index=False, sep='\\t')`` of kgwas/kgwas.py:205-212 -- byte for byte."""
import os

import numpy as np
import pandas as pd
import pytest

from kgwas_amd import utils


def _frame(n, seed=0):
    rng = np.random.default_rng(seed)
    # magnitudes over the whole double range, exact integers, powers of ten around both of repr's switch points
    p = np.concatenate([rng.uniform(size=n // 2) ** 8, 10.0 ** rng.uniform(-320, 308, n - n // 2) * rng.choice([-1, 1], n - n // 2)])
    special = [0.0, -0.0, 1.0, 1e-4, 1e-5, 9.999e-5, 1e15, 1e16, 9.999999999999998e15, 123456789012345678.0, 5e-324, 1.7976931348623157e308,
               np.nan, np.inf, -np.inf, 0.1, 1 / 3, 2.5, 100.0, 1e22, 1e23, 0.30000000000000004]
    k = min(n, len(special))
    p[:k] = special[:k]
    f32 = rng.standard_normal(n).astype(np.float32)
    f32s = np.array([0, -0.0, 1e-5, 1e16, np.nan, 3.4028235e38, 1e-45, 16777216.0, 1e-4, -1e-4, 1.0000001e-4, 9.999999e-5, 2e-4, 1e-3], dtype=np.float32)
    f32[:min(n, len(f32s))] = f32s[:min(n, len(f32s))]
    return pd.DataFrame({'#CHROM': 1, 'ID': [f'rs{i}' for i in range(n)], 'P': p, 'N': 5000.0, 'POS': rng.integers(-2 ** 62, 2 ** 62, n),
                         'small': rng.integers(0, 100, n).astype(np.int32), 'flag': rng.uniform(size=n) > 0.5, 'pred': f32,
                         'KGWAS_P': rng.uniform(size=n)})


def _same(df, tmp_path):
    a, b = str(tmp_path / 'pandas.tsv'), str(tmp_path / 'native.tsv')
    df.to_csv(a, index=False, sep='\t')
    utils.write_tsv(df, b)
    return open(a, 'rb').read() == open(b, 'rb').read()


def test_native_writer_is_built_and_loads():
    assert utils._host_lib(), 'libkgwas_host.so missing: python -m kgwas_amd.build'


@pytest.mark.parametrize('n', [1, 7, 30, 5000, 200_000])
def test_same_bytes_as_pandas(tmp_path, n):
    assert _same(_frame(n, seed=n), tmp_path)


def test_fallbacks_keep_the_output(tmp_path):
    df = _frame(50)
    for bad in ('has\ttab', 'has"quote', 'line\nbreak', '', 'ünï'):
        d2 = df.copy()
        d2.loc[3, 'ID'] = bad
        assert _same(d2, tmp_path), bad
    d3 = df.copy()
    d3['cat'] = pd.Categorical(['a', 'b'] * 25)
    d3['when'] = pd.Timestamp('2024-01-01')
    assert _same(d3, tmp_path)
    assert _same(df.iloc[:0], tmp_path)
    # duplicate column names, nullable integers with a missing value
    d4 = pd.concat([df[['P']], df[['N']].rename(columns={'N': 'P'})], axis=1)
    assert _same(d4, tmp_path)
    d5 = df.copy()
    d5['nullable'] = pd.array([1, None] * 25, dtype='Int64')
    assert _same(d5, tmp_path)
