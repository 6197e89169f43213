"""The disease-critical network / variant interpretation (kgwas_amd.utils.generate_viz; SURVEY.md 8 row f-2, second half) against
the frames the REFERENCE'S OWN generate_viz (kgwas/utils.py:523-724) returned for the inputs of tests/golden/viz_case.py --
tests/golden/viz_network.npz, written by tests/golden/make_viz_golden.py in the build container.  Row order inside the frames is
not part of the contract (the reference sorts with an unstable sort and concatenates pool results): frames are compared as
multisets of rows, the interpretation per query SNP."""
import os
import pickle

import numpy as np
import pandas as pd
import pytest

from tests.golden import viz_case as vc

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS_NET = ['Category', 'h_idx', 't_idx', 'rel_type', 'h_type', 't_type', 'h_id', 't_id']
KEYS_VAR = ['QUERY_SNP', 'h_type', 't_type', 'h_idx', 't_idx', 'rel_type', 'h_id', 't_id']


class _Obj:
    pass


def _golden(tag, cols):
    G = np.load(os.path.join(HERE, 'golden', 'viz_network.npz'), allow_pickle=False)
    return pd.DataFrame({c: G[f'{tag}_{c}'] for c in cols})


def _run():
    run = _Obj()
    run.kgwas_res = vc.gwas()
    run.data = _Obj()
    run.data.idx2id, run.data.id2idx = vc.id_maps()
    return run


def _same_rows(mine, ref, keys):
    assert len(mine) == len(ref), (len(mine), len(ref))
    norm = lambda df: df.assign(**{k: df[k].astype(np.float64) if k in ('h_idx', 't_idx') else df[k].astype(str) for k in keys})
    a, b = norm(mine), norm(ref)
    a = a.sort_values(keys + ['importance'], kind='stable').reset_index(drop=True)
    b = b.sort_values(keys + ['importance'], kind='stable').reset_index(drop=True)
    for k in keys:
        assert (a[k].to_numpy() == b[k].to_numpy()).all(), k
    np.testing.assert_allclose(a['importance'].to_numpy(np.float64), b['importance'].to_numpy(np.float64), rtol=1e-12, atol=1e-12)


def _generate(tmp_path, with_names=True):
    from kgwas_amd.utils import generate_viz
    if with_names:
        os.makedirs(tmp_path / 'misc_data', exist_ok=True)
        with open(tmp_path / 'misc_data' / 'go2name.pkl', 'wb') as f:
            pickle.dump(vc.go2name(), f)
    return generate_viz(_run(), vc.network(), str(tmp_path), K_neighbors=vc.K_NEIGHBORS)


def test_disease_critical_network_matches_the_reference_output(tmp_path):
    _, net = _generate(tmp_path)
    ref = _golden('net', KEYS_NET + ['importance'])
    assert list(net.columns) == ['h_idx', 't_idx', 'importance', 'h_type', 't_type', 'rel_type', 'Category', 'h_id', 't_id']
    assert set(net.Category) == {'V2G', 'G2G', 'G2P'} and not net.rel_type.isin(['TSS', 'rev_TSS']).any()
    _same_rows(net, ref, KEYS_NET)
    # the category blocks come in the reference's order, pairs ascending inside a block (groupby order)
    assert list(dict.fromkeys(net.Category)) == ['V2G', 'G2G', 'G2P']
    assert (net.Category.to_numpy() == ref.Category.to_numpy()).all()
    assert np.array_equal(net.h_idx.to_numpy(np.float64), ref.h_idx.to_numpy(np.float64)) and \
        np.array_equal(net.t_idx.to_numpy(np.float64), ref.t_idx.to_numpy(np.float64))


def test_variant_interpretation_matches_the_reference_output(tmp_path):
    var, _ = _generate(tmp_path)
    ref = _golden('var', KEYS_VAR + ['importance'])
    assert list(var.columns) == ['h_idx', 't_idx', 'importance', 'h_type', 't_type', 'rel_type', 'h_id', 't_id', 'QUERY_SNP']
    assert var.QUERY_SNP.nunique() == ref.QUERY_SNP.nunique() and len(ref) > 100
    _same_rows(var, ref, KEYS_VAR)
    # query SNPs in the order of the GWAS frame; the SNP whose P equals the threshold is not a hit; hits without a gene edge
    # are left out (the reference's bare except)
    assert list(dict.fromkeys(var.QUERY_SNP)) == list(dict.fromkeys(ref.QUERY_SNP.astype(str)))
    hits = vc.hit_snps()
    assert f'rs{hits[0]}' not in set(var.QUERY_SNP)
    # per query: at most K genes, each row block bounded by K per (gene, table)
    first = var[(var.h_type == 'Gene') & (var.t_type == 'SNP')]
    assert first.groupby('QUERY_SNP').size().max() <= vc.K_NEIGHBORS


def test_threshold_argument_names_and_refusals(tmp_path):
    from kgwas_amd.utils import generate_viz
    var, net = _generate(tmp_path, with_names=False)
    assert net[net.Category == 'G2P'].h_id.str.startswith('GO:').all()           # no go2name.pkl: ids stay
    # a stricter threshold than any P: no hits, empty frames, no error
    v0, n0 = generate_viz(_run(), vc.network(), str(tmp_path), variant_threshold=1e-12)
    assert len(v0) == 0 and (n0.Category == 'V2G').sum() == 0 and (n0.Category == 'G2G').sum() > 0
    with pytest.raises(NotImplementedError, match='MAGMA'):
        generate_viz(_run(), vc.network(), str(tmp_path), magma_path='/nonexistent/magma.genes.out')
