"""Prediction-weighted p-values: the post-processing KGWAS.train() ends with
(kgwas/kgwas.py:191-212).  Restates kgwas/eval_utils.py:11-28 (``find_closest_x``) and :509-596
(``storey_pi_estimator`` / ``storey_ribshirani_integrate``) with the per-bin work vectorised; checked against
golden vectors produced by the reference functions themselves (tests/golden/make_golden.py).

CPU statistics on ~0.5 M rows, once per run -- not a GPU kernel and not on the timed path."""
from __future__ import annotations

import numpy as np
import pandas as pd
from scipy import interpolate


def find_closest_x(df_pred, lower_bound=0, upper_bound=200, tolerance=0.01):
    """kgwas/eval_utils.py:11-28: bisection on the scale s such that #{1e-3 < s*P_weighted < 1e-2} matches
    #{1e-3 < P < 1e-2}."""
    upper, lower = 1e-2, 1e-3
    pw = np.asarray(df_pred.P_weighted.values, dtype=np.float64)
    p = np.asarray(df_pred.P.values, dtype=np.float64)
    res2 = int(np.count_nonzero((p < upper) & (p > lower)))
    mid = (lower_bound + upper_bound) / 2
    while lower_bound <= upper_bound:
        mid = (lower_bound + upper_bound) / 2
        res1 = int(np.count_nonzero((pw * mid < upper) & (pw * mid > lower)))
        result = res1 / res2
        if abs(result - 1) < tolerance:
            return mid
        elif result > 1:
            lower_bound = mid + tolerance
        else:
            upper_bound = mid - tolerance
    return mid


_LAM = np.arange(0.05, 0.95, 0.05)


def storey_pi_estimator(pvalue: np.ndarray) -> float:
    """Storey & Tibshirani (PNAS 2003) pi0 of one bin, kgwas/eval_utils.py:509-536: pi0(lambda) on the grid
    0.05..0.90, cubic spline, evaluated at the last lambda, capped at 1."""
    total = float(len(pvalue))
    counts = np.array([(pvalue > l).sum() for l in _LAM])
    pi0 = counts / (total * (1 - _LAM))
    lam = _LAM
    if not np.all(np.isfinite(pi0)):
        keep = np.isfinite(pi0)
        lam, pi0 = lam[keep], pi0[keep]
    est = float(interpolate.CubicSpline(lam, pi0)(lam[-1]))
    return 1.0 if est > 1 else est


def storey_ribshirani_integrate(gwas_data: pd.DataFrame, column='pred', num_bins=100) -> np.ndarray:
    """kgwas/eval_utils.py:539-596: bin SNPs by quantiles of ``column``, estimate pi0 per bin, weight
    w = (1-pi0)/pi0 normalised to mean 1, P_weighted = P / w (original P kept where that exceeds 1).
    Adds the same columns to ``gwas_data`` as the reference (bin_number, pi0, weights, P_weighted)."""
    num_bins = float(num_bins)
    quantiles = np.arange(0, 1 + 1 / (num_bins + 1), 1 / num_bins)
    q = gwas_data[column].quantile(quantiles)
    # expand the outer edges so every value falls inside.  Label-based like the reference
    # (eval_utils.py:545-546): the Series is indexed by the quantile level, so [0] is the 0.0 edge and
    # [1] the 1.0 edge (the LAST one), not the second entry.
    q[0] = q[0] - 1
    q[1] = q[1] + 1
    q = q.drop_duplicates()
    nb = len(q) - 1
    bins = pd.cut(gwas_data[column], q, labels=np.arange(nb))
    gwas_data['bin_number'] = bins
    if (gwas_data['P'].min() < 0) or (gwas_data['P'].max() > 1):
        print('detected p-values < 0 or > 1, please double check. we clipped it to 0-1 for now...')
        gwas_data['P'] = gwas_data['P'].clip(lower=0, upper=1)
    codes = np.asarray(bins.cat.codes if hasattr(bins, 'cat') else bins, dtype=np.int64)      # -1 = outside
    p = gwas_data['P'].to_numpy(dtype=np.float64)
    pi0 = np.full(len(gwas_data), np.nan)
    order = np.argsort(codes, kind='stable')
    sc = codes[order]
    starts = np.searchsorted(sc, np.arange(nb), side='left')
    ends = np.searchsorted(sc, np.arange(nb), side='right')
    for i in range(nb):
        if ends[i] > starts[i]:
            idx = order[starts[i]:ends[i]]
            v = storey_pi_estimator(p[idx])
            pi0[idx] = min(max(v, 1e-5), 1 - 1e-5)          # prevent exploding weights
    gwas_data['pi0'] = pi0
    weights = (1 - pi0) / pi0
    weights = weights / np.nanmean(weights)
    gwas_data['weights'] = weights
    pw = p / weights
    over = pw > 1
    pw[over] = p[over]                                        # keep the original p-value above 1
    pw[np.isnan(pw)] = 1.0
    gwas_data['P_weighted'] = pw
    return gwas_data['P_weighted'].values
