"""Harness helpers with the reference's names and contracts (kgwas/utils.py:20-45,203-233,397-434)."""
from __future__ import annotations

import os
import pickle
import sys

import numpy as np
import torch


def print_sys(s):
    """kgwas/utils.py:227-233."""
    print(s, flush=True, file=sys.stderr)


def save_dict(path, obj):
    with open(path, 'wb') as f:
        pickle.dump(obj, f)


def load_dict(path):
    with open(path, 'rb') as f:
        return pickle.load(f)


def evaluate_minibatch_clean(loader, model, device):
    """kgwas/utils.py:20-39: eval-mode forward over a loader; returns {'pred', 'truth'} numpy arrays.
    Differences by design: wrapped in no_grad (the reference builds and frees autograd graphs), predictions stay on
    the GPU until the end (one D2H copy instead of one per batch), and -- for this package's own NeighborLoader on a
    ROCm device -- the per-batch forward is a captured HIP graph (kgwas_amd/graph_step.py::GraphEvalStep; set
    KGW_EVAL_EAGER=1 to issue the launches one by one instead: same kernels, same values)."""
    model.eval()
    from .sampler import NeighborLoader
    if isinstance(loader, NeighborLoader) and torch.device(device).type == 'cuda' and \
            os.environ.get('KGW_EVAL_EAGER', '0') != '1' and len(loader) > 0:
        cache = loader.__dict__.setdefault('_graph_eval', {})
        ge = cache.get(id(model))
        if ge is None:
            try:
                from .graph_step import GraphEvalStep
                # (a drop_last loader -- the reference's val loader, kgwas.py:102-103 -- never sees its tail)
                ids = loader.ids[:len(loader) * loader.batch_size] if loader.drop_last else loader.ids
                ge = GraphEvalStep(model, loader.data, loader.num_layers, (loader.input_type, ids), loader.batch_size, device)
                ge.eval_ids = ids
            except ValueError:
                ge = False                       # too few nodes for a padded static layout: eager path below
            cache.clear()
            cache[id(model)] = ge
        if ge:
            pred = ge.run()
            truth = loader.dg.y[loader.input_type][ge.eval_ids]
            return {'pred': pred.float().cpu().numpy(), 'truth': truth.float().cpu().numpy()}
    preds, truths = [], []
    with torch.no_grad():
        for batch in loader:
            batch = batch.to(device)
            bs = batch['SNP'].batch_size
            out = model(batch.x_dict, batch.edge_index_dict, bs)
            preds.append(out.reshape(-1))
            truths.append(batch['SNP'].y[:bs])
    if not preds:
        return {'pred': np.zeros(0, np.float32), 'truth': np.zeros(0, np.float32)}
    return {'pred': torch.cat(preds).float().cpu().numpy(), 'truth': torch.cat(truths).float().cpu().numpy()}


def compute_metrics(results, binary=False, coverage=None, uncertainty_reg=1, loss_fct=None):
    """kgwas/utils.py:41-45: sklearn mean_squared_error + scipy pearsonr."""
    from scipy.stats import pearsonr
    from sklearn.metrics import mean_squared_error
    return {'mse': mean_squared_error(results['pred'], results['truth']),
            'pearsonr': pearsonr(results['pred'], results['truth'])[0]}


def save_model(model, config, path_dir):
    """kgwas/utils.py:203-207: model.pt (state_dict) + config.pkl."""
    os.makedirs(path_dir, exist_ok=True)
    torch.save(model.state_dict(), os.path.join(path_dir, 'model.pt'))
    save_dict(os.path.join(path_dir, 'config.pkl'), config)


def load_pretrained(path, model):
    """kgwas/utils.py:209-222 (strips a DataParallel ``module.`` prefix)."""
    state_dict = torch.load(os.path.join(path, 'model.pt'), map_location=torch.device('cpu'), weights_only=False)
    if next(iter(state_dict))[:7] == 'module.':
        from collections import OrderedDict
        state_dict = OrderedDict((k[7:], v) for k, v in state_dict.items())
    model.load_state_dict(state_dict)
    return model


def ldsc_regression_weights(ld, w_ld, N, M, hsq, intercept=None, ii=None):
    """LD-score regression weights: the ten-line formula of kgwas/utils.py:397-434, TRANSCRIBED (same name, arguments and
    order of operations, so the values match the reference's bit for bit -- tests/golden/ref_helpers.npz):
    1 / (2 (intercept + hsq N ld / M)^2 w_ld) with ld, w_ld floored at 1 and hsq clipped to [0, 1]."""
    M = float(M)
    if intercept is None:
        intercept = 1
    hsq = min(max(hsq, 0.0), 1.0)
    ld = np.fmax(ld, 1.0)
    w_ld = np.fmax(w_ld, 1.0)
    c = hsq * N / M
    het_w = 1.0 / (2 * np.square(intercept + np.multiply(c, ld)))
    oc_w = 1.0 / w_ld
    return np.multiply(het_w, oc_w)


def get_network_weight(run, data):
    """kgwas/utils.py:437-494: raw attention weight of every edge of the knowledge graph at every layer, from the
    best model, as a DataFrame [h_idx, t_idx, weight, h_type, rel_type, t_type, layer] (duplicates of
    (h_idx, t_idx, rel_type, layer) dropped).  The reference runs its modified HeteroConv on the CPU; here the
    whole-graph pass runs on the fused kernels (HeteroGNN.raw_attention_full_graph)."""
    import pandas as pd
    model = getattr(run, 'best_model', None) or run.model
    print('Retrieving weights...')
    was_training = model.training
    model.eval()
    layers = model.raw_attention_full_graph(data.data, run.device)
    model.train(was_training)
    print('Aggregating across node types...')
    frames = []
    node_types = list(data.data.node_types)
    for k, att in enumerate(layers):
        for node_type in node_types:                                     # utils.py:469-474 iteration order
            for et, (ei, w) in att.items():
                if et[2] != node_type:
                    continue
                ei = ei.cpu().numpy()
                df = pd.DataFrame({'h_idx': ei[0].astype(np.float64), 't_idx': ei[1].astype(np.float64),
                                   'weight': w.cpu().numpy().astype(np.float64)})
                df['h_type'], df['rel_type'], df['t_type'], df['layer'] = et[0], et[1], et[2], f'l{k + 1}'
                frames.append(df)
    cols = ['h_idx', 't_idx', 'weight', 'h_type', 'rel_type', 't_type', 'layer']
    df_all = pd.concat(frames)[cols] if frames else pd.DataFrame(columns=cols)
    return df_all.drop_duplicates(['h_idx', 't_idx', 'rel_type', 'layer'])
