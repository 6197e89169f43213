#!/usr/bin/env python
"""Write a self-contained case for an OFFLINE cross-check against real PyG + the reference (neither is installable in
the build container, so the oracle's GAT core is otherwise pinned only by the second derivation in oracle/dense_gat.py).

    python tools/dump_for_pyg.py [out.pt]          # here: inputs + state_dict + what this repo computes for them
    python tools/check_with_pyg.py out.pt          # on a machine with torch_geometric and snap-stanford/KGWAS installed

The file holds the tiny case of tests/golden/gat_case.py as a sampled batch in PyG's node order: ``x_dict``,
``edge_index_dict`` (int64 [2,E] per edge type, local ids), ``batch_size``, ``state_dict`` under the reference's
parameter names (kgwas/model.py:30-50, conv.py:86-108; both PyG edge-type key styles are accepted by the checker),
labels / LD weights of the seeds, and this repo's float64 results: prediction, loss, every parameter gradient."""
import os
import sys
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle.gat_oracle import weighted_mse
    from tests.golden import gat_case as gc
    from tests.golden_io import build_oracle, sampled_inputs
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'pyg_check_case.pt')
    x, ei, n_id, yb, wb = sampled_inputs()
    edge_types = list(ei.keys())
    oracle = build_oracle(edge_types)
    pred = oracle(x, ei, gc.BATCH)
    loss = weighted_mse(pred, yb, wb)
    loss.backward()
    d = {'x_dict': {t: v.float() for t, v in x.items()},
         'edge_index_dict': OrderedDict((et, e.long()) for et, e in ei.items()),
         'n_id': n_id, 'batch_size': gc.BATCH, 'y': yb.float(), 'ld_weight': wb,
         'dims': dict(gc.DIMS), 'hidden': gc.HIDDEN, 'num_layers': gc.NUM_LAYERS,
         'state_dict': OrderedDict((k, torch.from_numpy(v)) for k, v in gc.parameters(edge_types).items()),
         'expected': {'pred': pred.detach(), 'loss': loss.detach(),
                      'grads': {n: (p.grad.clone() if p.grad is not None else None) for n, p in oracle.named_parameters()}}}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    torch.save(d, out)
    print('wrote', out, 'pred[:4]', pred.detach().reshape(-1)[:4].tolist(), 'loss', float(loss.detach()))


if __name__ == '__main__':
    main()
