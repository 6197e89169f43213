#!/usr/bin/env python
"""Run the case written by tools/dump_for_pyg.py through the REAL reference model (snap-stanford/KGWAS ``kgwas.model.
HeteroGNN`` on torch_geometric) and compare with what this repo computed.  Needs torch_geometric and the reference on
PYTHONPATH -- i.e. NOT the build container; meant for a maintainer's machine.

    python tools/check_with_pyg.py pyg_check_case.pt"""
import sys

import torch


def main():
    from torch_geometric.data import HeteroData
    from kgwas.model import HeteroGNN                      # the reference (kgwas/model.py:24)
    d = torch.load(sys.argv[1], weights_only=False)
    data = HeteroData()
    for t, x in d['x_dict'].items():
        data[t].x = x
    for et, ei in d['edge_index_dict'].items():
        data[et].edge_index = ei
    dims = d['dims']
    model = HeteroGNN(data, d['hidden'], 1, d['num_layers'], 'GAT', 'sum', dims['SNP'], dims['Gene'], dims['GO'], 1)
    with torch.no_grad():                                   # materialise the lazy (-1,-1) Linears (conv.py:82-89)
        model(dict(d['x_dict']), d['edge_index_dict'], d['batch_size'])
    own = model.state_dict()

    def norm(k):                                            # PyG >= 2.4 writes '<src___rel___dst>', older 'src__rel__dst'
        return k.replace('<', '').replace('>', '').replace('___', '__')
    by_norm = {norm(k): k for k in own}
    new = {}
    for k, v in d['state_dict'].items():
        assert norm(k) in by_norm, f'parameter {k} not in the reference model'
        new[by_norm[norm(k)]] = v.reshape(own[by_norm[norm(k)]].shape) if not isinstance(
            own[by_norm[norm(k)]], torch.nn.parameter.UninitializedParameter) else v
    missing = [k for k in own if k not in new and 'lin_dst' not in k]
    assert not missing, missing
    model.load_state_dict(new, strict=False)
    model = model.double()
    x = {t: v.double() for t, v in d['x_dict'].items()}
    pred = model(x, d['edge_index_dict'], d['batch_size']).reshape(-1)
    loss = torch.mean(d['ld_weight'] * (pred - d['y'].double()) ** 2)          # kgwas/kgwas.py:145
    loss.backward()
    exp = d['expected']
    print('max |pred - expected|', float((pred.detach() - exp['pred'].reshape(-1)).abs().max()))
    print('loss', float(loss), 'expected', float(exp['loss']))
    worst = 0.0
    for n, p in model.named_parameters():
        g = exp['grads'].get(norm(n))
        if g is None or p.grad is None:
            assert (g is None or float(g.abs().max()) == 0) and (p.grad is None or float(p.grad.abs().max()) == 0), n
            continue
        worst = max(worst, float((p.grad - g.reshape(p.grad.shape)).abs().max()) / (float(g.abs().max()) + 1e-30))
    print('max relative gradient difference', worst)
    ok = float((pred.detach() - exp['pred'].reshape(-1)).abs().max()) < 1e-9 and worst < 1e-8
    print('PARITY WITH PyG:', 'OK' if ok else 'MISMATCH')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
