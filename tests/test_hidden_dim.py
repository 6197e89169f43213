"""gnn_hidden_dim (kgwas/kgwas.py:52): widths below the kernels' 128 are embedded zero-padded; the reference-named views
(state_dict, checkpoints, gradients) have the model's own width.  Host logic only (no GPU)."""
from collections import OrderedDict

import pytest
import torch


class _Schema:
    node_types = ['SNP', 'Gene', 'CellularComponent', 'BiologicalProcess', 'MolecularFunction']
    edge_types = [('SNP', 'ABC', 'Gene'), ('Gene', 'rev_ABC', 'SNP'), ('Gene', 'G2G', 'Gene'),
                  ('Gene', 'G-CC', 'CellularComponent'), ('CellularComponent', 'rev_G-CC', 'Gene')]


@pytest.mark.parametrize('backbone', ['GAT', 'SAGE'])
@pytest.mark.parametrize('hc', [128, 64, 20])
def test_state_dict_has_the_models_own_width_and_round_trips(hc, backbone):
    from kgwas_amd.model import HeteroGNN
    torch.manual_seed(0)
    m = HeteroGNN(_Schema, hc, 1, 2, backbone, 'sum', 20, 24, 16, 1)
    assert m.hidden == 128 and m.hidden_logical == hc
    sd = m.state_dict()
    assert sd['snp_feat_mlp.FC_hidden.weight'].shape == (hc, 20) and sd['gene_feat_mlp.FC_hidden2.weight'].shape == (hc, hc)
    assert sd['go_feat_mlp.FC_output.bias'].shape == (hc,) and sd['lin.weight'].shape == (1, hc)
    if backbone == 'GAT':
        assert sd['convs.0.convs.SNP__ABC__Gene.lin_src.weight'].shape == (hc, hc)
        assert sd['convs.1.convs.Gene__rev_ABC__SNP.att_dst'].shape == (1, 1, hc)
        assert sd['convs.0.convs.Gene__G2G__Gene.bias'].shape == (hc,)
    else:
        assert sd['convs.0.convs.SNP__ABC__Gene.lin_l.weight'].shape == (hc, hc)
    # everything outside the model's own block is exactly zero in the padded storage
    for p in m.parameters():
        if p.dim() == 3 and p.shape[-1] == 128:
            assert float(p[:, hc:, :].abs().sum()) == 0.0 and float(p[:, :, hc:].abs().sum()) == 0.0
    assert float(m.lin.weight[:, hc:].abs().sum()) == 0.0 and float(m.snp_feat_mlp.FC_hidden.weight[hc:].abs().sum()) == 0.0
    assert float(m.lin.weight[:, :hc].abs().sum()) > 0.0
    m2 = HeteroGNN(_Schema, hc, 1, 2, backbone, 'sum', 20, 24, 16, 1)
    res = m2.load_state_dict(sd)
    assert not res.missing_keys and not res.unexpected_keys
    for (n, a), (_, b) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), n
    if hc != 128:
        m3 = HeteroGNN(_Schema, 128, 1, 2, backbone, 'sum', 20, 24, 16, 1)
        with pytest.raises(RuntimeError):
            m3.load_state_dict(OrderedDict((k, v) for k, v in sd.items() if not isinstance(v, torch.nn.parameter.UninitializedParameter)))


def test_widths_above_128_are_refused():
    from kgwas_amd.model import HeteroGNN
    with pytest.raises(NotImplementedError, match='128'):
        HeteroGNN(_Schema, 256, 1, 2, 'GAT', 'sum', 20, 24, 16, 1)
