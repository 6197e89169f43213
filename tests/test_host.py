"""CPU: host-side logic of the product and the C ABI surface (no kernel launches without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from kgwas_amd import _lib
    lib = _lib.lib()
    hdr = open(os.path.join(ROOT, 'include', 'kgwas_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(kgw_[a-z_0-9]+)\s*\(', hdr)))
    assert len(declared) >= 9
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in kgwas_hip.h but not exported'
    assert sorted(_lib.EXPORTS) == declared
    ver = int(re.search(r'#define KGW_VERSION\s+(\d+)', hdr).group(1))
    assert lib.kgw_version() == ver == 125
    assert lib.kgw_status_string(-1) == b'null pointer argument'
    # the binding sketch a maintainer of the reference would start from names the same ABI, and every entry point of the header has
    # its row in INTEGRATION.md's table (the sketch had gone stale)
    integ = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    assert f'kgw_version() == {ver}' in integ
    missing = [n for n in declared if f'`{n}`' not in integ and f'`{n}(' not in integ]
    assert not missing, missing


def test_sampler_scratch_size_covers_every_sort_plan():
    """kgw_sampler_scan_ints (the size KgwBatchBuf.scan_tmp must have): at least the hops' scan scratch, and at least the
    [2 layers][buckets + 1][512 blocks] counts + bucket starts of the src-major radix sort for ANY row count up to trow_cap
    (static capacities below trow_cap use finer buckets: up to 4 000 of them, never fewer than 256 rows each)."""
    from kgwas_amd import _lib
    lib = _lib.lib()
    for trow_cap in (0, 100, 5_000, 1_087_035, 5_000_000, 60_000_000, 140_000_000):
        n = int(lib.kgw_sampler_scan_ints(10_000, 900_000, trow_cap))
        assert n >= 2 * (900_000 // _lib.KGW_TILE + 4)
        for rows in {trow_cap, trow_cap // 2, trow_cap // 5, min(trow_cap, 1_100_000)}:
            sh = 8
            while sh < 14 and (rows >> sh) + 1 > 4000:
                sh += 1
            nb = (rows >> sh) + 1
            if nb <= 9000:
                assert n >= 2 * (nb + 1) * 512 + 2 * (nb + 2), (trow_cap, rows, nb)
    assert int(lib.kgw_sampler_scan_ints(50_000_000, 10, 10)) >= 2 * (50_000_000 // _lib.KGW_TILE + 4)


def test_abi_struct_sizes_and_argument_checks():
    from kgwas_amd import _lib
    lib = _lib.lib()
    sizes = (C.c_int64 * 7)()
    assert lib.kgw_struct_sizes(sizes, 7) == 0
    assert list(sizes) == [C.sizeof(_lib.KgwGraph), C.sizeof(_lib.KgwBatchMeta), C.sizeof(_lib.KgwChunk),
                           C.sizeof(_lib.KgwBatchBuf), C.sizeof(_lib.KgwLayerArgs), C.sizeof(_lib.KgwTnJob),
                           C.sizeof(_lib.KgwGradSrc)]
    assert C.sizeof(_lib.KgwGradSrc) == 64
    assert C.sizeof(_lib.KgwChunk) == 32
    # argument errors are reported as negative status codes before any launch
    assert lib.kgw_sample_batch(None, None, None, 0, 0, 0, None) == -1
    assert lib.kgw_gat_aggregate_fwd(None, None) == -1
    assert lib.kgw_gather_rows(None, None, 5, 4, None, None) == -1
    g = _lib.KgwGraph()
    b = _lib.KgwBatchBuf()
    assert lib.kgw_sample_batch(C.byref(g), C.byref(b), None, 0, 0, 1, None) == -2     # n_types = 0 out of range
    # the multi-job entry points: nothing to do / null tables / too many jobs / odd shapes -- all before any launch
    assert lib.kgw_gather_rows_multi(0, None, None, None, 4, None, None) == 0
    assert lib.kgw_gather_rows_multi(2, None, None, None, 4, None, None) == -1
    P2, N2 = (C.c_void_p * 9)(), (C.c_int64 * 9)()
    assert lib.kgw_gather_rows_multi(9, P2, P2, N2, 4, P2, None) == -2
    assert lib.kgw_tn_gemm_multi(0, None, None) == 0
    assert lib.kgw_tn_gemm_multi(2, None, None) == -1
    jobs = (_lib.KgwTnJob * 5)()
    assert lib.kgw_tn_gemm_multi(5, jobs, None) == -2
    assert lib.kgw_tn_gemm_multi(1, jobs, None) == -1                                   # null operands
    # the fused optimiser launch and the partial products that feed it: argument checks before any launch
    assert lib.kgw_tn_gemm_multi_partial(2, jobs, None, None) == -1
    assert lib.kgw_tn_gemm_partial(None, 0, 0, None, 0, 0, 0, None, 0, 0, None, None, 0, None, None, None) == -1
    assert lib.kgw_gemm3_partial(None, 0, 0, 0, None, None, 0, None, 0, None, None) == -1
    assert lib.kgw_gemm3_flip() in (0, 1, 2, 4, 8, 16, 32, 64)
    assert lib.kgw_adam_fused(0, None, None, None, None, None, None, None, 1e-4, 0.9, 0.999, 1e-8, 0.0, None, 0, 0, None, None, None) == -1
    assert lib.kgw_adam_fused(_lib.ADAM_FUSED_MAX + 1, None, None, None, None, None, None, None, 1e-4, 0.9, 0.999, 1e-8, 0.0, None, 0, 0, None,
                              None, None) == -2
    assert lib.kgw_grad_finish(3, None, None, None, None, None) == -1
    assert lib.kgw_grad_finish(_lib.ADAM_FUSED_MAX + 1, None, None, None, None, None) == -2
    assert lib.kgw_grad_finish(0, None, None, None, None, None) == 0
    # round 5's launch merges: range / null checks before any launch, the plan / fold records' sizes, the pipe and row-block knobs
    assert lib.kgw_param_tail(-1, None, None, None, 0, None, 0, None) == -2
    assert lib.kgw_param_tail(2, None, None, None, 0, None, 0, None) == -1
    assert lib.kgw_param_tail(0, None, None, None, 0, None, 0, None) == 0                 # nothing to do
    assert lib.kgw_tn_reduce_launch(None, None) == -1
    plan = _lib.KgwTnReducePlan()
    assert C.sizeof(plan) == 16 + 96 * 8 and lib.kgw_tn_reduce_launch(C.byref(plan), None) == 0      # valid == 0: nothing pending
    assert lib.kgw_transform_bwd_ex(0, None, 0, None, 0, None, C.byref(plan), None, None, None) == 0
    assert lib.kgw_transform_bwd_ex(5, None, 0, None, 0, None, None, None, None, None) == -2
    fold = _lib.KgwReadoutFold()
    assert C.sizeof(fold) == 48 and lib.kgw_readout_train_fold(None, None) == -1 and lib.kgw_readout_train_fold(C.byref(fold), None) == -1
    assert lib.kgw_transform_bwd_ex(0, None, 0, None, 0, None, None, None, C.byref(fold), None) == -1     # a fold record without pointers
    assert lib.kgw_mlp2_bwd_first_packed(None, 0, None, 0, None, 0, None, 0, 0, 64, None, 0, None, None, 0, None, None, 0, None, 16, None,
                                         None) == -1
    assert lib.kgw_tn_split(-1) in (0, 1) and lib.kgw_tn_direct_rows(-1) >= 0
    was = lib.kgw_tn_split(0)
    assert lib.kgw_tn_split(was) == 0 and lib.kgw_tn_split(-1) == was
    assert lib.kgw_scatter_relu_rows(None, None, None, 8, None, None, None, None) == -1
    assert lib.kgw_scatter_relu_rows_workspace_floats(20032) == 256 * 128
    assert lib.kgw_tn_gemm_workspace_floats(1000, 128, 128) > 0


def test_product_refuses_to_run_without_gpu():
    """No CPU fallback: a CPU device is an error, not a slow path."""
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from kgwas_amd import _lib
    from kgwas_amd.kgwas_data import KGWAS_Data
    from kgwas_amd.sampler import NeighborLoader
    d = KGWAS_Data.from_synthetic(scale=0.002, seed=3, feat_dims={'Gene': 40}, data_path='/tmp/kgwas_synth_tiny')
    with pytest.raises(_lib.KgwasHipError):
        NeighborLoader(d.data, [-1, -1], d.train_input_nodes, batch_size=8, device='cpu')


def test_graph_schema_and_liveness(tiny_kg):
    from kgwas_amd.graph import GraphSchema
    sc = GraphSchema(tiny_kg.data.node_types, tiny_kg.data.edge_types)
    assert sc.node_types[:2] == ['SNP', 'Gene'] and sc.NR == 29
    live, types = sc.live_relations(2)
    # layer 2: only relations into SNP; layer 1: relations into SNP or Gene
    assert {sc.edge_types[r][2] for r in live[2]} == {'SNP'}
    assert {sc.edge_types[r][2] for r in live[1]} == {'SNP', 'Gene'}
    assert len(live[2]) == 6 and len(live[1]) == 23
    assert types[2] == {sc.type_id['SNP']} and types[1] == {sc.type_id['SNP'], sc.type_id['Gene']}
    # slots: position of a relation among the relations sharing its destination / source type
    for t in range(sc.NT):
        assert [int(sc.slot_dst[r]) for r in sc.rels_by_dst[t]] == list(range(int(sc.R_dst[t])))
        assert [int(sc.slot_src[r]) for r in sc.rels_by_src[t]] == list(range(int(sc.R_src[t])))


def test_build_csr_is_dst_major_src_sorted():
    from kgwas_amd.graph import build_csr
    ei = np.array([[5, 1, 3, 1, 0, 5], [2, 2, 0, 2, 1, 0]])
    rp, col = build_csr(ei, 6, 3)
    assert rp.tolist() == [0, 2, 3, 6]
    assert col.tolist() == [3, 5, 0, 1, 1, 5]          # duplicates kept, sources ascending inside a row
    with pytest.raises(ValueError):
        build_csr(np.array([[7], [0]]), 6, 3)


def test_synthetic_data_pipeline(small_kg):
    d = small_kg
    g = d.data
    assert g.node_types == ['SNP', 'Gene', 'CellularComponent', 'BiologicalProcess', 'MolecularFunction']
    assert len(g.edge_types) == 29 and ('Gene', 'rev_TSS', 'SNP') in g.edge_types
    n = g['SNP'].x.shape[0]
    assert g['SNP'].y.shape == (n,) and float(g['SNP'].y.min()) == -1.0       # -1 = unlabelled (kgwas_data.py:532)
    tr, va, te = (np.asarray(x[1]) for x in (d.train_input_nodes, d.val_input_nodes, d.test_input_nodes))
    assert len(set(tr) | set(va) | set(te)) == len(tr) + len(va) + len(te) == len(d.all_ids)
    assert abs(np.mean(d.ldsc_weight) - 1.0) < 1e-9
    assert d.rs_id_to_ldsc_weight[d.lr_uni.ID.values[0]] == d.ldsc_weight[0]
    assert d.idx2id['SNP'][5] == 'rs5' and d.id2idx['SNP']['rs5'] == 5
    assert (d.snp_init_dim_size, d.gene_init_dim_size, d.go_init_dim_size) == (20, 96, 128)


def test_model_state_dict_uses_reference_keys(tiny_kg):
    from kgwas_amd.model import HeteroGNN
    m = HeteroGNN(tiny_kg.data, 128, 1, 2, 'GAT', 'sum', 20, 40, 128, 1)
    sd = m.state_dict()
    assert sd['convs.0.convs.SNP__ABC__Gene.lin_src.weight'].shape == (128, 128)
    assert sd['convs.1.convs.Gene__rev_ABC__SNP.att_src'].shape == (1, 1, 128)
    assert sd['convs.0.convs.SNP__ABC__Gene.bias'].shape == (128,)
    assert sd['snp_feat_mlp.FC_hidden.weight'].shape == (128, 20) and sd['lin.weight'].shape == (1, 128)
    # same-type relations: lin_dst is a never-materialised lazy parameter in the reference
    assert isinstance(sd['convs.0.convs.Gene__Gene-Reaction-Gene__Gene.lin_dst.weight'],
                      torch.nn.parameter.UninitializedParameter)
    m2 = HeteroGNN(tiny_kg.data, 128, 1, 2, 'GAT', 'sum', 20, 40, 128, 1)
    m2.load_state_dict(sd)
    a, b = m.named_reference_tensors(), m2.named_reference_tensors()
    assert all(torch.equal(a[k], b[k]) for k in a)
    # PyG >= 2.4 key style loads too
    m2.load_state_dict({k.replace('SNP__ABC__Gene', '<SNP___ABC___Gene>'): v for k, v in sd.items()})
    with pytest.raises(RuntimeError):
        m2.load_state_dict({k: v for k, v in sd.items() if 'lin.bias' not in k})
    # unsupported reference options fail loudly instead of silently doing something else
    for kw in (dict(gnn_backbone='GCN'), dict(gnn_backbone='SGC'), dict(gnn_aggr='cat'), dict(gat_num_head=2)):
        args = dict(gnn_backbone='GAT', gnn_aggr='sum', gat_num_head=1); args.update(kw)
        with pytest.raises(NotImplementedError):
            HeteroGNN(tiny_kg.data, 128, 1, 2, args['gnn_backbone'], args['gnn_aggr'], 20, 40, 128, args['gat_num_head'])
    # the SAGE backbone (kgwas/model.py:38) is built: its checkpoints carry PyG SAGEConv's keys
    ms = HeteroGNN(tiny_kg.data, 128, 1, 2, 'SAGE', 'sum', 20, 40, 128, 1)
    ks = list(ms.state_dict())
    assert 'convs.0.convs.SNP__ABC__Gene.lin_l.weight' in ks and 'convs.0.convs.SNP__ABC__Gene.lin_l.bias' in ks and \
        'convs.1.convs.SNP__ABC__Gene.lin_r.weight' in ks and not any('att_' in k for k in ks)
    ms2 = HeteroGNN(tiny_kg.data, 128, 1, 2, 'SAGE', 'sum', 20, 40, 128, 1)
    ms2.load_state_dict(ms.state_dict())
    assert all(torch.equal(a, b) for a, b in zip(ms.state_dict().values(), ms2.state_dict().values()))


def test_shard_batches_partitions_every_batch():
    from kgwas_amd.dist import shard_batches
    ids = np.arange(1000)
    parts = [shard_batches(ids, 64, r, 4) for r in range(4)]
    assert all(len(p) == 15 * 16 for p in parts)
    for b in range(15):
        got = np.concatenate([p[b * 16:(b + 1) * 16] for p in parts])
        assert np.array_equal(np.sort(got), ids[b * 64:(b + 1) * 64])
    with pytest.raises(ValueError):
        shard_batches(ids, 10, 0, 4)


def test_library_is_current_with_its_sources():
    """The .so the tests load was built from the sources in the tree (content hashes, kgwas_amd/build.py): a stale library
    would make every other test a statement about some other code."""
    from kgwas_amd import build as kb
    assert not kb.needs_build(), 'libkgwas_hip.so is older than its sources: python -c "import __graft_entry__ as g; g.build()"'


def test_bench_starts_its_own_ranks_when_no_launcher_did(monkeypatch):
    """`python bench.py --gpus N` (N > 1) with no WORLD_SIZE in the environment re-executes itself under
    torch.distributed.run with N ranks on 127.0.0.1 (the command form the driver uses) instead of refusing to start."""
    import argparse
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '7', '--warmup', '2'])
    assert bench.self_launch(argparse.Namespace(gpus=4)) == 0
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '4', '--steps', '7', '--warmup', '2']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_gene_shard_scope_is_not_process_global_state():
    """ops.GENE_SHARD used to be set by a trainer and never cleared.  It is now active only inside a trainer's own
    forward passes, bound to ONE (X, W), and a probe records the layer without any collective."""
    from kgwas_amd import ops

    class Ctx:
        needs_input_grad = (False, True)
    X, W, X2 = torch.zeros(4, 4), torch.zeros(2, 4), torch.zeros(4, 4)
    assert ops.GENE_SHARD is None and ops.active_gene_shard(Ctx, 1, X, W, None) is None

    class GS:
        last = None
    gs, seen = GS(), []
    with ops.gene_shard_scope(gs, seen):
        assert ops.active_gene_shard(Ctx, 1, X, W, None) is gs
        gs.last = (X, W, None)
        assert ops.active_gene_shard(Ctx, 1, X, W, None) is gs
        assert ops.active_gene_shard(Ctx, 1, X2, W, None) is None      # a second resident layer: computed locally
        with ops.gene_shard_scope(None):
            assert ops.active_gene_shard(Ctx, 1, X, W, None) is None
        assert ops.active_gene_shard(Ctx, 0, X, W, None) is None       # inference: no weight gradient asked
    assert ops.GENE_SHARD is None and ops.RESIDENT_SEEN is None
    assert len(seen) == 3 and seen[0][0] is X
