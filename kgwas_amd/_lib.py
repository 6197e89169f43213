"""ctypes binding of libkgwas_hip.so (include/kgwas_hip.h).  Fails loudly when the library is missing:
there is NO CPU fallback for the product path."""
from __future__ import annotations

import ctypes as C
import os

KGW_MAX_TYPES = 8
KGW_MAX_RELS = 64
KGW_MAX_LAYERS = 4
KGW_CHUNK = 128
KGW_TILE = 1024
KGW_C = 128
PART_STRIDE = 132

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('KGW_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libkgwas_hip.so')     # (override: A/B runs of two builds)


class KgwGraph(C.Structure):
    _fields_ = [
        ('n_types', C.c_int32), ('n_rels', C.c_int32), ('n_layers', C.c_int32), ('n_hops', C.c_int32),
        ('n_nodes', C.c_int32 * KGW_MAX_TYPES),
        ('node_base', C.c_int32 * (KGW_MAX_TYPES + 1)),
        ('R_dst', C.c_int32 * KGW_MAX_TYPES),
        ('R_src', C.c_int32 * KGW_MAX_TYPES),
        ('rel_src', C.c_int32 * KGW_MAX_RELS),
        ('rel_dst', C.c_int32 * KGW_MAX_RELS),
        ('rel_slot_dst', C.c_int32 * KGW_MAX_RELS),
        ('rel_slot_src', C.c_int32 * KGW_MAX_RELS),
        ('rowptr_off', C.c_int64 * KGW_MAX_RELS),
        ('col_off', C.c_int64 * KGW_MAX_RELS),
        ('rel_live', (C.c_uint8 * KGW_MAX_RELS) * KGW_MAX_LAYERS),
        ('static_layout', C.c_int32),
        ('cap_rows', (C.c_int32 * KGW_MAX_TYPES) * KGW_MAX_LAYERS),
        ('cap_src', (C.c_int32 * KGW_MAX_TYPES) * KGW_MAX_LAYERS),
        ('short_types', C.c_uint32),
        ('g_rowptr', C.c_void_p),
        ('g_col', C.c_void_p),
    ]


class KgwBatchMeta(C.Structure):
    _fields_ = [
        ('hop_cnt', (C.c_int32 * (KGW_MAX_LAYERS + 1)) * KGW_MAX_TYPES),
        ('node_off', (C.c_int32 * (KGW_MAX_LAYERS + 2)) * KGW_MAX_TYPES),
        ('seg_off', (C.c_int32 * (KGW_MAX_RELS + 1)) * KGW_MAX_LAYERS),
        ('seg_end', C.c_int32 * KGW_MAX_LAYERS),
        ('edge_end', C.c_int32 * KGW_MAX_LAYERS),
        ('chunk_end', C.c_int32 * KGW_MAX_LAYERS),
        ('multi_cnt', C.c_int32 * KGW_MAX_LAYERS),
        ('n_rows', (C.c_int32 * KGW_MAX_TYPES) * KGW_MAX_LAYERS),
        ('z_base', (C.c_int32 * (KGW_MAX_TYPES + 1)) * KGW_MAX_LAYERS),
        ('n_src', (C.c_int32 * KGW_MAX_TYPES) * KGW_MAX_LAYERS),
        ('src_base', (C.c_int32 * (KGW_MAX_TYPES + 1)) * KGW_MAX_LAYERS),
        ('t_base', (C.c_int32 * (KGW_MAX_TYPES + 1)) * KGW_MAX_LAYERS),
        ('lay_rows', (C.c_int32 * KGW_MAX_TYPES) * KGW_MAX_LAYERS),
        ('lay_src', (C.c_int32 * KGW_MAX_TYPES) * KGW_MAX_LAYERS),
        ('n_chunks', C.c_int32 * KGW_MAX_LAYERS),
        ('n_edges', C.c_int32 * KGW_MAX_LAYERS),
        ('t_entries', C.c_int32 * KGW_MAX_LAYERS),
        ('cur', C.c_int32 * 8),
        ('error', C.c_int32),
        ('pad_', C.c_int32 * 3),
    ]


class KgwChunk(C.Structure):
    _fields_ = [('e0', C.c_int32), ('e1', C.c_int32), ('row', C.c_int32), ('rel', C.c_int32),
                ('first', C.c_int32), ('nch', C.c_int32), ('gpos_lo', C.c_int32), ('gpos_hi', C.c_int32)]


class KgwBatchBuf(C.Structure):
    _fields_ = [
        ('g2l', C.c_void_p), ('n_id', C.c_void_p), ('seg_deg', C.c_void_p), ('seg_nch', C.c_void_p),
        ('seg_ptr', C.c_void_p), ('seg_chptr', C.c_void_p), ('col_local', C.c_void_p),
        ('chunks', C.c_void_p), ('multi', C.c_void_p),
        ('t_cnt', C.c_void_p * KGW_MAX_LAYERS), ('t_ptr', C.c_void_p * KGW_MAX_LAYERS),
        ('t_edge', C.c_void_p * KGW_MAX_LAYERS), ('t_zrow', C.c_void_p * KGW_MAX_LAYERS),
        ('t_rel', C.c_void_p * KGW_MAX_LAYERS),
        ('scan_tmp', C.c_void_p), ('t_tmp', C.c_void_p), ('meta', C.c_void_p), ('meta_host', C.c_void_p),
        ('seg_cap', C.c_int64), ('edge_cap', C.c_int64), ('chunk_cap', C.c_int64),
        ('multi_cap', C.c_int64), ('trow_cap', C.c_int64), ('scan_cap', C.c_int64),
        ('grid_blocks', C.c_int32), ('pad_', C.c_int32),
    ]


class KgwLayerArgs(C.Structure):
    _fields_ = [
        ('layer', C.c_int32), ('n_chunks', C.c_int32), ('n_multi_hops', C.c_int32), ('n_src_rows', C.c_int32),
        ('neg_slope', C.c_float), ('inv_temp', C.c_float), ('flags', C.c_int32), ('pad_', C.c_int32),
        ('graph_host', C.c_void_p), ('meta_host', C.c_void_p), ('meta_dev', C.c_void_p),
        ('chunks', C.c_void_p), ('multi', C.c_void_p), ('multi_cap', C.c_int64),
        ('col_local', C.c_void_p), ('H', C.c_void_p), ('a_dst', C.c_void_p), ('V', C.c_void_p), ('U', C.c_void_p),
        ('Z', C.c_void_p), ('stat', C.c_void_p), ('e_edge', C.c_void_p), ('part', C.c_void_p),
        ('dZ', C.c_void_p), ('adp', C.c_void_p), ('da_dst', C.c_void_p), ('part_da', C.c_void_p),
        ('t_ptr', C.c_void_p), ('t_edge', C.c_void_p), ('t_zrow', C.c_void_p),
        ('dH', C.c_void_p), ('ev_before', C.c_void_p), ('ev_after', C.c_void_p), ('da_src', C.c_void_p),
        ('logit_bias', C.c_void_p), ('partial_rels', C.c_uint64),
        ('t_rel', C.c_void_p), ('oct_flags', C.c_void_p), ('rel_sums', C.c_void_p),
        ('part_du', C.c_void_p), ('seg_chptr', C.c_void_p), ('duv_ws', C.c_void_p), ('dU', C.c_void_p), ('dV', C.c_void_p),
    ]


class KgwTnJob(C.Structure):
    _fields_ = [('A', C.c_void_p), ('lda', C.c_int64), ('B', C.c_void_p), ('ldb', C.c_int64), ('rows', C.c_int64),
                ('C', C.c_void_p), ('ldc', C.c_int64), ('colsum_a', C.c_void_p), ('colsum_ld', C.c_int64),
                ('workspace', C.c_void_p), ('workspace_floats', C.c_int64), ('rows_dev', C.c_void_p),
                ('M', C.c_int32), ('N', C.c_int32), ('c_transposed', C.c_int32), ('colsum_repeat', C.c_int32)]


class KgwGradSrc(C.Structure):
    _fields_ = [('ws', C.c_void_p), ('kind', C.c_int32), ('nblk', C.c_int32), ('M', C.c_int32), ('N', C.c_int32), ('MT', C.c_int32),
                ('NT', C.c_int32), ('gy', C.c_int32), ('gz', C.c_int32), ('K1', C.c_int32), ('c_transposed', C.c_int32),
                ('packed', C.c_void_p), ('flip', C.c_int32), ('pad_', C.c_int32)]


KGW_SEGCOPY_MAX = 40


class KgwSegCopy(C.Structure):
    _fields_ = [('n', C.c_int32), ('to_slot', C.c_int32), ('ptr', C.c_void_p * KGW_SEGCOPY_MAX), ('units', C.c_int64 * KGW_SEGCOPY_MAX),
                ('slot_off', C.c_int64 * KGW_SEGCOPY_MAX), ('slots', C.c_void_p), ('slot_stride', C.c_int64), ('slot_index', C.c_void_p)]


ADAM_FUSED_MAX, ADAM_FUSED_SRC, ADAM_FUSED_COUNTERS = 40, 12, 65 * 32     # include/kgwas_hip.h: KGW_ADAM_FUSED_*


class KgwReadoutFold(C.Structure):
    _fields_ = [('scratch', C.c_void_p), ('terms', C.c_void_p), ('dw_lin', C.c_void_p), ('db_lin', C.c_void_p), ('loss', C.c_void_p),
                ('nb', C.c_int32), ('n', C.c_int32)]


class KgwTnReducePlan(C.Structure):
    _fields_ = [('valid', C.c_int32), ('blocks', C.c_int32), ('reserved', C.c_int32 * 2), ('opaque', C.c_int64 * 96)]


class KgwSplitKJob(C.Structure):
    _fields_ = [('X', C.c_void_p), ('ldx', C.c_int64), ('W', C.c_void_p), ('ldw', C.c_int64), ('bias', C.c_void_p),
                ('Y', C.c_void_p), ('ldy', C.c_int64), ('rows', C.c_int64), ('seg_stat', C.c_void_p), ('gamma', C.c_void_p),
                ('dgamma', C.c_void_p), ('K', C.c_int32), ('N', C.c_int32), ('relu', C.c_int32), ('w_is_kn', C.c_int32)]


class KgwRelvecJob(C.Structure):
    _fields_ = [('n_rels_total', C.c_int32), ('n_live', C.c_int32), ('n_blk', C.c_int32), ('duv_pieces', C.c_int32),
                ('live_of_rel', C.c_void_p), ('rel_ids', C.c_void_p), ('bip_pos', C.c_void_p),
                ('w_src_t', C.c_void_p), ('w_dst_t', C.c_void_p), ('att_src', C.c_void_p), ('att_dst', C.c_void_p),
                ('U_full', C.c_void_p), ('V', C.c_void_p), ('bias', C.c_void_p), ('blk_of_live', C.c_void_p), ('bias_sum', C.c_void_p),
                ('zero_buf', C.c_void_p), ('zero_floats', C.c_int64),
                ('dU_full', C.c_void_p), ('dV', C.c_void_p), ('dw_src_acc', C.c_void_p),
                ('dw_src_t', C.c_void_p), ('dw_dst_t', C.c_void_p), ('datt_src', C.c_void_p), ('datt_dst', C.c_void_p)]


class KgwFoldArgs(C.Structure):
    _fields_ = [('n', C.c_int32), ('n_rels', C.c_int32), ('n_mlp', C.c_int32), ('duv_pieces', C.c_int32),
                ('rel_ids_host', C.c_void_p), ('src_mlp_host', C.c_void_p), ('dst_mlp_host', C.c_void_p),
                ('w_src_t', C.c_void_p), ('fc_weight', C.c_void_p * 4), ('fc_bias', C.c_void_p * 4),
                ('U', C.c_void_p), ('V', C.c_void_p),
                ('Up', C.c_void_p), ('Vp', C.c_void_p), ('kappa', C.c_void_p), ('Wp', C.c_void_p), ('gamma', C.c_void_p),
                ('dUp', C.c_void_p), ('dVp', C.c_void_p), ('dkappa', C.c_void_p), ('dWp', C.c_void_p), ('dgamma', C.c_void_p),
                ('dU', C.c_void_p), ('dV', C.c_void_p), ('dws', C.c_void_p),
                ('d_fc_weight', C.c_void_p * 4), ('d_fc_bias', C.c_void_p * 4)]


EXPORTS = ['kgw_version', 'kgw_status_string', 'kgw_struct_sizes', 'kgw_sample_batch', 'kgw_sample_batch_parts', 'kgw_sampler_scan_ints', 'kgw_segments_copy',
           'kgw_softmax_pack', 'kgw_softmax_merge', 'kgw_scatter_rows', 'kgw_linear_splitk', 'kgw_linear_splitk_workspace_floats', 'kgw_linear_splitk_ind', 'kgw_ind_colsum', 'kgw_linear_splitk_multi', 'kgw_ind_colsum_multi', 'kgw_fold_fwd', 'kgw_fold_bwd', 'kgw_relation_sums',
           'kgw_gat_aggregate_fwd', 'kgw_gat_aggregate_bwd_dst', 'kgw_gat_aggregate_bwd_src',
           'kgw_gather_rows', 'kgw_gather_rows_multi', 'kgw_scatter_relu_rows', 'kgw_scatter_relu_rows_workspace_floats', 'kgw_edge_alpha', 'kgw_debug_reduce', 'kgw_debug_reduce8', 'kgw_tn_gemm', 'kgw_tn_gemm_ex', 'kgw_tn_gemm_multi', 'kgw_tn_gemm_workspace_floats', 'kgw_tn_gemm_partial', 'kgw_tn_gemm_multi_partial', 'kgw_tn_split', 'kgw_tn_direct_rows', 'kgw_param_tail', 'kgw_tn_reduce_launch', 'kgw_tn_gemm_partial_ride', 'kgw_transform_bwd_ex', 'kgw_mlp2_bwd_first_partial', 'kgw_mlp2_bwd_first_packed', 'kgw_adam_fused', 'kgw_gemm3_partial', 'kgw_gemm3_flip', 'kgw_transform_bwd', 'kgw_grad_finish',
           'kgw_linear', 'kgw_mlp2_fwd', 'kgw_mlp2w_fwd', 'kgw_mlp2_bwd_first', 'kgw_mlp2_bwd_first_workspace_floats', 'kgw_gemm3', 'kgw_gemm3_riders', 'kgw_gemm3_rider_blocks', 'kgw_gemm3_pack', 'kgw_gemm3_packed_bytes', 'kgw_gemm3_workspace_floats', 'kgw_adam', 'kgw_adam_notick', 'kgw_relvec_fwd', 'kgw_relvec_bwd', 'kgw_relvec_bwd_acc', 'kgw_relvec_fwd_multi', 'kgw_relvec_bwd_multi', 'kgw_wmse_fwd', 'kgw_wmse_bwd', 'kgw_readout_wmse_fwd', 'kgw_readout_wmse_bwd', 'kgw_readout_wmse_train', 'kgw_readout_wmse_train_parts', 'kgw_readout_train_fold', 'kgw_accumulate_stats', 'kgw_accumulate_stats_tick']

_lib = None


class KgwasHipError(RuntimeError):
    pass


def lib():
    """Load libkgwas_hip.so once.  Raises (never falls back) when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KgwasHipError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  kgwas_amd has no CPU fallback.')
    L = C.CDLL(LIB_PATH)
    L.kgw_version.restype = C.c_int
    L.kgw_status_string.restype = C.c_char_p
    L.kgw_status_string.argtypes = [C.c_int]
    L.kgw_struct_sizes.argtypes = [C.POINTER(C.c_int64), C.c_int]
    sizes = (C.c_int64 * 7)()
    L.kgw_struct_sizes(sizes, 7)
    mine = [C.sizeof(KgwGraph), C.sizeof(KgwBatchMeta), C.sizeof(KgwChunk), C.sizeof(KgwBatchBuf),
            C.sizeof(KgwLayerArgs), C.sizeof(KgwTnJob), C.sizeof(KgwGradSrc)]
    if list(sizes) != mine:
        raise KgwasHipError(f'ABI struct size mismatch: library {list(sizes)} vs binding {mine}')
    L.kgw_sample_batch.argtypes = [C.POINTER(KgwGraph), C.POINTER(KgwBatchBuf), C.c_void_p, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_void_p]
    L.kgw_sample_batch_parts.argtypes = [C.POINTER(KgwGraph), C.POINTER(KgwBatchBuf), C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.kgw_sampler_scan_ints.argtypes = [C.c_int64, C.c_int64, C.c_int64]
    L.kgw_sampler_scan_ints.restype = C.c_int64
    L.kgw_segments_copy.argtypes = [C.POINTER(KgwSegCopy), C.c_int32, C.c_void_p]
    L.kgw_softmax_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.kgw_softmax_merge.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kgw_scatter_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    for name in ('kgw_gat_aggregate_fwd', 'kgw_gat_aggregate_bwd_dst', 'kgw_gat_aggregate_bwd_src'):
        getattr(L, name).argtypes = [C.POINTER(KgwLayerArgs), C.c_void_p]
    L.kgw_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.kgw_tn_gemm_multi.argtypes = [C.c_int32, C.POINTER(KgwTnJob), C.c_void_p]
    L.kgw_scatter_relu_rows_workspace_floats.restype = C.c_int64
    L.kgw_scatter_relu_rows_workspace_floats.argtypes = [C.c_int64]
    L.kgw_scatter_relu_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kgw_gather_rows_multi.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32,
                                        C.POINTER(C.c_void_p), C.c_void_p]
    L.kgw_edge_alpha.argtypes = [C.POINTER(KgwLayerArgs), C.c_void_p, C.c_void_p]
    L.kgw_debug_reduce.argtypes = [C.c_void_p] * 5
    L.kgw_debug_reduce8.argtypes = [C.c_void_p] * 3
    L.kgw_tn_gemm_workspace_floats.restype = C.c_int64
    L.kgw_tn_gemm_workspace_floats.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    L.kgw_linear.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                             C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                             C.c_void_p]
    L.kgw_mlp2_fwd.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                               C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p]
    L.kgw_mlp2w_fwd.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.kgw_gemm3_packed_bytes.restype = C.c_int64
    L.kgw_gemm3_packed_bytes.argtypes = [C.c_int64]
    L.kgw_gemm3_workspace_floats.restype = C.c_int64
    L.kgw_gemm3_workspace_floats.argtypes = [C.c_int64, C.c_int64]
    L.kgw_gemm3_pack.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.kgw_gemm3.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                            C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.kgw_gemm3_riders.argtypes = L.kgw_gemm3.argtypes[:-1] + [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.kgw_gemm3_rider_blocks.argtypes = [C.c_int64, C.c_int64]
    L.kgw_mlp2_bwd_first_workspace_floats.restype = C.c_int64
    L.kgw_mlp2_bwd_first_workspace_floats.argtypes = [C.c_int64]
    L.kgw_mlp2_bwd_first.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32,
                                     C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_int64, C.c_void_p]
    L.kgw_linear_splitk_workspace_floats.restype = C.c_int64
    L.kgw_linear_splitk_workspace_floats.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    L.kgw_linear_splitk.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.kgw_linear_splitk_ind.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                        C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.kgw_ind_colsum.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.kgw_linear_splitk_multi.argtypes = [C.c_int32, C.POINTER(KgwSplitKJob), C.c_void_p]
    L.kgw_ind_colsum_multi.argtypes = [C.c_int32, C.POINTER(KgwSplitKJob), C.c_void_p]
    L.kgw_relation_sums.argtypes = [C.POINTER(KgwLayerArgs), C.c_void_p, C.c_void_p, C.c_void_p]
    L.kgw_fold_fwd.argtypes = [C.POINTER(KgwFoldArgs), C.c_void_p]
    L.kgw_fold_bwd.argtypes = [C.POINTER(KgwFoldArgs), C.c_void_p]
    L.kgw_adam.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.kgw_adam_notick.argtypes = L.kgw_adam.argtypes
    L.kgw_relvec_fwd.argtypes = [C.c_int32] + [C.c_void_p] * 8 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                 C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.kgw_relvec_fwd_multi.argtypes = [C.c_int32, C.POINTER(KgwRelvecJob), C.c_void_p]
    L.kgw_relvec_bwd_multi.argtypes = [C.c_int32, C.POINTER(KgwRelvecJob), C.c_void_p]
    L.kgw_relvec_bwd.argtypes = [C.c_int32] + [C.c_void_p] * 12 + [C.c_int32, C.c_void_p]
    L.kgw_relvec_bwd_acc.argtypes = [C.c_int32] + [C.c_void_p] * 13 + [C.c_int32, C.c_void_p]
    L.kgw_tn_gemm.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                              C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.kgw_tn_gemm_ex.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                 C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p,
                                 C.c_int64, C.c_void_p, C.c_void_p]
    L.kgw_tn_gemm_partial.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                      C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.POINTER(KgwGradSrc), C.c_void_p]
    L.kgw_tn_gemm_multi_partial.argtypes = [C.c_int32, C.POINTER(KgwTnJob), C.POINTER(KgwGradSrc), C.c_void_p]
    L.kgw_param_tail.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.kgw_mlp2_bwd_first_partial.argtypes = L.kgw_mlp2_bwd_first.argtypes[:-1] + [C.POINTER(KgwGradSrc), C.c_void_p]
    L.kgw_mlp2_bwd_first_packed.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32,
                                            C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.kgw_adam_fused.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(KgwGradSrc), C.c_void_p,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p]
    L.kgw_gemm3_partial.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                    C.POINTER(KgwGradSrc), C.c_void_p]
    L.kgw_gemm3_flip.restype = C.c_int
    L.kgw_grad_finish.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(KgwGradSrc), C.c_void_p]
    L.kgw_transform_bwd.argtypes = [C.c_int32, C.POINTER(KgwTnJob), C.c_int32, C.POINTER(KgwSplitKJob), C.c_int32, C.POINTER(KgwSplitKJob),
                                    C.c_void_p]
    L.kgw_transform_bwd_ex.argtypes = L.kgw_transform_bwd.argtypes[:-1] + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kgw_tn_reduce_launch.argtypes = [C.c_void_p, C.c_void_p]
    L.kgw_tn_direct_rows.restype = C.c_int64
    L.kgw_tn_direct_rows.argtypes = [C.c_int64]
    L.kgw_tn_gemm_partial_ride.argtypes = L.kgw_tn_gemm_partial.argtypes[:-1] + [C.c_void_p, C.c_void_p]
    L.kgw_wmse_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_void_p, C.c_void_p]
    L.kgw_wmse_bwd.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kgw_readout_wmse_fwd.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_int32] + [C.c_void_p] * 4
    L.kgw_readout_wmse_bwd.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_int64, C.c_int32] + [C.c_void_p] * 6
    L.kgw_readout_wmse_train.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_int64, C.c_int32] + [C.c_void_p] * 8
    L.kgw_readout_wmse_train_parts.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_int64, C.c_int32] + [C.c_void_p] * 9
    L.kgw_readout_train_fold.argtypes = [C.c_void_p, C.c_void_p]
    L.kgw_accumulate_stats.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.kgw_accumulate_stats_tick.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = L
    return L


KGW_E_UNSUPPORTED = -3


def check(status: int, what: str):
    if status != 0:
        msg = lib().kgw_status_string(int(status))
        raise KgwasHipError(f'{what} failed: status {status} ({msg.decode() if msg else "?"})')


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
