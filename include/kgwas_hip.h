/*
 * kgwas_hip.h -- C ABI of libkgwas_hip.so, the MI355X (gfx950) hot path of KGWAS.
 *
 * The reference (snap-stanford/KGWAS) is pure Python and has NO FFI / plugin boundary of its own
 * (SURVEY.md 8b): its native work happens inside third-party wheels (torch_geometric,
 * torch_sparse / pyg_lib, torch_scatter).  This header is therefore the seam *under* the
 * reference's Python API; each entry point names the reference call site whose native work it
 * replaces.  All pointers are device pointers unless named *_host; all buffers are owned by the
 * caller (torch-allocated); no function allocates, frees or synchronises; every function only
 * enqueues work on `stream` (a hipStream_t) and returns a status:
 *      0 = ok, <0 = argument error (KGW_E_*), >0 = hipError_t from a launch.
 * No C++ exceptions cross this boundary.  Functions are re-entrant (no global mutable state).  What the library keeps per
 * process is immutable after first use: a handful of launch-shape constants read ONCE from KGW_* environment variables (timing
 * experiments: KGW_AGG_GRID_CAP, KGW_G3_NW, KGW_SPLITK_BLOCKS, ... -- unset in normal use) and the one-time
 * hipFuncSetAttribute calls of the kernels that take more than 64 KB of LDS.
 *
 * Vocabulary (the reference's domain): node types, relations (= edge types), seeds, hops,
 * segments (= one destination row of one relation), chunks (<= KGW_CHUNK edges of one segment).
 */
#ifndef KGWAS_HIP_H
#define KGWAS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KGW_VERSION      125          /* 0.2.5 */
#define KGW_MAX_TYPES    8
#define KGW_MAX_RELS     64
#define KGW_MAX_LAYERS   4
#define KGW_CHUNK        128          /* edges per chunk (one wavefront processes one chunk; the longest chunks set
                                         the aggregate kernels' tail: 256 -> 128 took k_agg_fwd 64 -> 56 us)           */
#define KGW_TILE         1024         /* scan tile; per-type node regions are padded to this  */
#define KGW_C            128          /* hidden width (gnn_hidden_dim, kgwas/kgwas.py:52)     */

#define KGW_F_RAW_WEIGHTS 1           /* KgwLayerArgs.flags: no softmax (the reference's attention export,
                                         kgwas/utils.py:446-461 with return_raw_attention_weights)      */
#define KGW_F_DUV_PIECES  4           /* bwd_src with dU / dV riders: leave d u_r / d v_r as their eight level-1 pieces in duv_ws
                                         ([2][n_rels][8][128]: k_duv_fold is not launched, dU / dV are not written) -- the consumer
                                         adds them: KgwRelvecJob.duv_pieces / KgwFoldArgs.duv_pieces                              */
#define KGW_F_RELU_INPUT  2           /* bwd_src: H is the output of a ReLU (model.py:75) whose backward is folded
                                         into this pass: dH *= (H > 0)                                   */

#define KGW_OK           0
#define KGW_E_NULL      -1
#define KGW_E_RANGE     -2
#define KGW_E_UNSUPPORTED -3

typedef void* kgw_stream_t;           /* hipStream_t */

/* Resident knowledge graph: one dst-major CSR per relation, (dst, src)-sorted like the CSC that
 * NeighborLoader builds once per edge type (kgwas/kgwas.py:99-113; PyG to_csc).               */
typedef struct KgwGraph {
    int32_t n_types, n_rels, n_layers, n_hops;  /* n_hops = n_layers (minibatch) or 1 (full graph) */
    int32_t n_nodes[KGW_MAX_TYPES];       /* N_T                                             */
    int32_t node_base[KGW_MAX_TYPES + 1]; /* KGW_TILE-aligned start of type T in g2l / n_id   */
    int32_t R_dst[KGW_MAX_TYPES];         /* #relations whose destination type is T           */
    int32_t R_src[KGW_MAX_TYPES];         /* #relations whose source type is T                */
    int32_t rel_src[KGW_MAX_RELS];
    int32_t rel_dst[KGW_MAX_RELS];
    int32_t rel_slot_dst[KGW_MAX_RELS];   /* column block of relation r in Z[dst type]        */
    int32_t rel_slot_src[KGW_MAX_RELS];
    int64_t rowptr_off[KGW_MAX_RELS];     /* element offset of relation r inside g_rowptr     */
    int64_t col_off[KGW_MAX_RELS];        /* element offset of relation r inside g_col        */
    uint8_t rel_live[KGW_MAX_LAYERS][KGW_MAX_RELS]; /* [l-1][r]: layer l computes relation r  */
    /* Static layout (HIP-graph capture): when static_layout != 0 the row blocks of layer l are laid out
     * with these fixed capacities instead of the batch's own counts, so every buffer address and launch
     * geometry is batch independent; a batch that needs more sets KgwBatchMeta.error bit 5.          */
    int32_t static_layout;
    int32_t cap_rows[KGW_MAX_LAYERS][KGW_MAX_TYPES];  /* destination rows of type T in layer l        */
    int32_t cap_src[KGW_MAX_LAYERS][KGW_MAX_TYPES];   /* source rows of type T in layer l             */
    uint32_t short_types;                 /* bit T: nodes of type T have few out-edges (mean <= 4): kgw_sample_batch marks the
                                             groups of 8 consecutive source rows of such a type that the backward can process
                                             8 lanes per row (KgwLayerArgs.oct_flags); a scheduling hint, never a correctness one */
    const int32_t* g_rowptr;              /* per relation N_dst+1 entries, relative to col_off */
    const int32_t* g_col;                 /* global source ids                                */
} KgwGraph;

/* Counts and layouts of one sampled batch; written by the device, copied to meta_host.        */
typedef struct KgwBatchMeta {
    int32_t hop_cnt[KGW_MAX_TYPES][KGW_MAX_LAYERS + 1];   /* new nodes of type T at hop k      */
    int32_t node_off[KGW_MAX_TYPES][KGW_MAX_LAYERS + 2];  /* prefix over hops (local id ranges) */
    int32_t seg_off[KGW_MAX_LAYERS][KGW_MAX_RELS + 1];    /* first segment of (dst hop h, rel r) */
    int32_t seg_end[KGW_MAX_LAYERS];     /* cumulative #segments through dst hop h             */
    int32_t edge_end[KGW_MAX_LAYERS];    /* cumulative #edges    through dst hop h             */
    int32_t chunk_end[KGW_MAX_LAYERS];   /* cumulative #chunks   through dst hop h             */
    int32_t multi_cnt[KGW_MAX_LAYERS];   /* #multi-chunk segments in dst hop h                 */
    /* per layer l (index l-1): */
    int32_t n_rows[KGW_MAX_LAYERS][KGW_MAX_TYPES];   /* destination rows of type T            */
    int32_t z_base[KGW_MAX_LAYERS][KGW_MAX_TYPES + 1]; /* first Z row (of 128 floats) of type T */
    int32_t n_src[KGW_MAX_LAYERS][KGW_MAX_TYPES];    /* source rows of type T in the layer input */
    int32_t src_base[KGW_MAX_LAYERS][KGW_MAX_TYPES + 1]; /* first H row of type T             */
    int32_t t_base[KGW_MAX_LAYERS][KGW_MAX_TYPES + 1];   /* first transposed row of type T    */
    int32_t lay_rows[KGW_MAX_LAYERS][KGW_MAX_TYPES]; /* row-block sizes the bases above were built from */
    int32_t lay_src[KGW_MAX_LAYERS][KGW_MAX_TYPES];  /*  (= n_rows / n_src unless static_layout)        */
    int32_t n_chunks[KGW_MAX_LAYERS];    /* chunks used by layer l                             */
    int32_t n_edges[KGW_MAX_LAYERS];     /* edges aggregated by layer l                        */
    int32_t t_entries[KGW_MAX_LAYERS];   /* entries in the layer's src-major structure         */
    int32_t cur[8];                      /* device-side scratch cursors                        */
    int32_t error;                       /* !=0: a capacity was exceeded                       */
    int32_t pad_[3];
} KgwBatchMeta;

/* Chunk record: <= KGW_CHUNK consecutive edges of one segment.  8 x int32 = 32 bytes.          */
typedef struct KgwChunk {
    int32_t e0, e1;        /* local (batch) edge range                                        */
    int32_t row;           /* destination row (local id inside its node type)                 */
    int32_t rel;           /* relation id                                                     */
    int32_t first;         /* index of the first chunk of this segment                        */
    int32_t nch;           /* #chunks of this segment (1 => this chunk writes final results)   */
    int32_t gpos_lo, gpos_hi; /* position of edge e0 inside g_col (64-bit)                     */
} KgwChunk;

/* Buffers of one sampled batch (all device memory, worst-case sized by the caller).            */
typedef struct KgwBatchBuf {
    int32_t* g2l;          /* [node_base[n_types]] global -> local id, -1 = not sampled        */
    int32_t* n_id;         /* [node_base[n_types]] local -> global id, per type region         */
    int32_t* seg_deg;      /* [seg_cap]                                                        */
    int32_t* seg_nch;      /* [seg_cap]                                                        */
    int32_t* seg_ptr;      /* [seg_cap + 1]  first local edge of each segment                  */
    int32_t* seg_chptr;    /* [seg_cap + 1]  first chunk of each segment                       */
    int32_t* col_local;    /* [edge_cap]     source local id of each local edge                */
    KgwChunk* chunks;      /* [chunk_cap]                                                      */
    int32_t* multi;        /* [n_hops][multi_cap][4] = {first, nch, row, rel}                  */
    int32_t* t_cnt[KGW_MAX_LAYERS];   /* [trow_cap + 1] histogram / cursor scratch             */
    int32_t* t_ptr[KGW_MAX_LAYERS];   /* [trow_cap + 1] src-major row pointers                 */
    int32_t* t_edge[KGW_MAX_LAYERS];  /* [edge_cap] local edge id of each entry                */
    int32_t* t_zrow[KGW_MAX_LAYERS];  /* [edge_cap] Z row (dst row * R_dst + slot) of the entry */
    uint8_t* t_rel[KGW_MAX_LAYERS];   /* [edge_cap] relation id of the entry (optional: NULL = not written)  */
    int32_t* scan_tmp;     /* [scan_cap] >= kgw_sampler_scan_ints(seg_cap, node slots, trow_cap): scan scratch of the hops, then
                              the [layer][bucket][block] counts of the src-major sort                                             */
    int32_t* t_tmp;        /* [8 * (edge_cap + 1)], 16-B aligned, four arrays of edge_cap + 1 per layer of a pair (the second layer's
                              behind the first's): row key of every edge | (first layer's slot only) chunk of every edge, written
                              hop by hop by the relabelling pass and shared by all layers | keys | edge ids stably sorted by bucket
                              (the src-major radix sort)                                                                           */
    KgwBatchMeta* meta;    /* device                                                           */
    KgwBatchMeta* meta_host; /* pinned host mirror (async D2H at the end of sampling)          */
    int64_t seg_cap, edge_cap, chunk_cap, multi_cap, trow_cap, scan_cap;
    int32_t grid_blocks;   /* blocks of the sampler's grid-stride launches; 0 = default (2048: the call has the GPU to
                              itself).  A sampler replayed BESIDE a training step (second HIP graph on a side stream):
                              1024 -- what it costs the step is how long each of its kernels stays resident */
    int32_t pad_;
} KgwBatchBuf;

/* One attention-aggregate layer over a sampled batch (kgwas/conv.py:177-190 for every relation
 * of the layer at once + the relation sum of PyG HeteroConv, kgwas/model.py:74).               */
typedef struct KgwLayerArgs {
    int32_t layer;                 /* 1-based                                                  */
    int32_t n_chunks, n_multi_hops;/* launch-size HINT (>= the batch's chunk count; the kernels read the
                                      actual counts from meta_dev) ; multi lists of dst hops [0,n_multi_hops) */
    int32_t n_src_rows;            /* rows of H / dH in the layout (src_base[n_types])         */
    float   neg_slope, inv_temp;   /* LeakyReLU slope (0.2), 1/temperature (1)                 */
    int32_t flags;                 /* KGW_F_RAW_WEIGHTS: forward weights messages by the raw logits   */
    int32_t pad_;
    const KgwGraph* graph_host;    /* host copy, passed by value to the kernels               */
    const KgwBatchMeta* meta_host; /* host struct holding the LAYOUT (z_base, src_base, t_base, lay_*) */
    const KgwBatchMeta* meta_dev;  /* device struct written by kgw_sample_batch (actual counts)        */
    const KgwChunk* chunks;
    const int32_t* multi;  int64_t multi_cap;
    const int32_t* col_local;
    const float* H;                /* [n_src_rows][128] layer input, type-major (src_base)     */
    const float* a_dst;            /* [z rows]  <h_dst[i], W_dst^T att_dst> per (row, relation); read only if V == NULL */
    const float* V;                /* [n_rels][128]  v_r = W_dst^T att_dst: a_dst is computed in-kernel from the
                                      destination node's own row of H (every destination type has a block in H)  */
    const float* U;                /* [n_rels][128]  u_r = W_src^T att_src                     */
    float* Z;                      /* [z rows][128]  sum_j alpha_ij h_src[j]  (pre-zeroed)      */
    float* stat;                   /* [z rows][2]    (row max, denominator)                    */
    float* e_edge;                 /* [n_edges]      leaky_relu logits per local edge          */
    float* part;                   /* [n_chunks][130] partial (m, s, acc) of multi-chunk segs  */
    /* backward */
    const float* dZ;               /* [z rows][128]                                            */
    float* adp;                    /* [n_edges][2]   (alpha, d pre-activation) per local edge  */
    float* da_dst;                 /* [z rows]                                                 */
    float* part_da;                /* [n_chunks][4], 16-byte aligned: per-chunk sums of a multi-chunk row's d a_dst */
    const int32_t* t_ptr; const int32_t* t_edge; const int32_t* t_zrow;
    float* dH;                     /* [n_src_rows][128]                                        */
    /* optional profiling hooks (NULL = off): hipEvent_t handles owned by the caller, recorded on `stream`
     * immediately before and after the MAIN kernel of the call (k_agg_fwd / k_agg_bwd_dst / k_agg_bwd_src), i.e.
     * excluding the small combine launches for hub rows -- bench.py's per-kernel roofline timing */
    void* ev_before; void* ev_after;
    float* da_src;                 /* [n_src_rows][2*ld], ld = (n_rels+3)&~3: columns [0,ld) d a_src, [ld,2ld) d a_dst
                                      of the node, one column per relation id (n_rels <= KGW_MAX_RELS)  */
    const float* logit_bias;       /* optional [n_rels]: constant added to the pre-activation logit of every edge of relation r
                                      (FC_output folded into the layer-1 relation parameters: the feature MLP's last Linear
                                      H = h2 T + c enters conv.py:150-152 only through <H, u_r> and <H, v_r>, whose constant
                                      parts <c_src, u_r> + <c_dst, v_r> land here while T is multiplied into U and V)     */
    uint64_t partial_rels;         /* forward: bit r set => the segments of relation r are written as PARTIAL online-softmax
                                      states -- Z = sum_j exp(e_ij - m) h_j (not divided), stat = (m, sum_j exp(e_ij - m)) --
                                      for the caller to merge across GPUs (SNP-sharded mode: a rank holds only its own SNP
                                      sources of a SNP->Gene relation; kgw_softmax_merge) before anything reads them;
                                      backward (dst pass): d a_dst of these rows is the plain sum over this rank's edges (the
                                      ranks' values are added), without the row-consistent correction of whole rows          */
    const uint8_t* t_rel;          /* optional (NULL = off), backward src pass: relation id of every src-major entry
                                      (KgwBatchBuf.t_rel) and                                                              */
    const int32_t* oct_flags;      /* one flag per group of 8 consecutive source rows (row index / 8; KgwBatchBuf.t_cnt[layer-1]
                                      after kgw_sample_batch): != 0 => the eight rows are real rows of one node type of
                                      KgwGraph.short_types, none of them a destination row of the layer, each with at most 8
                                      entries.  Such a group is processed by ONE wavefront, 8 lanes per row, without the per-row
                                      slot bookkeeping of the general path (the SNP rows: ~2 entries, two thirds of all rows)   */
    float* rel_sums;               /* optional, kgw_gat_aggregate_bwd_src: [n_rels] <- per relation, the sum of d a_dst over its
                                      destination rows = the gradient of logit_bias (what kgw_relation_sums computes), by
                                      n_rels extra blocks of the same launch.  Needs V (in-kernel a_dst) so that da_dst is final */
    /* d u_r / d v_r without the [d a_src | d a_dst] rows (da_src may then be NULL): kgw_gat_aggregate_bwd_dst leaves, per chunk,
     * sum_e dpre_e h_src(e) in part_du [n_chunks][128]; kgw_gat_aggregate_bwd_src adds them up per relation (and
     * d v_r = sum_i d a_dst[i, r] h_dst[i] over the destination rows) with extra blocks of its launch plus one small fold launch,
     * into dU / dV [n_rels][128] (zero rows for relations the layer does not compute).  seg_chptr = KgwBatchBuf.seg_chptr;
     * duv_ws: 2 * n_rels * 8 * 128 floats of workspace.  All five or none.                                              */
    float* part_du; const int32_t* seg_chptr; float* duv_ws; float* dU; float* dV;
} KgwLayerArgs;

/* ---- entry points ------------------------------------------------------------------------ */

int kgw_version(void);
const char* kgw_status_string(int status);
/* sizeof() of the ABI structs in the order Graph, BatchMeta, Chunk, BatchBuf, LayerArgs -- lets a
 * binding verify its mirror structs.                                                          */
int kgw_struct_sizes(int64_t* out, int n);

/* Replaces: NeighborLoader.__next__ (kgwas/kgwas.py:99-113,129; torch_sparse/pyg_lib
 * hetero_neighbor_sample + relabel).  Expands `n_seeds` seed nodes of type `seed_type` over
 * n_hops hops taking ALL in-neighbours, writes local ids, per-segment CSR, chunk lists and the
 * src-major (transposed) structure of every layer, then copies KgwBatchMeta to meta_host.
 * full_graph != 0: every node of every type is a seed (whole-graph inference / attention export,
 * kgwas/utils.py:446-461).                                                                    */
int kgw_sample_batch(const KgwGraph* graph, const KgwBatchBuf* buf, const int64_t* seeds,
                     int32_t n_seeds, int32_t seed_type, int32_t full_graph, kgw_stream_t stream);
/* ints KgwBatchBuf.scan_tmp must hold (KgwBatchBuf.scan_cap) for buffers of these capacities.  (The src-major sort of
 * kgw_sample_batch takes blocks of up to 147 M src-major rows = sum over node types of rows x relations leaving the type;
 * beyond that the call returns KGW_E_UNSUPPORTED.)                                                                          */
int64_t kgw_sampler_scan_ints(int64_t seg_cap, int64_t node_slots, int64_t trow_cap);
/* The same call in parts [part_begin, part_end] (kgw_sample_batch = parts 0 .. 2*n_hops): hop h is part 2h (its segments
 * and chunks, and the flag KGW_PENDING = -2 in g2l on every not-yet-sampled source node) and part 2h+1 (flags -> local
 * ids, relabelled edges); part 2*n_hops = layer tables, src-major structures, meta export.  SNP-sharded multi-GPU mode
 * (BASELINE.json north_star; no counterpart in the single-device reference, kgwas/kgwas.py:38-39): between parts 2h and
 * 2h+1 the ranks take the element-wise MIN of the g2l regions of the REPLICATED node types (one RCCL all-reduce of
 * ~44 k int32), so every rank expands the same Gene / GO frontier although it only sees its own SNP seeds.        */
int kgw_sample_batch_parts(const KgwGraph* graph, const KgwBatchBuf* buf, const int64_t* seeds,
                           int32_t n_seeds, int32_t seed_type, int32_t full_graph, int32_t part_begin,
                           int32_t part_end, kgw_stream_t stream);

/* Replaces: GATConv.edge_update + message + aggregate (kgwas/conv.py:200-228,182) and their
 * autograd for all relations of one layer.                                                    */
int kgw_gat_aggregate_fwd(const KgwLayerArgs* args, kgw_stream_t stream);
int kgw_gat_aggregate_bwd_dst(const KgwLayerArgs* args, kgw_stream_t stream);
int kgw_gat_aggregate_bwd_src(const KgwLayerArgs* args, kgw_stream_t stream);

/* SNP-sharded mode, the exchange step of a layer: the softmax of kgwas/conv.py:223 runs over ALL
 * in-edges of a destination gene, but a rank only aggregated the edges whose SNP source it owns.
 * kgw_softmax_pack: record x (132 floats: [0] = m, [1] = s, [4..131] = acc) <- the partial state of segment
 *   seg_zrow[x] (Z row / stat pair written by kgw_gat_aggregate_fwd under KgwLayerArgs.partial_rels).
 * kgw_softmax_merge: parts [n_ranks][n_seg][132] (every rank's pack, gathered in rank order) ->
 *   Z[seg_zrow[x]] = sum_p acc_p e^(m_p - m*) / (sum_p s_p e^(m_p - m*) + 1e-16), stat = (m*, that denominator), with
 *   m* = max over the ranks that saw an edge (s_p > 0); fixed rank order, so every rank computes the same bits.
 * kgw_scatter_rows: dst[ids[i]] = src[i] (rows of `width` floats): puts the all-reduced dZ rows of the exchanged segments
 *   back (the gather is kgw_gather_rows).                                                                          */
int kgw_softmax_pack(const float* Z, const float* stat, const int32_t* seg_zrow, int64_t n_seg, float* parts,
                     kgw_stream_t stream);
int kgw_softmax_merge(const float* parts, int32_t n_ranks, const int32_t* seg_zrow, int64_t n_seg, float* Z,
                      float* stat, kgw_stream_t stream);
int kgw_scatter_rows(const float* src, const int32_t* ids, int64_t n_rows, int32_t width, float* dst,
                     kgw_stream_t stream);

/* out[r] = sum over the destination rows i of relation r of x[Z row of segment (i, r)] for every relation the layer
 * computes, 0 for the others (x: one float per segment, e.g. KgwLayerArgs.da_dst).  The gradient of the per-relation
 * logit constant (KgwLayerArgs.logit_bias) is the sum of d pre-activation over all edges of the relation = this sum of
 * d a_dst.  One block per relation, fixed reduction tree.                                                          */
int kgw_relation_sums(const KgwLayerArgs* args, const float* x, float* out, kgw_stream_t stream);

/* Running totals over the batches of a captured training loop: stats[l] += edges aggregated by layer l+1 (l < n_layers),
 * stats[n_layers] += edges sampled, stats[n_layers+1] |= KgwBatchMeta.error.  stats: n_layers + 2 device int64.   */
int kgw_accumulate_stats(const KgwBatchMeta* meta_dev, int32_t n_layers, int32_t n_hops, int64_t* stats,
                         kgw_stream_t stream);
/* ... and, tick != NULL, advances that int32 by one in the same launch: the step counter of a kgw_adam_notick earlier in the
 * captured step (one single-thread launch less per step).                                                           */
int kgw_accumulate_stats_tick(const KgwBatchMeta* meta_dev, int32_t n_layers, int32_t n_hops, int64_t* stats, int32_t* tick,
                              kgw_stream_t stream);

/* A sampled batch kept for later epochs.  The reference's training loader has a fixed batch order (NeighborLoader without
 * shuffle, kgwas/kgwas.py:93-101), so epoch >= 2 asks the sampler for exactly the structures of epoch 1.  kgw_segments_copy moves
 * up to KGW_SEGCOPY_MAX device segments (the arrays of a KgwBatchBuf the training step reads) to slot *slot_index of a resident
 * cache (to_slot != 0) or back, in ONE launch; the slot index is read on the DEVICE, so the launch is captured once.  Pointers,
 * slot offsets and the stride are 16-byte aligned; units = 16-byte units of the segment.                                      */
#define KGW_SEGCOPY_MAX 40
typedef struct KgwSegCopy {
    int32_t n, to_slot;
    void* ptr[KGW_SEGCOPY_MAX];
    int64_t units[KGW_SEGCOPY_MAX];
    int64_t slot_off[KGW_SEGCOPY_MAX];
    uint8_t* slots;                  /* [n slots][slot_stride] */
    int64_t slot_stride;
    const int64_t* slot_index;       /* device */
} KgwSegCopy;
int kgw_segments_copy(const KgwSegCopy* plan, int32_t grid_blocks /* 0: default */, kgw_stream_t stream);

/* Replaces: the index_select feature slicing of the loader (x[n_id], kgwas/kgwas.py:135).      */
int kgw_gather_rows(const float* src, const int32_t* ids, int64_t n_rows, int32_t width,
                    float* dst, kgw_stream_t stream);
/* The same for up to 8 (src, ids, dst) jobs of one row width in a single launch (the three GO node
 * types' features go through one shared MLP as one matrix, kgwas/model.py:58-60).               */
int kgw_gather_rows_multi(int32_t n_jobs, const float* const* src, const int32_t* const* ids,
                          const int64_t* n_rows, int32_t width, float* const* dst, kgw_stream_t stream);

/* Backward of "rows n_id of relu(X W^T + b) computed on the RESIDENT feature matrix" (the wide gene layer,
 * kgwas/model.py:17-19 applied before the x[n_id] slicing of kgwas/kgwas.py:135): for every row of the resident
 * matrix dz[row] = g[g2l[row]] * (h[row] > 0) (zero where g2l[row] < 0: node not in the batch) and
 * colsum[c] = sum_row dz[row][c] (the bias gradient).  g [n_batch][128], h / dz [n_rows][128], g2l [n_rows];
 * workspace: kgw_scatter_relu_rows_workspace_floats(n_rows) floats.  Two launches, deterministic.        */
int64_t kgw_scatter_relu_rows_workspace_floats(int64_t n_rows);
int kgw_scatter_relu_rows(const float* g, const int32_t* g2l, const float* h, int64_t n_rows, float* dz,
                          float* colsum, float* workspace, kgw_stream_t stream);

/* alpha_e = exp(e - max)/den per local edge of one layer (attention export,
 * kgwas/conv.py:192-196; kgwas/utils.py:446-461).                                             */
int kgw_edge_alpha(const KgwLayerArgs* args, float* alpha_out, kgw_stream_t stream);

/* Where a gradient tensor's values are when its producer left the last reduction to the optimiser's launch
 * (kgw_adam_fused): DIRECT = in the gradient tensor itself; the other kinds = per-block partial sums in a workspace, added by
 * kgw_adam_fused in exactly the order the producer's own second launch (k_tn_reduce / k_mlp2_bwd_fold) uses -- bit-identical
 * gradients, one launch less per product.  Filled by kgw_tn_gemm_partial / kgw_tn_gemm_multi_partial /
 * kgw_mlp2_bwd_first_partial; the caller only carries the record to kgw_adam_fused.  (Weight / bias gradients of the Linears of
 * kgwas/model.py:13-21 on their way to the Adam update of kgwas/kgwas.py:116,151.)                                             */
enum { KGW_GRAD_DIRECT = 0, KGW_GRAD_TN = 1, KGW_GRAD_TN_COLSUM = 2, KGW_GRAD_MLP2_W = 3, KGW_GRAD_MLP2_B = 4, KGW_GRAD_G3T = 5 };
typedef struct KgwGradSrc {
    const float* ws;                 /* first partial record                                                     */
    int32_t kind, nblk;              /* KGW_GRAD_*; partial records per output element                           */
    int32_t M, N, MT, NT, gy, gz;    /* TN kinds: product shape and tiling; G3T: M = rows of the kgw_gemm3 product */
    int32_t K1, c_transposed;        /* MLP2 kinds: width of the first layer's input; TN: C stored transposed     */
    /* G3T only, set by the CALLER of kgw_adam_fused (NULL: off): the kgw_gemm3 operand image of the UPDATED parameter
     * ([128, M] = the B^T of the layer's forward product) is written by the same launch -- what kgw_gemm3_pack(S = parameter,
     * K = M, k_valid = M, s_is_kn = 0) would produce for the next forward, without that launch.                          */
    void* packed;
    int32_t flip, pad_;              /* sign period of the image in 32-k chunks (kgw_gemm3_flip()), 0 = none      */
} KgwGradSrc;

/* C[M,N] = A[rows,M]^T * B[rows,N] (row-major, leading dimensions lda/ldb/ldc), optionally
 * colsum_a[M] = column sums of A.  Split-K over the rows on fp32 MFMA, deterministic.  Replaces the
 * weight / bias gradient GEMMs autograd runs for the Linear layers of the path
 * (kgwas/model.py:13-21,50; kgwas/conv.py:82-89,150-151) whose reduction dimension is the number of
 * sampled nodes.  workspace: kgw_tn_gemm_workspace_floats(rows, M, N) floats.                    */
int64_t kgw_tn_gemm_workspace_floats(int64_t rows, int32_t M, int32_t N);
/* One product of kgw_tn_gemm_multi: the arguments of kgw_tn_gemm_ex as a record.                  */
typedef struct KgwTnJob {
    const float* A; int64_t lda; const float* B; int64_t ldb;
    int64_t rows;
    float* C; int64_t ldc;
    float* colsum_a; int64_t colsum_ld;
    float* workspace; int64_t workspace_floats;
    const int32_t* rows_dev;
    int32_t M, N, c_transposed, colsum_repeat;
} KgwTnJob;
/* Up to 4 such products in ONE pair of launches (the weight / bias gradients of the Linears of one
 * MLP, kgwas/model.py:17-21, become available together at the end of its backward).  Needs even M, N,
 * lda, ldb and 8-byte aligned A, B (else KGW_E_UNSUPPORTED: use kgw_tn_gemm_ex per product).        */
int kgw_tn_gemm_multi(int32_t n_jobs, const KgwTnJob* jobs, kgw_stream_t stream);
int kgw_tn_gemm(const float* A, int64_t lda, int32_t M, const float* B, int64_t ldb, int32_t N,
                int64_t rows, float* C, int64_t ldc, float* colsum_a, float* workspace,
                int64_t workspace_floats, kgw_stream_t stream);
/* Same product with two output options that let the gradient land where the caller keeps it (no copy / reduce
 * launches afterwards): c_transposed != 0 stores C^T, i.e. element (m,n) at C[n*ldc + m] (the packed per-relation
 * weights are kept transposed, [in, out]); colsum_a is written colsum_repeat times, copy q at
 * colsum_a + q*colsum_ld (the bias gradient of every relation summed into one destination type is the same
 * vector: HeteroConv sum, kgwas/model.py:74).  rows_dev (nullable, device int32): actual row count <= rows (static
 * capacity of a captured step); only those rows are read.                                                   */
int kgw_tn_gemm_ex(const float* A, int64_t lda, int32_t M, const float* B, int64_t ldb, int32_t N,
                   int64_t rows, float* C, int64_t ldc, int32_t c_transposed, float* colsum_a,
                   int32_t colsum_repeat, int64_t colsum_ld, float* workspace, int64_t workspace_floats,
                   const int32_t* rows_dev, kgw_stream_t stream);

/* The first launch of kgw_tn_gemm_ex / kgw_tn_gemm_multi only (colsum_repeat 1, dense C): src[0] / src[1] (per job: src[2q],
 * src[2q + 1]) describe the partial sums of the product / of the column sums for kgw_adam_fused; C and colsum_a are the
 * gradient tensors that launch will also fill.                                                                            */
int kgw_tn_gemm_partial(const float* A, int64_t lda, int32_t M, const float* B, int64_t ldb, int32_t N,
                        int64_t rows, float* C, int64_t ldc, int32_t c_transposed, float* colsum_a,
                        float* workspace, int64_t workspace_floats, const int32_t* rows_dev, KgwGradSrc* src,
                        kgw_stream_t stream);
int kgw_tn_gemm_multi_partial(int32_t n_jobs, const KgwTnJob* jobs, KgwGradSrc* src, kgw_stream_t stream);
/* The products that run on the 64 x 64-per-wavefront tiling (M, N, lda, ldb even, 8-byte aligned operands: every weight-gradient
 * product of the step but the narrow ones) use the BF16 matrix pipe with fp32 error since round 5: each operand value is split
 * exactly into three bf16 pieces and the six piece products of weight >= 2^-16 are accumulated in fp32 (the scheme of kgw_gemm3;
 * the dropped products are below the rounding of one fp32 multiply-add), 24 MFMAs of 32 cycles per 16 rows against 32 of 64 on
 * the fp32 pipe.  kgw_tn_split(0 / 1) selects the fp32 / bf16 pipe for later calls and returns the previous setting (< 0: query);
 * the environment variable KGW_TN_SPLIT=0 sets the initial value.                                                              */
int kgw_tn_split(int on);
/* Products of up to this many rows (and at least 16 output tiles) run as ONE row block whose blocks write the result directly: no
 * partial slabs, no second launch.  Default 0 (never): measured -3 us per step and one launch less at 2 048, but a wavefront's
 * seven-times longer sum then shows twice the error of the row-blocked form (on either pipe).  Sets the limit for later calls
 * (< 0: query), returns the previous one; KGW_TN_DIRECT_ROWS sets the initial value.                                           */
int64_t kgw_tn_direct_rows(int64_t rows);

/* A product group's SECOND launch (the sums over its row blocks: k_tn_reduce) that has not been issued.  The gradients it
 * finishes -- the relation transform's d W^T / d bias, kgwas/conv.py:138,190 -- feed nothing before the end of the backward pass,
 * so in a captured step its few blocks ride in a later launch instead of being one of their own (5 - 8 us each):
 *   kgw_transform_bwd_ex(.., ride_in, defer_out, ..)  kgw_transform_bwd that (ride_in, nullable) carries a pending plan's blocks
 *                                                      and (defer_out, nullable) leaves its own second launch as a plan;
 *   kgw_tn_gemm_partial_ride(.., ride_in, ..)          kgw_tn_gemm_partial that carries a pending plan's blocks;
 *   kgw_tn_reduce_launch(plan)                          the plan as a launch of its own (nothing came by to carry it).
 * Same blocks, same code, same order of additions wherever they run.  plan->valid == 0: nothing pending (a group whose products
 * all had one row block).  The plan points into the products' workspaces and outputs: keep them alive until it has run.
 * Riding pays for a SMALL plan only: the blocks take the carrier kernel's registers and LDS, so a carrier built for one or two
 * blocks per CU runs thousands of them a few at a time (measured: layer 1's 9 088 blocks in the SNP product's launch, 78 us against
 * 8 + 47 apart) -- ``blocks`` is there for the caller to decide.                                                               */
/* The read-out node's SECOND launch in a training step (kgw_readout_wmse_train: the single-block fold of the per-block partial sums
 * into d w_lin, d b_lin and the loss), not issued: nothing reads the three before the optimiser / the host, so in a captured step the
 * fold is one more block of the launch that follows (kgw_transform_bwd_ex's fold_in) instead of a 5 us launch of one block.
 * Filled by kgw_readout_wmse_train_parts; kgw_readout_train_fold runs it as a launch of its own.                                */
typedef struct KgwReadoutFold {
    const float* scratch; const double* terms; float* dw_lin; float* db_lin; double* loss;
    int32_t nb, n;
} KgwReadoutFold;

typedef struct KgwTnReducePlan { int32_t valid; int32_t blocks /* 256-thread blocks of the launch */; int32_t reserved[2]; int64_t opaque[96]; } KgwTnReducePlan;
int kgw_tn_reduce_launch(const KgwTnReducePlan* plan, kgw_stream_t stream);
int kgw_tn_gemm_partial_ride(const float* A, int64_t lda, int32_t M, const float* B, int64_t ldb, int32_t N, int64_t rows,
                             float* C, int64_t ldc, int32_t c_transposed, float* colsum_a, float* workspace,
                             int64_t workspace_floats, const int32_t* rows_dev, KgwGradSrc* src, const KgwTnReducePlan* ride_in,
                             kgw_stream_t stream);

/* Y[rows,N] = act(X[rows,K] * Wop + bias) * (mask > 0): the Linear layers of the path on fp32 MFMA.
 * w_is_kn = 0: W is [N,K] (nn.Linear / PyG Linear forward, kgwas/model.py:13-21,50; kgwas/conv.py:138,142);
 * w_is_kn = 1: W is [K,N] (the dX = dY * W product of their backward).  bias, mask may be NULL; relu 0/1.
 * K, ldx, ldw (and N when w_is_kn) must be multiples of 4, X and W 16-byte aligned, else KGW_E_UNSUPPORTED.
 * rows_dev (nullable, device int32): the number of rows the batch really has when `rows` is the static capacity of a
 * captured step (KgwBatchMeta.n_src / n_rows entry): rows [*rows_dev, rows) are not computed, Y gets zeros there. */
int kgw_linear(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias,
               const float* mask, int64_t ldm, float* Y, int64_t ldy, int64_t rows, int32_t K, int32_t N,
               int32_t relu, int32_t w_is_kn, const int32_t* rows_dev, kgw_stream_t stream);

/* SimpleMLP's two hidden layers (kgwas/model.py:17-20) for a NARROW input in one launch:
 * H2 = relu(relu(X W1^T + b1) W2^T + b2), X [rows, K1] with K1 <= 20 and K1 % 4 == 0 (the 20-wide SNP features), W1 [128, K1],
 * W2 [128, 128] (nn.Linear layout), hidden width 128.  The hidden state goes from the first product's accumulators straight
 * into the second product's operand registers; it is ALSO written to H1 (nullable) for the backward, which reads it as the
 * ReLU mask and in the weight gradient.  rows_dev: see kgw_linear.  ids (nullable): input row r is X[ids[r]] -- the
 * loader's x[n_id] slicing (kgwas/kgwas.py:135) folded in; Xg (nullable) then receives the gathered rows [rows, K1] for the
 * backward's first-layer weight gradient.                                                                          */
int kgw_mlp2_fwd(const float* X, int64_t ldx, int32_t K1, const float* W1, int64_t ldw1, const float* b1, const float* W2,
                 int64_t ldw2, const float* b2, float* H1, int64_t ldh1, float* H2, int64_t ldh2, int64_t rows,
                 const int32_t* rows_dev, const int32_t* ids, float* Xg, int64_t ldxg, kgw_stream_t stream);

/* The same two layers for a 128-WIDE input on few rows, the rows gathered from up to four resident feature matrices
 * (src[j] [*, 128] with row stride ldx, ids[j] [n_rows[j]]: the three GO node types share go_feat_mlp, kgwas/model.py:58-60):
 * rows of job 0, then job 1, ...  Outputs [sum n_rows, 128] with row stride ldo: Xg = the gathered input rows, H1, H2.    */
int kgw_mlp2w_fwd(int32_t n_jobs, const float* const* src, const int32_t* const* ids, const int64_t* n_rows, int64_t ldx,
                  const float* W1, int64_t ldw1, const float* b1, const float* W2, int64_t ldw2, const float* b2, float* Xg,
                  float* H1, float* H2, int64_t ldo, kgw_stream_t stream);

/* Backward of the NARROW first layer behind kgw_mlp2_fwd (no input gradient wanted): d W1 [128, K1] (row stride ldw1) and
 * d b1 [128] of  h1 = relu(x W1^T + b1)  given the upstream dH2 [rows, 128] of h2 = relu(h1 W2^T + b2) (already multiplied
 * by h2 > 0), W2, h1 and x [rows, K1 <= 31]: dh1 = (dH2 W2) * (h1 > 0) is formed tile by tile and consumed in place by the
 * d W1 product -- never written.  workspace: kgw_mlp2_bwd_first_workspace_floats(rows) floats.
 * Variant for a WIDE first layer computed on a resident feature matrix (the gene layer): in_ids [rows] (row r of the product
 * reads dH2[in_ids[r]], < 0: the node is not in the batch, its dh1 row is zero), dZ [rows, 128] receives the masked dh1 rows
 * (the layer's own weight gradient is a library product over them) and K1 = 0 leaves only d b1 (X, dW1 unused).        */
int64_t kgw_mlp2_bwd_first_workspace_floats(int64_t rows);
int kgw_mlp2_bwd_first(const float* dH2, int64_t ldd, const float* W2, int64_t ldw2, const float* H1, int64_t ldh1,
                       const float* X, int64_t ldx, int32_t K1, int64_t rows, const int32_t* rows_dev, float* dW1,
                       int64_t ldw1, float* db1, float* workspace, int64_t workspace_floats, const int32_t* in_ids, float* dZ,
                       int64_t ldz, kgw_stream_t stream);

/* The same without the second launch (the fold of the blocks' partial d W1 / d b1): src[0] describes d W1's partial sums,
 * src[1] d b1's, for kgw_adam_fused (ldw1 must be K1: a dense gradient tensor).                                             */
int kgw_mlp2_bwd_first_partial(const float* dH2, int64_t ldd, const float* W2, int64_t ldw2, const float* H1, int64_t ldh1,
                               const float* X, int64_t ldx, int32_t K1, int64_t rows, const int32_t* rows_dev, float* dW1,
                               int64_t ldw1, float* db1, float* workspace, int64_t workspace_floats, const int32_t* in_ids,
                               float* dZ, int64_t ldz, KgwGradSrc* src, kgw_stream_t stream);
/* ... that ALSO writes the masked dh1 rows as kgw_gemm3's B operand image (packed: kgw_gemm3_packed_bytes(rows rounded up to 32)
 * bytes, the s_is_kn form of kgw_gemm3_pack; flip = kgw_gemm3_flip()): the resident first layer's weight gradient
 * (kgwas/model.py:13 under loss.backward(), dW1 = dh1^T X on kgw_gemm3) then needs neither the kgw_gemm3_pack launch nor the fp32
 * rows -- dZ may be null.  src nullable (null: the fold launch runs here).  KGW_E_UNSUPPORTED: the fp32-pipe variant is selected.  */
int kgw_mlp2_bwd_first_packed(const float* dH2, int64_t ldd, const float* W2, int64_t ldw2, const float* H1, int64_t ldh1,
                              const float* X, int64_t ldx, int32_t K1, int64_t rows, float* dW1, int64_t ldw1, float* db1,
                              float* workspace, int64_t workspace_floats, const int32_t* in_ids, float* dZ, int64_t ldz,
                              void* packed, int32_t flip, KgwGradSrc* src, kgw_stream_t stream);

/* C[M, 128] = A[M, K] B[K, 128] for a tall RESIDENT fp32 matrix A -- the first gene Linear, kgwas/model.py:13,19 over the
 * 5 120-wide gene features (kgwas_data.py:236,244): forward with A = X [genes, K], B = W1^T (out = relu(C + bias)), and its
 * weight gradient with A = X^T [K, genes], B = dz (transpose_out: out[n, m] = C[m, n], i.e. dW1 [128, K]).  Runs on the bf16
 * matrix pipe with fp32 error: every fp32 operand is split exactly into three bf16 pieces and the six piece products of
 * weight >= 2^-16 are accumulated in fp32; the dropped products are below the rounding of one fp32 multiply-add (see
 * kgwas_amd/csrc/kgw_gemm3.hip).  kgw_gemm3_pack splits B once per call into the kernel's operand image (packed:
 * kgw_gemm3_packed_bytes(K) bytes; s_is_kn: B[k, n] = S[k * lds + n], else B[k, n] = S[n * lds + k], n < 128; S holds
 * k < k_valid only, B[k >= k_valid] = 0: a gene count / feature width that is not a multiple of 32 is padded HERE and in a
 * zero-padded resident A, never by falling back to a library product).  A is split in
 * the kernel.  lda == 0: A is stored in 32 x 32 tiles, [ceil(M / 32)][K / 32][32][32] floats (rows past M present, any value) --
 * the layout for a resident copy, every 4 KB a wavefront reads per step is contiguous.  K % 32 == 0, lda % 4 == 0, 16-byte
 * aligned pointers, else KGW_E_UNSUPPORTED.  workspace:
 * kgw_gemm3_workspace_floats(M, K) floats (partial products of the K ranges, added in index order: deterministic).      */
int64_t kgw_gemm3_packed_bytes(int64_t K);
int64_t kgw_gemm3_workspace_floats(int64_t M, int64_t K);
int kgw_gemm3_pack(const float* S, int64_t lds, int64_t K, int64_t k_valid, int32_t s_is_kn, void* packed, kgw_stream_t stream);
int kgw_gemm3(const float* A, int64_t lda, int64_t M, int64_t K, const void* packed, float* workspace, int64_t workspace_floats,
              const float* bias, int32_t relu, float* out, int64_t ldo, int32_t transpose_out, const int32_t* row_map,
              float* out_rows, int64_t ld_rows, int64_t out_rows_n, const int32_t* out_rows_real, kgw_stream_t stream);
/* row_map (nullable, not with transpose_out) [M]: row m of the result is ALSO written to out_rows[row_map[m]] when
 * row_map[m] >= 0 -- the loader's x[n_id] slicing of the layer's output (kgwas/kgwas.py:135) without a gather launch.
 * out_rows_real (nullable, device int32) with out_rows_n (<= M): out_rows is a row block of out_rows_n rows of which only the
 * first *out_rows_real are the batch's (static capacity of a captured step); the rest is written as zeros by the same launch. */

/* The same product for FEW rows when one of K, N is 128 and the other a multiple of 128 -- the per-relation transform
 * of a layer after aggregate-then-transform, [N_dst, R*128] x [R*128, 128] with N_dst ~ 0.5-1.2 k destination rows of a
 * 512-seed batch (kgwas/conv.py:138-144 for all relations into one destination type + bias :190 + HeteroConv sum
 * model.py:74 + ReLU :75), and its dZ twin [N_dst, 128] x [128, R*128]: the long dimension is cut into 128-wide slabs
 * (= relations); a block keeps its slab's 128 x 128 weights stationary in MFMA operand registers and streams 32-row
 * tiles through LDS; K slabs are added in order by a second launch (deterministic).  workspace:
 * kgw_linear_splitk_workspace_floats floats (0 when K == 128).  Other shapes: KGW_E_UNSUPPORTED (use kgw_linear).   */
int64_t kgw_linear_splitk_workspace_floats(int64_t rows, int32_t K, int32_t N);
int kgw_linear_splitk(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y,
                      int64_t ldy, int64_t rows, int32_t K, int32_t N, int32_t relu, int32_t w_is_kn,
                      float* workspace, int64_t workspace_floats, const int32_t* rows_dev, kgw_stream_t stream);
/* The forward transform (N == 128, K = R*128) with a per-segment constant: Y[i] = act(X[i] W + bias + sum_r ind(i, r)
 * gamma[r]), ind(i, r) = seg_stat[2 (i R + r) + 1] > 0 -- segment (destination row i, relation slot r) has at least one
 * edge (its softmax denominator as kgw_gat_aggregate_fwd left it).  With FC_output folded into layer 1 the aggregate
 * works on the MLP's hidden state h2 and the bias c of the folded Linear re-enters as c W_r wherever the attention
 * weights of a segment sum to 1, i.e. wherever it is not empty.  gamma [R][128].  kgw_ind_colsum is the backward:
 * dgamma[r] = sum_i ind(i, r) dY[i]  (one block per relation slot, rows added in order: deterministic).           */
int kgw_linear_splitk_ind(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y,
                          int64_t ldy, int64_t rows, int32_t K, int32_t relu, const float* seg_stat, const float* gamma,
                          float* workspace, int64_t workspace_floats, const int32_t* rows_dev, kgw_stream_t stream);
int kgw_ind_colsum(const float* seg_stat, const float* dY, int64_t ldy, int64_t rows, int32_t R, float* dgamma,
                   kgw_stream_t stream);
/* One product of kgw_linear_splitk_multi / one sum of kgw_ind_colsum_multi: the arguments above as a record.  A layer has one
 * transform per DESTINATION TYPE (kgwas/model.py:74: HeteroConv sums the relations into each type) -- genes and seed SNPs in
 * layer 1 of a 512-seed batch -- and each of them is a problem of a few hundred rows that leaves most of the chip idle: all of
 * a layer's forward transforms go in ONE launch, all its dZ twins in one, all its d gamma sums in one.                        */
typedef struct KgwSplitKJob {
    const float* X; int64_t ldx;
    const float* W; int64_t ldw;
    const float* bias;            /* nullable */
    float* Y; int64_t ldy;
    int64_t rows;
    const float* seg_stat;        /* forward with a per-segment constant (kgw_linear_splitk_ind), else NULL; for
                                     kgw_ind_colsum_multi: the segment statistics (required)                       */
    const float* gamma;           /* with seg_stat (forward); kgw_ind_colsum_multi: unused                       */
    float* dgamma;                /* kgw_ind_colsum_multi: output [K / 128][128]; X / W / bias / gamma unused, Y = dY */
    int32_t K, N, relu, w_is_kn;
} KgwSplitKJob;
/* Up to 4 products in one launch.  Every job must be of the SAME kind: either the forward transform (K > 128 a multiple of
 * 128, N == 128, w_is_kn = 1: what kgw_linear_splitk / _ind run as ONE launch) or the dZ twin (K == 128, N a multiple of 128,
 * no seg_stat); anything else: KGW_E_UNSUPPORTED (use the single-product calls).                                           */
int kgw_linear_splitk_multi(int32_t n_jobs, const KgwSplitKJob* jobs, kgw_stream_t stream);
/* dgamma of up to 4 transforms in one launch (job: seg_stat, Y = dY, ldy, rows, K = R * 128, dgamma).                        */
int kgw_ind_colsum_multi(int32_t n_jobs, const KgwSplitKJob* jobs, kgw_stream_t stream);

/* The backward of a layer's relation transform (kgwas/conv.py:138-144,190 + kgwas/model.py:74-75 under loss.backward()) in ONE
 * launch (+ the split-K products' second): everything that is a function of d(output) alone -- the weight / bias gradients of
 * every destination type (tn_jobs: the records of kgw_tn_gemm_multi), the dZ twins (sk_jobs: kgw_linear_splitk_multi's K == 128
 * records with [N, K] weights) and the d gamma sums of a folded layer (cs_jobs: kgw_ind_colsum_multi's records) -- as blocks of
 * one grid; values identical to the three separate calls.  Any of the three lists may be empty.                              */
int kgw_transform_bwd(int32_t n_tn, const KgwTnJob* tn_jobs, int32_t n_sk, const KgwSplitKJob* sk_jobs, int32_t n_cs,
                      const KgwSplitKJob* cs_jobs, kgw_stream_t stream);
/* ... with the pending second launch of an EARLIER product group riding as extra blocks (ride_in) and / or this call's own second
 * launch left pending (defer_out): see KgwTnReducePlan.                                                                         */
int kgw_transform_bwd_ex(int32_t n_tn, const KgwTnJob* tn_jobs, int32_t n_sk, const KgwSplitKJob* sk_jobs, int32_t n_cs,
                         const KgwSplitKJob* cs_jobs, const KgwTnReducePlan* ride_in, KgwTnReducePlan* defer_out,
                         const KgwReadoutFold* fold_in /* nullable: see KgwReadoutFold */, kgw_stream_t stream);

/* One Adam step (torch.optim.Adam semantics, weight_decay as L2: kgwas/kgwas.py:116,151) over up to 64
 * parameter tensors in a single launch.  The pointer arrays are HOST arrays of device pointers (passed to the
 * kernel by value); step_dev is a device int32 counter incremented by the call (graph-capturable).        */
int kgw_adam(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
             float* const* exp_avg_sq, const int64_t* numel, int32_t* step_dev, float lr, float beta1,
             float beta2, float eps, float weight_decay, kgw_stream_t stream);
/* The same, leaving *step_dev alone: the caller advances it after this call (kgw_accumulate_stats_tick).            */
int kgw_adam_notick(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int64_t* numel, int32_t* step_dev, float lr, float beta1,
                    float beta2, float eps, float weight_decay, kgw_stream_t stream);

/* The weight-gradient form of kgw_gemm3 (transpose_out: out [128, M], ldo == M, M a multiple of 32) without its range-sum
 * launch: src describes the K ranges' partial products for kgw_adam_fused.  kgw_gemm3_flip(): the sign period kgw_gemm3_pack
 * builds its images with (KgwGradSrc.flip).                                                                                */
int kgw_gemm3_partial(const float* A, int64_t lda, int64_t M, int64_t K, const void* packed, float* workspace,
                      int64_t workspace_floats, float* out, int64_t ldo, KgwGradSrc* src, kgw_stream_t stream);
int kgw_gemm3_flip(void);

/* The optimiser launch of a captured training step: kgw_adam_notick over up to KGW_ADAM_FUSED_MAX tensors, and in the SAME launch
 * (a) the last reduction of every gradient whose producer left partial sums (src[i].kind != KGW_GRAD_DIRECT, at most
 *     KGW_ADAM_FUSED_SRC of them; src == NULL: all direct) -- the sums are taken in the producers' own order and also written
 *     to grads[i], so the gradient tensors hold what the unfused path leaves in them;
 * (b) kgw_accumulate_stats_tick (meta_dev != NULL: the running totals; the step counter advances in any case) -- done by the
 *     block that finishes last (done_counters: KGW_ADAM_FUSED_COUNTERS device int32 the caller zeroes once; the launch leaves
 *     them at zero).                                                                                                         */
enum { KGW_ADAM_FUSED_MAX = 40, KGW_ADAM_FUSED_SRC = 12, KGW_ADAM_FUSED_COUNTERS = 65 * 32 };
int kgw_adam_fused(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const int64_t* numel, const KgwGradSrc* src, int32_t* step_dev, float lr,
                   float beta1, float beta2, float eps, float weight_decay, const KgwBatchMeta* meta_dev, int32_t n_layers,
                   int32_t n_hops, int64_t* stats, int32_t* done_counters, kgw_stream_t stream);

/* The same work units WITHOUT the update, for the multi-GPU step (the ranks' gradients must be complete before the all-reduce):
 * dst[i] (the tensor's slot in a flat gradient bucket) = the finished gradient of tensor i -- a copy of grads[i] where src[i] is
 * KGW_GRAD_DIRECT (or src NULL), the sum of the producer's partial records otherwise (then also stored into grads[i]).  One launch
 * instead of the producers' second launches + the bucket's concatenation.                                                      */
int kgw_grad_finish(int32_t n_tensors, float* const* dst, float* const* grads, const int64_t* numel, const KgwGradSrc* src,
                    kgw_stream_t stream);

/* Attention vectors of all relations of a layer (kgwas/conv.py:138-151 reduced to what the path consumes):
 * U_full[r] = W_src^T att_src for every relation id r the layer computes (live_of_rel[r] = its index i in the
 * packed parameter arrays, -1 => row of zeros), V[i] = W_dst^T att_dst (bip_pos[i] >= 0: index into w_dst_t) or
 * W_src^T att_dst (same-type relation).  Packed weights are transposed: w_*_t[i][k][c] = W_i[c][k], C = 128.
 * v_by_rel != 0: V / dV are [n_rels_total][128] indexed by relation id like U_full (zero rows for relations the
 * layer does not compute) -- the form kgw_gat_aggregate_* consume.
 * Optional (bias != NULL): bias_sum[b] = sum of bias[i] over the packed relations i with blk_of_live[i] == b, the
 * bias of the relation-summed output of destination block b (HeteroConv sum of conv.py:190's bias, model.py:74).
 * Optional (zero_buf != NULL): zero_floats (multiple of 4) floats at zero_buf are cleared by extra blocks of the same
 * launch -- the Z / stat / d a_dst workspace of the kgw_gat_aggregate_fwd call that follows (it needs them zeroed).
 * _bwd: gradients of the packed parameters from (dU_full, dV); every output element is written.          */
int kgw_relvec_fwd(int32_t n_rels_total, const int32_t* live_of_rel, const int32_t* bip_pos, const float* w_src_t,
                   const float* w_dst_t, const float* att_src, const float* att_dst, float* U_full, float* V,
                   int32_t v_by_rel, int32_t n_live, const float* bias, const int32_t* blk_of_live, int32_t n_blk,
                   float* bias_sum, float* zero_buf, int64_t zero_floats, kgw_stream_t stream);
int kgw_relvec_bwd(int32_t n_live, const int32_t* rel_ids, const int32_t* bip_pos, const float* w_src_t,
                   const float* w_dst_t, const float* att_src, const float* att_dst, const float* dU_full,
                   const float* dV, float* dw_src_t, float* dw_dst_t, float* datt_src, float* datt_dst,
                   int32_t v_by_rel, kgw_stream_t stream);
/* The relation vectors of SEVERAL layers in one launch (and their backward in one): they depend on the parameters only, a
 * model has one pack per layer (kgwas/model.py:47: one HeteroConv per layer), and each launch is ~30 blocks of latency.  A job =
 * the arguments of kgw_relvec_fwd (v_by_rel = 1) / kgw_relvec_bwd_acc as a record; the forward reads the first group of fields,
 * the backward the second.  Up to KGW_MAX_LAYERS jobs.                                                                  */
typedef struct KgwRelvecJob {
    int32_t n_rels_total, n_live, n_blk;
    int32_t duv_pieces;              /* backward, != 0: dU_full / dV point at [rows][8][128] pieces (KGW_F_DUV_PIECES), added here
                                        in k_duv_fold's order                                                                 */
    const int32_t* live_of_rel; const int32_t* rel_ids; const int32_t* bip_pos;
    const float* w_src_t; const float* w_dst_t; const float* att_src; const float* att_dst;
    /* forward */
    float* U_full; float* V; const float* bias; const int32_t* blk_of_live; float* bias_sum;
    float* zero_buf; int64_t zero_floats;
    /* backward (dU_full / dV / dw_src_acc nullable = zero) */
    const float* dU_full; const float* dV; const float* dw_src_acc;
    float* dw_src_t; float* dw_dst_t; float* datt_src; float* datt_dst;
} KgwRelvecJob;
int kgw_relvec_fwd_multi(int32_t n_jobs, const KgwRelvecJob* jobs, kgw_stream_t stream);
int kgw_relvec_bwd_multi(int32_t n_jobs, const KgwRelvecJob* jobs, kgw_stream_t stream);
/* _bwd_acc: the same with (a) dw_src_acc (nullable): a gradient of w_src_t that reached the caller by another path -- the
 * layer's transform GEMM (conv.py:138-144) or the FC_output fold use the same weights -- and is added into dw_src_t here
 * instead of by a separate launch; (b) dU_full / dV nullable = zero.                                              */
int kgw_relvec_bwd_acc(int32_t n_live, const int32_t* rel_ids, const int32_t* bip_pos, const float* w_src_t,
                       const float* w_dst_t, const float* att_src, const float* att_dst, const float* dU_full,
                       const float* dV, const float* dw_src_acc, float* dw_src_t, float* dw_dst_t, float* datt_src,
                       float* datt_dst, int32_t v_by_rel, kgw_stream_t stream);

/* FC_output of the feature MLPs folded into the layer-1 relation parameters (exact re-association; no counterpart call in
 * the reference, which runs SimpleMLP.FC_output, kgwas/model.py:15,21, on every sampled node).  H = h2 T + c with
 * T = FC_output.weight^T enters GATConv only through <H_j, u_r>, <H_i, v_r> and sum_j alpha_ij H_j (kgwas/conv.py:150-152,
 * 227-228), so layer 1 runs on h2 with
 *     U'_r = T_src U_r,  V'_r = T_dst V_r,  kappa_r = <c_src, U_r> + <c_dst, V_r>,  W'_i = T_src W_i^T,  gamma_i = c_src W_i^T
 * (src / dst: the MLP of the relation's source / destination node type).  n packed relations (slot i = relation id
 * rel_ids_host[i]) of n_rels; n_mlp <= 4 MLPs.  U, V / U', V' / kappa are indexed by relation id (rows of relations
 * outside the pack are written as zeros), W_i^T / W'_i / gamma_i by packed slot.  _fwd fills Up, Vp, kappa, Wp, gamma;
 * _bwd fills dU, dV (feed kgw_relvec_bwd), dws (the fold's share of d W_i^T), d_fc_weight[m], d_fc_bias[m] from dUp, dVp,
 * dkappa, dWp, dgamma -- every element written, fixed summation order.  *_host arrays are host int32 arrays.          */
typedef struct KgwFoldArgs {
    int32_t n, n_rels, n_mlp;
    int32_t duv_pieces;              /* kgw_fold_bwd, != 0: dUp / dVp point at [n_rels][8][128] pieces (KGW_F_DUV_PIECES)        */
    const int32_t* rel_ids_host; const int32_t* src_mlp_host; const int32_t* dst_mlp_host;
    const float* w_src_t;            /* [n][128][128]                                   */
    const float* fc_weight[4];       /* FC_output.weight [128 out][128 in] of MLP m     */
    const float* fc_bias[4];         /* FC_output.bias [128]                            */
    const float* U; const float* V;  /* [n_rels][128]                                   */
    float* Up; float* Vp; float* kappa; float* Wp; float* gamma;
    const float* dUp; const float* dVp; const float* dkappa; const float* dWp; const float* dgamma;
    float* dU; float* dV; float* dws;
    float* d_fc_weight[4]; float* d_fc_bias[4];
} KgwFoldArgs;
int kgw_fold_fwd(const KgwFoldArgs* args, kgw_stream_t stream);
int kgw_fold_bwd(const KgwFoldArgs* args, kgw_stream_t stream);

/* The END of a captured step's backward pass as ONE launch: the deferred weight-gradient products of the MLPs (n_tn jobs + their
 * 2 n_tn records: kgw_tn_gemm_multi_partial's arguments), kgw_fold_bwd (fold, nullable) and kgw_relvec_bwd_multi (n_relvec jobs) as
 * the blocks of one grid -- parameter-only work that nothing but the optimiser waits for and of which no launch fills the chip.
 * Job fold_job of relvec is the fold's layer: its dU_full / dV / dw_src_acc are NOT read -- the blocks of that layer compute the
 * fold's d U_r, d V_r and d W share themselves and hand them on through LDS -- and fold->dU / dV / dws are not written.  Every value
 * is computed with the expressions and in the order of the kernel it comes from (bit-identical to the three launches).
 * KGW_E_UNSUPPORTED (nothing launched): a product outside the 64 x 64-per-wavefront tiling, an empty job.                       */
int kgw_param_tail(int32_t n_tn, const KgwTnJob* tn_jobs, KgwGradSrc* src, const KgwFoldArgs* fold, int32_t n_relvec,
                   const KgwRelvecJob* relvec, int32_t fold_job, kgw_stream_t stream);

/* kgw_gemm3 with RIDERS: the launch carries the step's parameter-only forward work as extra blocks on the compute units the
 * product leaves idle (the benchmark's forward product: 240 one-CU blocks on 256 CUs for ~145 us) -- what kgw_relvec_fwd_multi
 * (n_relvec jobs: the relation vectors of every layer, summed biases, the aggregates' zero fills; kgwas/conv.py:138-151) and
 * kgw_fold_fwd (fold, nullable: FC_output into the relations of job fold_job, kgwas/model.py:15,21; its U / V must be that
 * job's U_full / V) compute, with the same arithmetic (bit-identical values), one wavefront per task, no launch of their own.
 * kgw_gemm3_rider_blocks(M, K): the idle compute units such a launch has (0: none worth using -- kgw_gemm3_riders then returns
 * KGW_E_UNSUPPORTED before launching anything and the caller runs the riders' own kernels).                                   */
int kgw_gemm3_rider_blocks(int64_t M, int64_t K);

int kgw_gemm3_riders(const float* A, int64_t lda, int64_t M, int64_t K, const void* packed, float* workspace, int64_t workspace_floats,
                     const float* bias, int32_t relu, float* out, int64_t ldo, int32_t transpose_out, const int32_t* row_map,
                     float* out_rows, int64_t ld_rows, int64_t out_rows_n, const int32_t* out_rows_real, int32_t n_relvec,
                     const KgwRelvecJob* relvec, const KgwFoldArgs* fold, int32_t fold_job, kgw_stream_t stream);

/* LD-score weighted MSE over the seed rows (kgwas/kgwas.py:139-145): loss = mean_i w[n_id[i]] * (pred[i] - y[n_id[i]])^2,
 * float32 residual / square, float64 weight and mean; _bwd: dpred[i] = grad_loss * d loss / d pred[i].
 * y [N] float32 labels and w [N] float64 weights are indexed by GLOBAL node id, n_id [n] = the seeds' global ids. */
int kgw_wmse_fwd(const float* pred, const int32_t* n_id, const float* y, const double* w, int32_t n,
                 double* loss, kgw_stream_t stream);
int kgw_wmse_bwd(const float* pred, const int32_t* n_id, const float* y, const double* w, int32_t n,
                 const double* grad_loss, float* dpred, kgw_stream_t stream);

/* The training step's tail as two launches: read-out pred[i] = [relu](<H[i], w_lin> + b_lin) of the n seed rows
 * (HeteroGNN.lin, kgwas/model.py:50,83-86; hidden 128 -> 1) fused with the weighted MSE above; _bwd writes dH [rows][128]
 * (zero beyond the seeds), d w_lin [128] and d b_lin [1].  `relu` bit 0: ReLU on pred; bit 1 (_bwd): H is itself a ReLU
 * output whose backward is folded in (dH *= H > 0).  One wavefront per seed; partial results go to `scratch`
 * (_fwd: n doubles; _bwd: ceil(rows/4) * 129 floats) and a second single-block launch folds them in index order
 * (deterministic).                                                                                              */
int kgw_readout_wmse_fwd(const float* H, const float* w_lin, const float* b_lin, const int32_t* n_id,
                         const float* y, const double* w, int32_t n, int32_t relu, float* pred, double* loss,
                         double* scratch, kgw_stream_t stream);
int kgw_readout_wmse_bwd(const float* H, const float* w_lin, const float* pred, const int32_t* n_id,
                         const float* y, const double* w, int32_t n, int64_t rows, int32_t relu,
                         const double* grad_loss, float* dH, float* dw_lin, float* db_lin, float* scratch,
                         kgw_stream_t stream);

/* The two calls above fused for a training step whose loss gradient is 1 (loss.backward() of kgwas/kgwas.py:147):
 * prediction, loss, dH (zero beyond the seeds), d w_lin, d b_lin in two launches instead of four.  relu: bit 0 = ReLU
 * read-out, bit 1 = H is the output of a ReLU whose backward is folded in (dH *= H > 0).  terms [n] doubles and
 * scratch [((rows+3)/4) * 129] floats are workspaces.                                                             */
int kgw_readout_wmse_train(const float* H, const float* w_lin, const float* b_lin, const int32_t* n_id, const float* y,
                           const double* w, int32_t n, int64_t rows, int32_t relu, float* pred, double* loss, float* dH,
                           float* dw_lin, float* db_lin, double* terms, float* scratch, kgw_stream_t stream);
/* kgw_readout_wmse_train's first launch only; *fold_out describes the second (KgwReadoutFold).                                 */
int kgw_readout_wmse_train_parts(const float* H, const float* w_lin, const float* b_lin, const int32_t* n_id, const float* y,
                                 const double* w, int32_t n, int64_t rows, int32_t relu, float* pred, double* loss, float* dH,
                                 float* dw_lin, float* db_lin, double* terms, float* scratch, KgwReadoutFold* fold_out,
                                 kgw_stream_t stream);
int kgw_readout_train_fold(const KgwReadoutFold* fold, kgw_stream_t stream);

/* Self-test of the cross-lane reductions used by the aggregate kernels (one wavefront):
 * out_half[l] = sum over l's 32-lane half, out_wave[l] = sum over the wavefront,
 * out_steps[4][64] = the four intra-row DPP butterfly stages.                                  */
int kgw_debug_reduce(const float* in, float* out_half, float* out_wave, float* out_steps,
                     kgw_stream_t stream);
/* Self-test of the transposed 8-way reduction primitives (kgw_common.h): in [64][8]; out [6][64] = half_reduce8,
 * max8, sum8, bcast8<3>, xor4, xor8 of (in[lane][*] resp. in[lane][0]).                                          */
int kgw_debug_reduce8(const float* in, float* out, kgw_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KGWAS_HIP_H */
