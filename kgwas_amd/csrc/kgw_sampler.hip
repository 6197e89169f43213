// kgw_sampler.hip -- device-resident full-neighbourhood minibatch sampler for KGWAS (gfx950).
//
// Replaces the work PyG's NeighborLoader(num_neighbors=[-1]*L) does on the CPU for the reference
// (kgwas/kgwas.py:99-113,129): hop-by-hop expansion over ALL in-neighbours of every relation,
// global->local relabelling (seeds first), per-relation local CSR -- plus what the fused kernels
// need and PyG never builds: a chunk list (<= KGW_CHUNK edges of one destination row each, so hub
// rows are split across wavefronts) and the src-major (transposed) structure of every layer for an
// atomics-free backward.
//
// Everything is HBM-bound integer work: coalesced reads of the resident CSR (rowptr/col), byte-free
// int32 maps, prefix sums.  All kernels are grid-stride over DEVICE-side counts (KgwBatchMeta), so
// the whole batch is enqueued without a single host round trip; the host reads KgwBatchMeta once,
// after the final async D2H copy.
//
// Local node order: per node type, hop-major; hop 0 = seeds in seed order, later hops sorted by
// global id (flag + prefix-sum compaction => deterministic).  PyG's order for non-seed nodes is
// first-seen; only the seeds-first contract is consumed by the reference (kgwas/model.py:86).
#include "kgw_common.h"

namespace {

struct SampArgs {
    KgwGraph G;        // by value: descriptor reads become scalar loads from the kernarg segment
    KgwBatchBuf B;
};

// ---- small helpers -----------------------------------------------------------------------------
__device__ __forceinline__ int find_rel(const int32_t* seg_off, int n_rels, int sigma) {
    // seg_off[0..n_rels] ascending; return r with seg_off[r] <= sigma < seg_off[r+1]
    int lo = 0, hi = n_rels;            // invariant: seg_off[lo] <= sigma < seg_off[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (seg_off[mid] <= sigma) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int type_of_node_slot(const KgwGraph& G, int i) {
    int t = 0;
    while (t + 1 < G.n_types && i >= G.node_base[t + 1]) ++t;
    return t;
}

// ---- fills / meta export as KERNELS (not hipMemsetAsync / hipMemcpyAsync) ------------------------------
// The whole batch must be capturable in a HIP graph together with framework work that allocates from a
// graph-private pool; on ROCm 7.2 a captured graph that mixes memset / D2H-memcpy nodes issued from here with
// such allocations faulted on its second replay, while kernel nodes replay fine.  int4 stores, grid-stride.
__global__ void __launch_bounds__(KGW_BLK) k_fill_i32(int32_t* __restrict__ p, int32_t v, int64_t n) {
    const int64_t tid = (int64_t)blockIdx.x * KGW_BLK + threadIdx.x, nthr = (int64_t)gridDim.x * KGW_BLK;
    const int64_t n4 = n >> 2;
    int4* p4 = (int4*)p;
    const int4 v4 = make_int4(v, v, v, v);
    for (int64_t i = tid; i < n4; i += nthr) p4[i] = v4;
    for (int64_t i = (n4 << 2) + tid; i < n; i += nthr) p[i] = v;
}

inline int fill_i32(int32_t* p, int32_t v, int64_t n, hipStream_t st, int max_blocks = KGW_GRID) {
    if (n <= 0) return KGW_OK;
    int64_t g = (n / 4 + KGW_BLK - 1) / KGW_BLK;
    if (g > max_blocks) g = max_blocks;
    if (g < 1) g = 1;
    k_fill_i32<<<(int)g, KGW_BLK, 0, st>>>(p, v, n);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? KGW_OK : (int)e;
}

// device meta -> pinned host mirror (zero-copy store; visible to the host once the stream has drained)
__global__ void k_meta_to_host(const KgwBatchMeta* __restrict__ src, KgwBatchMeta* __restrict__ dst) {
    const int n = sizeof(KgwBatchMeta) / sizeof(int32_t);
    const int32_t* s = (const int32_t*)src;
    int32_t* d = (int32_t*)dst;
    for (int i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
    __threadfence_system();
}

// ---- init --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(KGW_BLK) k_init(SampArgs A, const int64_t* seeds, int n_seeds,
                                                  int seed_type, int full) {
    const KgwGraph& G = A.G;
    KgwBatchMeta* M = A.B.meta;
    const int64_t tid = (int64_t)blockIdx.x * KGW_BLK + threadIdx.x;
    const int64_t nthr = (int64_t)gridDim.x * KGW_BLK;
    if (full) {
        const int total = G.node_base[G.n_types];
        for (int64_t i = tid; i < total; i += nthr) {
            int t = type_of_node_slot(G, (int)i);
            int g = (int)i - G.node_base[t];
            if (g < G.n_nodes[t]) { A.B.g2l[i] = g; A.B.n_id[i] = g; }
        }
    } else {
        const int base = G.node_base[seed_type];
        for (int64_t i = tid; i < n_seeds; i += nthr) {
            int g = (int)seeds[i];
            A.B.g2l[base + g] = (int)i;
            A.B.n_id[base + i] = g;
        }
    }
    if (tid == 0) {
        for (int t = 0; t < G.n_types; ++t) {
            int c = full ? G.n_nodes[t] : (t == seed_type ? n_seeds : 0);
            M->hop_cnt[t][0] = c;
            M->node_off[t][0] = 0;
            M->node_off[t][1] = c;
        }
    }
}

// ---- per hop: segment bookkeeping --------------------------------------------------------------
// Segment offsets of hop h (a prefix over <= 64 relations) are recomputed by EVERY block into LDS -- scalar work of a
// microsecond -- instead of by a one-thread launch of their own; block 0 also publishes them (KgwBatchMeta.seg_off / seg_end /
// cur) for the kernels that follow.
__device__ __forceinline__ void hop_offsets(const SampArgs& A, int h, int* s_off, int* s_cur, bool publish) {
    const KgwGraph& G = A.G;
    KgwBatchMeta* M = A.B.meta;
    int s = (h == 0) ? 0 : M->seg_end[h - 1];
    const int begin = s;
    for (int r = 0; r < G.n_rels; ++r) {
        s_off[r] = s;
        s += M->hop_cnt[G.rel_dst[r]][h];
    }
    s_off[G.n_rels] = s;
    const bool over = (int64_t)s > A.B.seg_cap;
    if (publish) {
        for (int r = 0; r <= G.n_rels; ++r) M->seg_off[h][r] = s_off[r];
        M->seg_end[h] = s;
        if (over) M->error |= 1;
    }
    if (over) s = begin;                              // empty range: nothing runs past capacity
    s_cur[0] = begin; s_cur[1] = s;
    if (publish) {
        M->cur[0] = begin;   // scan range [cur0, cur1)
        M->cur[1] = s;
        M->cur[2] = (h == 0) ? 0 : M->edge_end[h - 1];    // carry-in for seg_ptr
        M->cur[3] = (h == 0) ? 0 : M->chunk_end[h - 1];   // carry-in for seg_chptr
    }
}

__global__ void __launch_bounds__(KGW_BLK) k_seg_deg(SampArgs A, int h) {
    __shared__ int s_off[KGW_MAX_RELS + 1], s_cur[2];
    const KgwGraph& G = A.G;
    const KgwBatchMeta* M = A.B.meta;
    if (threadIdx.x == 0) hop_offsets(A, h, s_off, s_cur, blockIdx.x == 0);
    __syncthreads();
    const int begin = s_cur[0], end = s_cur[1];
    for (int sg = begin + blockIdx.x * KGW_BLK + threadIdx.x; sg < end; sg += gridDim.x * KGW_BLK) {
        int r = find_rel(s_off, G.n_rels, sg);
        int d = G.rel_dst[r];
        int li = M->node_off[d][h] + (sg - s_off[r]);
        int g = A.B.n_id[G.node_base[d] + li];
        const int32_t* rp = G.g_rowptr + G.rowptr_off[r];
        int deg = rp[g + 1] - rp[g];
        A.B.seg_deg[sg] = deg;
        A.B.seg_nch[sg] = (deg + KGW_CHUNK - 1) / KGW_CHUNK;
    }
}

// ---- ONE-block exclusive scan of a device-side range (the segments of a minibatch hop: a few 10 k entries) ------------
// Replaces the three launches of the tiled scan below where the range is small: 1024 threads walk the range in tiles of 4096,
// one int4 per thread (coalesced), scan inside the thread, the wavefront (shuffles) and across the 16 wavefronts (LDS), carry
// from tile to tile.  K arrays together; carries in cur[2 + k], totals out to cur[4 + k] and the end sentinels.
template <int K>
__global__ void __launch_bounds__(1024) k_scan_block(const int32_t* in0, const int32_t* in1, int32_t* out0, int32_t* out1,
                                                     KgwBatchMeta* M, int use_carry) {
    __shared__ int s_w[16], s_tot;
    const int begin = M->cur[0], end = M->cur[1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int k = 0; k < K; ++k) {
        const int32_t* in = k ? in1 : in0;
        int32_t* out = k ? out1 : out0;
        int carry = use_carry ? M->cur[2 + k] : 0;
        for (int base = begin; base < end; base += 4096) {
            const int i = base + 4 * tid;
            int v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (i + j < end) ? in[i + j] : 0;
            const int mine = v[0] + v[1] + v[2] + v[3];
            int incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d, 64);
                incl += lane >= d ? o : 0;
            }
            __syncthreads();                       // (s_w / s_tot of the previous tile are read)
            if (lane == 63) s_w[wv] = incl;
            __syncthreads();
            if (tid < 64) {
                const int w = tid < 16 ? s_w[tid] : 0;
                int wi = w;
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    const int o = __shfl_up(wi, d, 64);
                    wi += tid >= d ? o : 0;
                }
                if (tid < 16) s_w[tid] = wi - w;   // exclusive prefix of the wavefront sums
                if (tid == 15) s_tot = wi;
            }
            __syncthreads();
            int ex = carry + s_w[wv] + incl - mine;
#pragma unroll
            for (int j = 0; j < 4; ++j) { if (i + j < end) out[i + j] = ex; ex += v[j]; }
            carry += s_tot;
        }
        if (tid == 0) { M->cur[4 + k] = carry; out[end] = carry; }
    }
}

// ---- generic 3-kernel exclusive scan over a device-side range ------------------------------------
// K arrays scanned together.  Range [cur[0], cur[1]) ; tile t covers begin + t*KGW_TILE.
template <int K>
__global__ void __launch_bounds__(KGW_BLK) k_scan_tiles(const int32_t* in0, const int32_t* in1,
                                                        const KgwBatchMeta* M, int32_t* tile_sums) {
    __shared__ int sm[KGW_BLK];
    const int begin = M->cur[0], end = M->cur[1];
    const int ntiles = (end - begin + KGW_TILE - 1) / KGW_TILE;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int s0 = 0, s1 = 0;
        for (int k = 0; k < KGW_TILE / KGW_BLK; ++k) {
            int i = begin + t * KGW_TILE + k * KGW_BLK + threadIdx.x;
            if (i < end) { s0 += in0[i]; if (K == 2) s1 += in1[i]; }
        }
        int tot;
        kgw_block_exscan(s0, sm, &tot);
        if (threadIdx.x == 0) tile_sums[K * t] = tot;
        if (K == 2) {
            kgw_block_exscan(s1, sm, &tot);
            if (threadIdx.x == 0) tile_sums[K * t + 1] = tot;
        }
    }
}

// single block: exclusive scan of the tile sums (+ carry-in cur[2], cur[3]); total at [ntiles].
template <int K>
__global__ void __launch_bounds__(KGW_BLK) k_scan_top(int32_t* tile_sums, KgwBatchMeta* M, int use_carry) {
    __shared__ int sm[KGW_BLK];
    const int begin = M->cur[0], end = M->cur[1];
    const int ntiles = (end - begin + KGW_TILE - 1) / KGW_TILE;
    for (int k = 0; k < K; ++k) {
        int carry = use_carry ? M->cur[2 + k] : 0;
        for (int base = 0; base <= ntiles; base += KGW_BLK) {
            int i = base + threadIdx.x;
            int v = (i < ntiles) ? tile_sums[K * i + k] : 0;
            int tot;
            int ex = kgw_block_exscan(v, sm, &tot);
            if (i <= ntiles) tile_sums[K * i + k] = carry + ex;
            carry += tot;
        }
        if (threadIdx.x == 0) M->cur[4 + k] = carry;   // grand total (incl. carry-in)
    }
}

template <int K>
__global__ void __launch_bounds__(KGW_BLK) k_scan_apply(const int32_t* in0, const int32_t* in1,
                                                        int32_t* out0, int32_t* out1,
                                                        const KgwBatchMeta* M, const int32_t* tile_sums) {
    __shared__ int sm[KGW_BLK];
    const int begin = M->cur[0], end = M->cur[1];
    const int ntiles = (end - begin + KGW_TILE - 1) / KGW_TILE;
    constexpr int PER = KGW_TILE / KGW_BLK;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int i0 = begin + t * KGW_TILE + threadIdx.x * PER;   // PER consecutive elements per thread
        for (int k = 0; k < K; ++k) {
            const int32_t* in = k ? in1 : in0;
            int32_t* out = k ? out1 : out0;
            int v[PER], s = 0;
            for (int j = 0; j < PER; ++j) { v[j] = (i0 + j < end) ? in[i0 + j] : 0; s += v[j]; }
            int tot;
            int ex = kgw_block_exscan(s, sm, &tot) + tile_sums[K * t + k];
            for (int j = 0; j < PER; ++j) { if (i0 + j < end) out[i0 + j] = ex; ex += v[j]; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {          // end sentinel
        out0[end] = M->cur[4];
        if (K == 2) out1[end] = M->cur[5];
    }
}

// (the bookkeeping of the finished scans -- edge_end / chunk_end of the hop, capacity checks -- rides in this launch: every
//  block evaluates the checks for itself, block 0 publishes)
__global__ void __launch_bounds__(KGW_BLK) k_fill_chunks(SampArgs A, int h) {
    __shared__ int s_off[KGW_MAX_RELS + 1], s_err;
    const KgwGraph& G = A.G;
    KgwBatchMeta* M = A.B.meta;
    if (threadIdx.x == 0) {
        const int e4 = M->cur[4], e5 = M->cur[5];
        int err = M->error;
        if ((int64_t)e4 > A.B.edge_cap) err |= 2;
        if ((int64_t)e5 > A.B.chunk_cap) err |= 4;
        s_err = err;
        if (blockIdx.x == 0) {
            M->edge_end[h] = e4;
            M->chunk_end[h] = e5;
            if (err != M->error) atomicOr(&M->error, err);
        }
        for (int r = 0; r <= G.n_rels; ++r) s_off[r] = M->seg_off[h][r];
    }
    __syncthreads();
    if (s_err) return;
    const int begin = M->cur[0], end = M->cur[1];
    for (int sg = begin + blockIdx.x * KGW_BLK + threadIdx.x; sg < end; sg += gridDim.x * KGW_BLK) {
        const int nch = A.B.seg_nch[sg];
        if (nch == 0) continue;
        const int r = find_rel(s_off, G.n_rels, sg);
        const int d = G.rel_dst[r];
        const int li = M->node_off[d][h] + (sg - s_off[r]);
        const int g = A.B.n_id[G.node_base[d] + li];
        const int32_t* rp = G.g_rowptr + G.rowptr_off[r];
        int64_t gpos = G.col_off[r] + rp[g];
        int e = A.B.seg_ptr[sg];
        const int e_end = e + A.B.seg_deg[sg];
        const int c0 = A.B.seg_chptr[sg];
        for (int c = 0; c < nch; ++c) {
            KgwChunk ck;
            ck.e0 = e;
            ck.e1 = min(e + KGW_CHUNK, e_end);
            ck.row = li; ck.rel = r; ck.first = c0; ck.nch = nch;
            ck.gpos_lo = (int32_t)(gpos & 0xFFFFFFFFll);
            ck.gpos_hi = (int32_t)(gpos >> 32);
            A.B.chunks[c0 + c] = ck;
            e += KGW_CHUNK; gpos += KGW_CHUNK;
        }
        if (nch > 1) {
            int idx = atomicAdd(&M->multi_cnt[h], 1);
            if ((int64_t)idx < A.B.multi_cap) {
                int32_t* mm = A.B.multi + ((int64_t)h * A.B.multi_cap + idx) * 4;
                mm[0] = c0; mm[1] = nch; mm[2] = li; mm[3] = r;
            } else {
                atomicOr(&M->error, 8);
            }
        }
    }
}

__device__ __forceinline__ int64_t chunk_gpos(const KgwChunk& c) {
    return ((int64_t)c.gpos_hi << 32) | (uint32_t)c.gpos_lo;
}

// one wavefront per FOUR consecutive chunks of dst hop h: flag every not-yet-sampled source node.  (Round 4: a chunk is a
// chain of three dependent round trips -- chunk record, its column ids, their table entries -- with two loads per lane in
// flight; beside a training step the launch has 1 024 wavefronts for ~8 k chunks, so the chain, not bandwidth, set its
// 56 us.  Four chunks per wavefront-iteration put 8 column loads, then 8 table reads, in flight per lane.)
constexpr int KGW_WALK = 4;
__global__ void __launch_bounds__(KGW_BLK) k_mark(SampArgs A, int h) {
    const KgwGraph& G = A.G;
    const KgwBatchMeta* M = A.B.meta;
    if (M->error) return;
    const int cb = (h == 0) ? 0 : M->chunk_end[h - 1], ce = M->chunk_end[h];
    const int lane = kgw_lane();
    for (int c = cb + (blockIdx.x * 4 + (threadIdx.x >> 6)) * KGW_WALK; c < ce; c += gridDim.x * 4 * KGW_WALK) {
        KgwChunk ck[KGW_WALK];
#pragma unroll
        for (int q = 0; q < KGW_WALK; ++q) ck[q] = A.B.chunks[min(c + q, ce - 1)];
        int g[KGW_WALK][KGW_CHUNK / 64];
#pragma unroll
        for (int q = 0; q < KGW_WALK; ++q) {
            const int32_t* col = G.g_col + chunk_gpos(ck[q]);
            const int n = (c + q < ce) ? ck[q].e1 - ck[q].e0 : 0;
#pragma unroll
            for (int u = 0; u < KGW_CHUNK / 64; ++u) g[q][u] = (lane + 64 * u < n) ? col[lane + 64 * u] : -1;
        }
        int cur[KGW_WALK][KGW_CHUNK / 64];
#pragma unroll
        for (int q = 0; q < KGW_WALK; ++q) {
            const int32_t* g2l = A.B.g2l + G.node_base[G.rel_src[ck[q].rel]];
#pragma unroll
            for (int u = 0; u < KGW_CHUNK / 64; ++u) cur[q][u] = g[q][u] >= 0 ? g2l[g[q][u]] : 0;
        }
#pragma unroll
        for (int q = 0; q < KGW_WALK; ++q) {
            int32_t* g2l = A.B.g2l + G.node_base[G.rel_src[ck[q].rel]];
#pragma unroll
            for (int u = 0; u < KGW_CHUNK / 64; ++u)
                if (g[q][u] >= 0 && cur[q][u] == -1) g2l[g[q][u]] = KGW_PENDING;   // benign race: every writer stores the same value
        }
    }
}

// compaction of the PENDING flags over the padded concatenated node space (host-known size)
__global__ void __launch_bounds__(KGW_BLK) k_count_pending(SampArgs A, int32_t* tile_sums, int ntiles) {
    __shared__ int sm[KGW_BLK];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int s = 0;
        for (int k = 0; k < KGW_TILE / KGW_BLK; ++k)
            s += (A.B.g2l[t * KGW_TILE + k * KGW_BLK + threadIdx.x] == KGW_PENDING);
        int tot;
        kgw_block_exscan(s, sm, &tot);
        if (threadIdx.x == 0) tile_sums[t] = tot;
    }
}

__global__ void __launch_bounds__(KGW_BLK) k_scan_top_fixed(int32_t* tile_sums, int ntiles) {
    __shared__ int sm[KGW_BLK];
    int carry = 0;
    for (int base = 0; base <= ntiles; base += KGW_BLK) {
        int i = base + threadIdx.x;
        int v = (i < ntiles) ? tile_sums[i] : 0;
        int tot;
        int ex = kgw_block_exscan(v, sm, &tot);
        if (i <= ntiles) tile_sums[i] = carry + ex;
        carry += tot;
    }
}

__global__ void __launch_bounds__(KGW_BLK) k_assign(SampArgs A, const int32_t* tile_pref, int ntiles, int h) {
    __shared__ int sm[KGW_BLK];
    const KgwGraph& G = A.G;
    const KgwBatchMeta* M = A.B.meta;
    constexpr int PER = KGW_TILE / KGW_BLK;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (tile_pref[t + 1] == tile_pref[t]) continue;          // nothing pending in this tile
        const int i0 = t * KGW_TILE + threadIdx.x * PER;
        const int ty = type_of_node_slot(G, t * KGW_TILE);      // regions are tile aligned
        const int base = G.node_base[ty];
        const int off = M->node_off[ty][h + 1] + tile_pref[t] - tile_pref[base / KGW_TILE];
        int f[PER], s = 0;
        for (int j = 0; j < PER; ++j) { f[j] = (A.B.g2l[i0 + j] == KGW_PENDING); s += f[j]; }
        int tot;
        int ex = kgw_block_exscan(s, sm, &tot) + off;
        for (int j = 0; j < PER; ++j) {
            if (f[j]) { A.B.g2l[i0 + j] = ex; A.B.n_id[base + ex] = i0 + j - base; ++ex; }
        }
    }
}

// (block 0 also closes the hop: node counts of hop h + 1 from the compaction's tile prefix -- k_hop_end's one-thread launch)
__global__ void __launch_bounds__(KGW_BLK) k_relabel(SampArgs A, const int32_t* tile_pref, int h) {
    const KgwGraph& G = A.G;
    KgwBatchMeta* M = A.B.meta;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int t = 0; t < G.n_types; ++t) {
            int c = tile_pref[G.node_base[t + 1] / KGW_TILE] - tile_pref[G.node_base[t] / KGW_TILE];
            M->hop_cnt[t][h + 1] = c;
            M->node_off[t][h + 2] = M->node_off[t][h + 1] + c;
        }
    }
    if (M->error) return;
    const int cb = (h == 0) ? 0 : M->chunk_end[h - 1], ce = M->chunk_end[h];
    const int lane = kgw_lane();
    // (four chunks per wavefront-iteration, like k_mark: 8 column loads, then 8 table reads in flight per lane)
    int32_t* cE = A.B.t_tmp + (A.B.edge_cap + 1);
    for (int c = cb + (blockIdx.x * 4 + (threadIdx.x >> 6)) * KGW_WALK; c < ce; c += gridDim.x * 4 * KGW_WALK) {
        KgwChunk ck[KGW_WALK];
#pragma unroll
        for (int q = 0; q < KGW_WALK; ++q) ck[q] = A.B.chunks[min(c + q, ce - 1)];
        int g[KGW_WALK][KGW_CHUNK / 64];
#pragma unroll
        for (int q = 0; q < KGW_WALK; ++q) {
            const int32_t* col = G.g_col + chunk_gpos(ck[q]);
            const int n = (c + q < ce) ? ck[q].e1 - ck[q].e0 : 0;
#pragma unroll
            for (int u = 0; u < KGW_CHUNK / 64; ++u) g[q][u] = (lane + 64 * u < n) ? col[lane + 64 * u] : -1;
        }
        int loc[KGW_WALK][KGW_CHUNK / 64];
#pragma unroll
        for (int q = 0; q < KGW_WALK; ++q) {
            const int32_t* g2l = A.B.g2l + G.node_base[G.rel_src[ck[q].rel]];
#pragma unroll
            for (int u = 0; u < KGW_CHUNK / 64; ++u) loc[q][u] = g[q][u] >= 0 ? g2l[g[q][u]] : 0;
        }
#pragma unroll
        for (int q = 0; q < KGW_WALK; ++q)
#pragma unroll
            for (int u = 0; u < KGW_CHUNK / 64; ++u)
                if (g[q][u] >= 0) {
                    A.B.col_local[ck[q].e0 + lane + 64 * u] = loc[q][u];
                    cE[ck[q].e0 + lane + 64 * u] = c + q;          // chunk of the edge: what k_ts_keys / k_t_end look up
                }
    }
}

// ---- per-layer layout tables -------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_layer_tables(SampArgs A) {
    // one lane per (layer, node type); prefix sums over the <= KGW_MAX_TYPES types of a layer through LDS
    __shared__ int s_lr[KGW_MAX_LAYERS][KGW_MAX_TYPES], s_ls[KGW_MAX_LAYERS][KGW_MAX_TYPES];
    const KgwGraph& G = A.G;
    KgwBatchMeta* M = A.B.meta;
    const int L = G.n_layers, NT = G.n_types;
    const int l = threadIdx.x / KGW_MAX_TYPES + 1, t = threadIdx.x % KGW_MAX_TYPES;
    const bool on = (l <= L) && (t < NT) && threadIdx.x < KGW_MAX_LAYERS * KGW_MAX_TYPES;
    int hd = 0, nr = 0, ns = 0, lr = 0, ls = 0, err = 0;
    if (on) {
        hd = min(L - l, G.n_hops - 1);             // destination rows: hops <= hd
        bool dst_live = false, src_live = false;
        for (int r = 0; r < G.n_rels; ++r) {
            if (!G.rel_live[l - 1][r]) continue;
            dst_live |= (G.rel_dst[r] == t);
            src_live |= (G.rel_src[r] == t);
        }
        // a destination type always has a block in the layer input too (its rows are the destination-side
        // operand of the attention logits), even when no live relation leaves it
        src_live |= dst_live;
        nr = dst_live ? M->node_off[t][hd + 1] : 0;
        ns = src_live ? M->node_off[t][hd + 2] : 0;
        // row-block sizes of the layout: the batch's own counts, or fixed capacities (graph capture)
        lr = nr; ls = ns;
        if (G.static_layout) {
            lr = dst_live ? G.cap_rows[l - 1][t] : 0;
            ls = src_live ? G.cap_src[l - 1][t] : 0;
            if (nr > lr || ns > ls) err |= 32;
        }
        s_lr[l - 1][t] = lr;
        s_ls[l - 1][t] = ls;
    }
    __syncthreads();
    if (on) {
        int zb = 0, sb = 0, tb = 0;
        for (int q = 0; q < t; ++q) {
            zb += s_lr[l - 1][q] * G.R_dst[q];
            sb += s_ls[l - 1][q];
            tb += s_ls[l - 1][q] * G.R_src[q];
        }
        M->n_rows[l - 1][t] = nr;
        M->lay_rows[l - 1][t] = lr;
        M->z_base[l - 1][t] = zb;
        M->n_src[l - 1][t] = ns;
        M->lay_src[l - 1][t] = ls;
        M->src_base[l - 1][t] = sb;
        M->t_base[l - 1][t] = tb;
        if (t == NT - 1) {
            zb += lr * G.R_dst[t]; sb += ls; tb += ls * G.R_src[t];
            M->z_base[l - 1][NT] = zb;
            M->src_base[l - 1][NT] = sb;
            M->t_base[l - 1][NT] = tb;
            M->n_chunks[l - 1] = M->chunk_end[hd];
            M->n_edges[l - 1] = M->edge_end[hd];
            if ((int64_t)tb > A.B.trow_cap) err |= 16;
        }
        if (err) atomicOr(&M->error, err);
    }
}

// ---- src-major (transposed) structures: up to TWO layers per launch ------------------------------------------------------
// (layers l0 and l0 + 1: every kernel below loops over the pair, so the default 2-layer model builds both structures with
//  one set of launches.)
//
// The structure is the layer's edge list STABLY SORTED by src-major row (row = (source node, relation slot), t_base + col * R_src
// + slot): inside a row the entries then stand in ascending edge order, the summation order of the backward pass, the same run
// to run.  Round 3: a two-digit radix sort without global atomics instead of atomic histogram + atomic cursor fill + in-row rank
// (those were 0.25 of the sampler's 0.43 ms):
//   k_ts_keys      lane per edge: chunk of the edge (the chunk list is walked: one binary search per wavefront, then the end
//                  offsets of 64 chunks in registers and a 6-step search through shuffles), row key (INVALID for relations the
//                  layer does not compute), the HIGH digit (key >> sh; one bucket = 2^sh consecutive rows) counted per block in
//                  LDS.  Blocks own CONTIGUOUS edge ranges;
//   k_ts_scan_rows / k_ts_scan_tot   starts of the (digit, block) cells: prefix over the blocks inside a digit, then over the
//                  digits' totals;
//   k_ts_scatter   stable scatter by high digit: a block's four wavefronts own contiguous quarters of its range, count them,
//                  start behind each other, and walk their quarter in order 64 edges at a time -- the rank of an edge among the
//                  lanes with the same digit comes from ballots (no atomics, no dependence on arrival order);
//   k_ts_rows      one block per bucket, done the same way on the low digit: counts of the bucket's rows in LDS, exclusive scan
//                  = the row pointers (written for every row, empty ones included), then the entries' edge ids placed in order;
//   k_t_end        Z row and relation of every entry looked up from its chunk (independent gathers), octet flags.
// Measured (512-seed batch of the benchmark graph, sampler alone, 256-block launches): 433 -> 355 us per batch; beside the
// training step it now costs the step 40 - 50 us instead of 70 - 80 (no global atomics: 1.18 -> 1.155 ms per step).
// KgwBatchBuf.t_tmp per layer: key[edge], (first layer's slot: chunk[edge] of ALL layers, written by k_relabel), then the (key, edge)
// PAIRS sorted by bucket (round 6: one scattered 8-byte store per entry instead of two 4-byte ones) -- 4 x (edge_cap + 1) ints; the counts live in
// KgwBatchBuf.scan_tmp ([layer][digit][block]); KgwBatchMeta.cur[4 + k] = entries of layer l0 + k.
// (blocks of k_ts_keys / k_ts_scatter = contiguous edge ranges: 256 beside a training step, 512 when the call has the GPU)
constexpr int TS_INVALID = 0x7fffffff;
constexpr int TS_MAX_NB = 4000;                // buckets aimed at: 4 x (nb + 1) counters fit 64 KB of LDS in k_ts_scatter
constexpr int TS_HARD_MAX_NB = 9000;           // buckets at most (graphs above 65 M src-major rows: 2^14-row buckets, 144 KB of LDS)

__device__ __forceinline__ void ts_block_range(int n, int b, int nblk, int& beg, int& end) {
    int per = (n + nblk - 1) / nblk;
    per = (per + 255) & ~255;                  // whole groups of 64 for each of the four wavefronts
    beg = min(n, b * per);
    end = min(n, beg + per);
}

// rank of this lane among the (valid) lanes whose ``d`` equals its own, and their number: ``nbits`` ballots
__device__ __forceinline__ void ts_match(int d, bool valid, int nbits, int lane, int& rank, int& cnt) {
    unsigned long long m = __ballot(valid);
    for (int b = 0; b < nbits; ++b) {
        const bool bit = (d >> b) & 1;
        const unsigned long long bal = __ballot(valid && bit);
        m &= bit ? bal : ~bal;
    }
    rank = __popcll(m & ((1ull << lane) - 1ull));
    cnt = __popcll(m);
}

__global__ void __launch_bounds__(KGW_BLK) k_ts_keys(SampArgs A, int l0, int nl, int sh, int nb) {
    extern __shared__ int ts_lds[];            // [nb + 1] digit counts of this block
    const KgwGraph& G = A.G;
    KgwBatchMeta* M = A.B.meta;
    if (M->error) return;
    const int64_t E1 = A.B.edge_cap + 1;
    const int lane = kgw_lane(), wv = threadIdx.x >> 6;
    for (int k = 0; k < nl; ++k) {
        const int l = l0 + k;
        const int n = M->n_edges[l - 1], nc = M->n_chunks[l - 1];
        int32_t* keyE = A.B.t_tmp + (int64_t)k * 4 * E1;
        const int32_t* cE = A.B.t_tmp + E1;              // chunk of every edge (k_relabel; the same for every layer)
        int32_t* H = A.B.scan_tmp + (int64_t)k * (nb + 1) * gridDim.x;
        for (int d = threadIdx.x; d <= nb; d += KGW_BLK) ts_lds[d] = 0;
        __syncthreads();
        int beg, end;
        ts_block_range(n, blockIdx.x, gridDim.x, beg, end);
        // the chunk of every edge was written by k_relabel (round 4; until then the chunk list was WALKED here -- one binary
        // search per wavefront, then the end offsets of 64 chunks and a 6-step search through shuffles per group of 64 edges, each
        // group starting where the last one ended: a serial chain of ~3 round trips x 16 groups per wavefront, 100 us of the
        // sampler's 565).  Now every edge is independent: four groups of 64 per iteration, 12 loads in flight per lane.
        for (int g = beg + threadIdx.x; g < end; g += 4 * KGW_BLK) {
            int c[4], cl[4], r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = g + u * KGW_BLK;
                c[u] = e < end ? cE[e] : 0;
                cl[u] = e < end ? A.B.col_local[e] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = A.B.chunks[c[u]].rel;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = g + u * KGW_BLK;
                if (e >= end) continue;
                int key = TS_INVALID;
                if (G.rel_live[l - 1][r[u]]) {
                    const int sT = G.rel_src[r[u]];
                    key = M->t_base[l - 1][sT] + cl[u] * G.R_src[sT] + G.rel_slot_src[r[u]];
                }
                keyE[e] = key;
                atomicAdd(&ts_lds[key == TS_INVALID ? nb : (key >> sh)], 1);
            }
        }
        __syncthreads();
        for (int d = threadIdx.x; d <= nb; d += KGW_BLK) H[(int64_t)d * gridDim.x + blockIdx.x] = ts_lds[d];
        __syncthreads();
    }
}

// Offsets of the (digit, block) cells: inside a digit the exclusive prefix over its blocks (one wavefront per digit row, in
// place) and the digit's total; then the exclusive scan of the totals (one block per layer).  Start of cell (d, b) =
// dbase[d] + H[d][b]; dbase[nb] = entries of the layer (digit nb collects the edges of relations the layer does not compute).
template <int V>      // V = nblk / 64 consecutive cells per lane
__global__ void __launch_bounds__(KGW_BLK) k_ts_scan_rows(int32_t* scan_tmp, int nl, int nb, const KgwBatchMeta* __restrict__ M) {
    if (M->error) return;
    constexpr int nblk = 64 * V;
    const int lane = kgw_lane();
    int32_t* tot0 = scan_tmp + (int64_t)nl * (nb + 1) * nblk;
    const int nrow = nl * (nb + 1);
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < nrow; row += gridDim.x * 4) {
        int32_t* H = scan_tmp + (int64_t)row * nblk + V * lane;
        int v[V];
#pragma unroll
        for (int j = 0; j < V; j += 2) { const int2 t = *(const int2*)(H + j); v[j] = t.x; v[j + 1] = t.y; }
        int mine = 0;
#pragma unroll
        for (int j = 0; j < V; ++j) mine += v[j];
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            incl += lane >= d ? o : 0;
        }
        int ex = incl - mine;
#pragma unroll
        for (int j = 0; j < V; j += 2) {
            const int a = ex, b = ex + v[j];
            *(int2*)(H + j) = make_int2(a, b);
            ex = b + v[j + 1];
        }
        if (lane == 63) tot0[(row / (nb + 1)) * (nb + 2) + row % (nb + 1)] = incl;
    }
}

__global__ void __launch_bounds__(1024) k_ts_scan_tot(int32_t* scan_tmp, int nl, int nb, int nblk, const KgwBatchMeta* __restrict__ M) {
    __shared__ int s_w[16];
    if (M->error) return;
    int32_t* T = scan_tmp + (int64_t)nl * (nb + 1) * nblk + (int64_t)blockIdx.x * (nb + 2);      // [nb + 1] totals -> [nb + 2] starts
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = nb + 1;
    int carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = i < n ? T[i] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            incl += lane >= d ? o : 0;
        }
        __syncthreads();
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int wbase = 0, total = 0;
        for (int w = 0; w < 16; ++w) { wbase += w < wv ? s_w[w] : 0; total += s_w[w]; }
        if (i < n) T[i] = carry + wbase + incl - v;
        carry += total;
    }
    if (tid == 0) T[n] = carry;
}

// (ordered walks below: a global store inside the loop would put its acknowledgement on the critical path of the next
//  group's loads -- vector memory operations return in issue order -- so a tile's keys are loaded up front, its positions kept
//  in registers, and the stores of the whole tile issued together)
constexpr int TS_TILE = 512;                   // entries of an ordered tile (8 groups of 64)

__global__ void __launch_bounds__(KGW_BLK) k_ts_scatter(SampArgs A, int l0, int nl, int sh, int nb, int nbits) {
    extern __shared__ int ts_lds[];            // [4][nb + 1]: per wavefront, first its counts, then its running write positions
    const KgwBatchMeta* M = A.B.meta;
    if (M->error) return;
    const int64_t E1 = A.B.edge_cap + 1;
    const int lane = kgw_lane(), wv = threadIdx.x >> 6;
    int* mine = ts_lds + wv * (nb + 1);
    for (int k = 0; k < nl; ++k) {
        const int l = l0 + k;
        const int n = M->n_edges[l - 1];
        const int32_t* keyE = A.B.t_tmp + (int64_t)k * 4 * E1;
        int2* pairS = (int2*)(A.B.t_tmp + (int64_t)k * 4 * E1 + 2 * E1);      // (key, edge) of every entry, sorted by bucket
        const int32_t* H = A.B.scan_tmp + (int64_t)k * (nb + 1) * gridDim.x;
        const int32_t* dbase = A.B.scan_tmp + (int64_t)nl * (nb + 1) * gridDim.x + (int64_t)k * (nb + 2);
        for (int d = threadIdx.x; d < 4 * (nb + 1); d += KGW_BLK) ts_lds[d] = 0;
        __syncthreads();
        int beg, end;
        ts_block_range(n, blockIdx.x, gridDim.x, beg, end);
        const int q = (((end - beg) + 3) / 4 + 63) & ~63;          // a wavefront's quarter: whole groups of 64
        const int wb = min(end, beg + wv * q), we = min(end, wb + q);
        for (int g = wb; g < we; g += TS_TILE) {
            int kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kk[u] = (g + 64 * u + lane < we) ? keyE[g + 64 * u + lane] : -1;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (kk[u] >= 0) atomicAdd(&mine[kk[u] == TS_INVALID ? nb : (kk[u] >> sh)], 1);
        }
        __syncthreads();
        for (int d = threadIdx.x; d <= nb; d += KGW_BLK) {
            int run = dbase[d] + H[(int64_t)d * gridDim.x + blockIdx.x];
            for (int w = 0; w < 4; ++w) { const int t = ts_lds[w * (nb + 1) + d]; ts_lds[w * (nb + 1) + d] = run; run += t; }
        }
        __syncthreads();
        for (int g = wb; g < we; g += TS_TILE) {
            int kk[8], pp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kk[u] = (g + 64 * u + lane < we) ? keyE[g + 64 * u + lane] : -1;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool valid = kk[u] >= 0;
                const int d = (!valid || kk[u] == TS_INVALID) ? nb : (kk[u] >> sh);
                int rank, cnt;
                ts_match(d, valid, nbits, lane, rank, cnt);
                int pos = -1;
                if (valid) pos = mine[d] + rank;
                __builtin_amdgcn_wave_barrier();                      // (every lane has read its digit's position)
                if (valid && rank == cnt - 1) mine[d] += cnt;
                __builtin_amdgcn_wave_barrier();
                pp[u] = (valid && d < nb) ? pos : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (pp[u] >= 0) pairS[pp[u]] = make_int2(kk[u], g + 64 * u + lane);     // (ONE scattered 8-byte store per entry)
        }
        __syncthreads();
    }
}

// W wavefronts per bucket: each owns a contiguous W-th of the bucket's (already edge-ordered) entries, counts its rows, starts
// behind the wavefronts before it, and places its entries in order -- the longest bucket (a few thousand entries around the most
// connected genes) sets the kernel's time, W = 4 divides it.
template <int W>
__global__ void __launch_bounds__(64 * W) k_ts_rows(SampArgs A, int l0, int nl, int sh, int nb, int nblk) {
    extern __shared__ int ts_lds[];            // [W][2^sh]: per wavefront the counts of the bucket's rows, then its running positions
    __shared__ int s_w[W];
    const KgwGraph& G = A.G;
    KgwBatchMeta* M = A.B.meta;
    if (M->error) return;
    const int64_t E1 = A.B.edge_cap + 1;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nrow = 1 << sh;
    int* mine = ts_lds + wv * nrow;
    for (int k = 0; k < nl; ++k) {
        const int l = l0 + k;
        const int TR = M->t_base[l - 1][G.n_types];
        const int2* pairS = (const int2*)(A.B.t_tmp + (int64_t)k * 4 * E1 + 2 * E1);
        const int32_t* dbase = A.B.scan_tmp + (int64_t)nl * (nb + 1) * nblk + (int64_t)k * (nb + 2);
        int32_t* tp = A.B.t_ptr[l - 1];
        int32_t* te = A.B.t_edge[l - 1];
        if (blockIdx.x == 0 && tid == 0) M->cur[4 + k] = dbase[nb];                             // entries of the layer
        for (int b = blockIdx.x; b < nb; b += gridDim.x) {
            const int row0 = b << sh;
            if (row0 > TR) break;
            const int s0 = dbase[b], s1 = dbase[b + 1];
            for (int i = tid; i < W * nrow; i += 64 * W) ts_lds[i] = 0;
            __syncthreads();
            const int q = (((s1 - s0) + W - 1) / W + 63) & ~63;      // a wavefront's share: whole groups of 64
            const int wb = min(s1, s0 + wv * q), we = min(s1, wb + q);
            for (int p = wb; p < we; p += TS_TILE) {
                int kk[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) kk[u] = (p + 64 * u + lane < we) ? pairS[p + 64 * u + lane].x : -1;
#pragma unroll
                for (int u = 0; u < 8; ++u) if (kk[u] >= 0) atomicAdd(&mine[kk[u] - row0], 1);
            }
            __syncthreads();
            // row pointers = exclusive scan of the rows' totals (every row of the bucket, empty ones included); each wavefront's
            // counter becomes its first write position in the row
            int carry = s0;
            for (int i0 = 0; i0 < nrow && row0 + i0 <= TR; i0 += 64 * W) {
                const int i = i0 + tid;
                int c[W], tot = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) { c[w] = ts_lds[w * nrow + i]; tot += c[w]; }
                int incl = tot;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(incl, d, 64);
                    incl += lane >= d ? o : 0;
                }
                if (lane == 63) s_w[wv] = incl;
                __syncthreads();
                int wbase = 0, total = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) { wbase += w < wv ? s_w[w] : 0; total += s_w[w]; }
                int run = carry + wbase + incl - tot;
                if (row0 + i <= TR) tp[row0 + i] = run;
#pragma unroll
                for (int w = 0; w < W; ++w) { ts_lds[w * nrow + i] = run; run += c[w]; }
                carry += total;
                __syncthreads();
            }
            // the wavefront's entries in order, a tile of 8 groups at a time: keys and edge ids loaded up front, positions kept in
            // registers, the tile's stores issued together
            for (int g = wb; g < we; g += TS_TILE) {
                int kk[8], ee[8], pp[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool in = g + 64 * u + lane < we;
                    const int2 pr = in ? pairS[g + 64 * u + lane] : make_int2(-1, 0);
                    kk[u] = pr.x;
                    ee[u] = pr.y;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool valid = kk[u] >= 0;
                    const int low = valid ? kk[u] - row0 : 0;
                    int rank, cnt;
                    ts_match(low, valid, sh, lane, rank, cnt);
                    pp[u] = -1;
                    if (valid) pp[u] = mine[low] + rank;
                    __builtin_amdgcn_wave_barrier();
                    if (valid && rank == cnt - 1) mine[low] += cnt;
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (pp[u] >= 0) te[pp[u]] = ee[u];
            }
            __syncthreads();
        }
    }
}

// KgwBatchBuf.t_cnt (the histogram of the round-2 build; nothing else uses it now) receives one flag
// per group of 8 consecutive source rows of the layer input ("octet", row index / 8) for the backward's src-major pass
// (kgw_gat_aggregate_bwd_src, KgwLayerArgs.oct_flags): 1 = eight real rows of ONE node type of KgwGraph.short_types, none
// of them a destination row of the layer, each with at most 8 entries over all its relation slots.
__global__ void __launch_bounds__(KGW_BLK) k_t_end(SampArgs A, int l0, int nl) {
    const KgwGraph& G = A.G;
    KgwBatchMeta* M = A.B.meta;
    const int NT = G.n_types;
    const int64_t E1 = A.B.edge_cap + 1;
    for (int k = 0; k < nl; ++k) {
        const int l = l0 + k;
        if (blockIdx.x == 0 && threadIdx.x == 0) M->t_entries[l - 1] = M->cur[4 + k];
        if (!M->error) {
            // Z row and relation of every entry, from the chunk of its edge (independent gathers: this is the parallel half of
            // the placement, k_ts_rows does the ordered half)
            const int32_t* cE = A.B.t_tmp + E1;           // chunk of every edge (k_relabel)
            const int n_ent = M->cur[4 + k];
            const int nthr = gridDim.x * KGW_BLK;
            for (int j0 = blockIdx.x * KGW_BLK + threadIdx.x; j0 < n_ent; j0 += 4 * nthr) {       // (four independent chains in flight)
                int e[4], c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) e[u] = (j0 + u * nthr < n_ent) ? A.B.t_edge[l - 1][j0 + u * nthr] : -1;
#pragma unroll
                for (int u = 0; u < 4; ++u) c[u] = e[u] >= 0 ? cE[e[u]] : 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (e[u] < 0) continue;
                    const KgwChunk ck = A.B.chunks[c[u]];
                    const int r = ck.rel, dT = G.rel_dst[r];
                    A.B.t_zrow[l - 1][j0 + u * nthr] = M->z_base[l - 1][dT] + ck.row * G.R_dst[dT] + G.rel_slot_dst[r];
                    if (A.B.t_rel[l - 1]) A.B.t_rel[l - 1][j0 + u * nthr] = (uint8_t)r;
                }
            }
        }
        const int n_oct = M->error ? 0 : (M->src_base[l - 1][NT] + 7) >> 3;
        const int32_t* tp = A.B.t_ptr[l - 1];
        int32_t* out = A.B.t_cnt[l - 1];
        for (int o = blockIdx.x * KGW_BLK + threadIdx.x; o < n_oct; o += gridDim.x * KGW_BLK) {
            const int u0 = 8 * o;
            int ty = 0;
            while (ty + 1 < NT && u0 >= M->src_base[l - 1][ty + 1]) ++ty;
            const int j0 = u0 - M->src_base[l - 1][ty];
            int ok = ((G.short_types >> ty) & 1u) && j0 + 8 <= M->n_src[l - 1][ty] &&
                     !(G.R_dst[ty] > 0 && j0 < M->n_rows[l - 1][ty]);
            if (ok) {
                const int Rs = G.R_src[ty];
                const int tb = M->t_base[l - 1][ty] + j0 * Rs;
                int prev = tp[tb];
                for (int q = 1; q <= 8; ++q) {
                    const int cur = tp[tb + q * Rs];
                    if (cur - prev > 8) ok = 0;
                    prev = cur;
                }
            }
            out[o] = ok;
        }
    }
}

}  // namespace

// Parts of one sampling call (kgw_sample_batch_parts): hop h contributes part 2h (segments, chunks, and the flags on every
// not-yet-sampled source node: everything up to and including k_mark) and part 2h + 1 (compaction of the flags into local
// ids, relabelling of the hop's edges); part 2 * n_hops builds the layer tables and the src-major structures and exports
// the meta block.  Between part 2h and 2h + 1 the caller may merge the flags of node types that are REPLICATED across
// ranks (SNP-sharded multi-GPU mode: element-wise MIN over the ranks' g2l regions -- KGW_PENDING = -2 < -1 = unsampled,
// local ids >= 0 are equal on every rank), so that every rank expands the same replicated frontier at the next hop.
// buckets of 2^sh rows: as many as k_ts_scatter's LDS counters allow (small buckets = many wavefronts in k_ts_rows, short
// ordered walks), at most 2^14 rows each (one wavefront keeps a bucket's row counters in LDS); digits 0 .. nb (nb = "relation
// not computed by the layer")
static int ts_plan(int64_t trows, int* sh_, int* nb_, int* nbits_) {
    // Buckets of at least 2^11 rows (round 6; 2^8 until then).  Measured beside the training step of the benchmark (1.09 M src-major
    // rows, 1 024-block hop launches): 2^9-row buckets 1.034 ms per step, 2^10 1.023, 2^11 1.012, 2^12 1.009 - 1.016, 2^13 (one
    // wavefront per bucket) 1.134 -- although the sampler ALONE is fastest with the small ones (0.257 ms against 0.272 / 0.291):
    // 4x fewer buckets are 4x fewer cells of the [bucket][block] count table (one scattered 4-byte access each in k_ts_keys and
    // k_ts_scatter) and a quarter of k_ts_scatter's LDS, and what the sampler costs the step is what it takes from the step's
    // kernels, not its own length.  KGW_TS_MIN_SHIFT (tests): force a plan -- 8: the fine buckets; 12 / 14: the coarse
    // ones (64 KB-of-LDS launches) that otherwise need 8 M / 65 M src-major rows.
    static const int sh_env = getenv("KGW_TS_MIN_SHIFT") ? atoi(getenv("KGW_TS_MIN_SHIFT")) : 11;
    int sh = sh_env < 8 ? 8 : (sh_env > 14 ? 14 : sh_env);
    while (sh < 14 && (trows >> sh) + 1 > TS_MAX_NB) ++sh;
    const int64_t nb = (trows >> sh) + 1;
    if (nb > TS_HARD_MAX_NB) return 1;
    int nbits = 1;
    while ((1 << nbits) <= nb) ++nbits;
    *sh_ = sh; *nb_ = (int)nb; *nbits_ = nbits;
    return 0;
}

extern "C" int64_t kgw_sampler_scan_ints(int64_t seg_cap, int64_t node_slots, int64_t trow_cap) {
    int64_t m = seg_cap > node_slots ? seg_cap : node_slots;
    int64_t need = 2 * (m / KGW_TILE + 4);
    // the sort's [layer][bucket][block] counts + bucket starts: the bucket count depends on the rows a CALL can have (static
    // capacities may be far below trow_cap and then use finer buckets), so the bound is the largest plan there is
    int64_t nbmax = trow_cap / 256 + 2;                                  // (sh >= 8)
    if (nbmax > TS_MAX_NB) nbmax = TS_MAX_NB;
    if ((trow_cap >> 14) + 2 > nbmax) nbmax = (trow_cap >> 14) + 2;      // (beyond 65 M rows the 2^14-row buckets outnumber that)
    const int64_t t = 2 * (nbmax + 1) * 512 + 2 * (nbmax + 2);
    return t > need ? t : need;
}

extern "C" int kgw_sample_batch_parts(const KgwGraph* graph, const KgwBatchBuf* buf, const int64_t* seeds,
                                      int32_t n_seeds, int32_t seed_type, int32_t full_graph, int32_t part_begin,
                                      int32_t part_end, kgw_stream_t stream_) {
    if (!graph || !buf) return KGW_E_NULL;
    // grid of the sampler's grid-stride kernels: KgwBatchBuf.grid_blocks (a sampler replayed BESIDE a training step keeps
    // its launches small), else the whole-GPU default
    const int SG = buf->grid_blocks > 0 ? (buf->grid_blocks < KGW_GRID ? buf->grid_blocks : KGW_GRID) : KGW_GRID;
    if (!full_graph && (!seeds || n_seeds <= 0)) return KGW_E_NULL;
    if (graph->n_types < 1 || graph->n_types > KGW_MAX_TYPES || graph->n_rels < 1 ||
        graph->n_rels > KGW_MAX_RELS || graph->n_layers < 1 || graph->n_layers > KGW_MAX_LAYERS ||
        graph->n_hops < 1 || graph->n_hops > graph->n_layers)
        return KGW_E_RANGE;
    if (!full_graph && (seed_type < 0 || seed_type >= graph->n_types || n_seeds > graph->n_nodes[seed_type]))
        return KGW_E_RANGE;
    const int last_part = 2 * graph->n_hops;
    if (part_begin < 0 || part_end > last_part || part_begin > part_end) return KGW_E_RANGE;
    hipStream_t st = (hipStream_t)stream_;
    SampArgs A;
    A.G = *graph;
    A.B = *buf;
    const int total_slots = graph->node_base[graph->n_types];
    const int ntiles_nodes = total_slots / KGW_TILE;
    if ((int64_t)ntiles_nodes + 2 > buf->scan_cap) return KGW_E_RANGE;

    if (part_begin == 0) {
        { int rc = fill_i32(buf->g2l, -1, total_slots, st, SG); if (rc) return rc; }
        { int rc = fill_i32((int32_t*)buf->meta, 0, sizeof(KgwBatchMeta) / sizeof(int32_t), st); if (rc) return rc; }
        k_init<<<full_graph ? KGW_GRID : 64, KGW_BLK, 0, st>>>(A, seeds, n_seeds, seed_type, full_graph);
        KGW_LAUNCH_CHECK();
    }

    for (int h = 0; h < graph->n_hops; ++h) {
        if (2 * h >= part_begin && 2 * h <= part_end) {
            k_seg_deg<<<SG, KGW_BLK, 0, st>>>(A, h);                       // (+ the hop's segment offsets)
            if (!full_graph) {
                // a minibatch hop has a few 10 k segments: one block scans them (one launch instead of three)
                k_scan_block<2><<<1, 1024, 0, st>>>(buf->seg_deg, buf->seg_nch, buf->seg_ptr, buf->seg_chptr, buf->meta, 1);
            } else {
                k_scan_tiles<2><<<SG, KGW_BLK, 0, st>>>(buf->seg_deg, buf->seg_nch, buf->meta, buf->scan_tmp);
                k_scan_top<2><<<1, KGW_BLK, 0, st>>>(buf->scan_tmp, buf->meta, 1);
                k_scan_apply<2><<<SG, KGW_BLK, 0, st>>>(buf->seg_deg, buf->seg_nch, buf->seg_ptr,
                                                              buf->seg_chptr, buf->meta, buf->scan_tmp);
            }
            k_fill_chunks<<<SG, KGW_BLK, 0, st>>>(A, h);                   // (+ edge / chunk totals of the hop, capacity checks)
            KGW_LAUNCH_CHECK();
            if (!full_graph) {
                k_mark<<<SG, KGW_BLK, 0, st>>>(A, h);
                KGW_LAUNCH_CHECK();
            }
        }
        if (2 * h + 1 >= part_begin && 2 * h + 1 <= part_end) {
            if (!full_graph) {
                k_count_pending<<<SG, KGW_BLK, 0, st>>>(A, buf->scan_tmp, ntiles_nodes);
                k_scan_top_fixed<<<1, KGW_BLK, 0, st>>>(buf->scan_tmp, ntiles_nodes);
                k_assign<<<SG, KGW_BLK, 0, st>>>(A, buf->scan_tmp, ntiles_nodes, h);
            } else {
                // every node is already a seed: hop h+1 adds nothing
                { int rc = fill_i32(buf->scan_tmp, 0, ntiles_nodes + 2, st, SG); if (rc) return rc; }
            }
            k_relabel<<<SG, KGW_BLK, 0, st>>>(A, buf->scan_tmp, h);        // (+ node counts of hop h + 1)
            KGW_LAUNCH_CHECK();
        }
    }
    if (part_end < last_part) return KGW_OK;
    k_layer_tables<<<1, 64, 0, st>>>(A);
    KGW_LAUNCH_CHECK();

    // src-major structures, two layers per set of launches (the default model has two)
    for (int l0 = 1; l0 <= graph->n_layers; l0 += 2) {
        const int nl = l0 + 1 <= graph->n_layers ? 2 : 1;
        int64_t trows = 0;
        for (int k = 0; k < nl; ++k) {
            const int l = l0 + k;
            if (!buf->t_cnt[l - 1] || !buf->t_ptr[l - 1] || !buf->t_edge[l - 1] || !buf->t_zrow[l - 1] || !buf->t_tmp)
                return KGW_E_NULL;
            // histogram of the src-major rows: only the rows the layers can have need clearing -- with a static layout that is
            // the capacity of their source blocks (~1.1 M of the 5 M rows the whole graph would need: 16 MB less to write per layer)
            int64_t tr = buf->trow_cap;
            if (graph->static_layout) {
                int64_t tb = 0;
                for (int t = 0; t < graph->n_types; ++t) tb += (int64_t)graph->cap_src[l - 1][t] * graph->R_src[t];
                if (tb < tr) tr = tb;
            }
            if (tr > trows) trows = tr;
        }
        int sh, nb, nbits;
        if (ts_plan(trows, &sh, &nb, &nbits)) return KGW_E_UNSUPPORTED;  // (> 147 M src-major rows in one block)
        // (round 6, beside the training step at 1 024-block hop launches: 128 / 256 / 512 key blocks measured 1.035 / 1.034 / 1.048 ms
        //  per step -- the [bucket][block] count table is scattered traffic that grows with the blocks)
        const int nblk = SG >= 2048 ? 512 : (SG >= 256 ? 256 : 128);
        if ((int64_t)nl * (nb + 1) * nblk + (int64_t)nl * (nb + 2) > buf->scan_cap) return KGW_E_RANGE;
        k_ts_keys<<<nblk, KGW_BLK, (size_t)(nb + 1) * sizeof(int), st>>>(A, l0, nl, sh, nb);
        static KgwPerDevice attr_once;
        if (attr_once.need()) {
            KGW_HIP(hipFuncSetAttribute((const void*)k_ts_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            // (k_ts_rows: 4 << 12 / 1 << 14 counters = 64 KB of DYNAMIC LDS on top of the kernel's static words -- above the
            //  64 KB a launch may use by default; the attribute is the dynamic part only, static + dynamic <= 160 KB)
            KGW_HIP(hipFuncSetAttribute((const void*)k_ts_rows<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            KGW_HIP(hipFuncSetAttribute((const void*)k_ts_rows<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        }
        if (nblk == 128) k_ts_scan_rows<2><<<SG < 1024 ? SG : 1024, KGW_BLK, 0, st>>>(buf->scan_tmp, nl, nb, buf->meta);
        else if (nblk == 256) k_ts_scan_rows<4><<<SG < 1024 ? SG : 1024, KGW_BLK, 0, st>>>(buf->scan_tmp, nl, nb, buf->meta);
        else if (nblk == 512) k_ts_scan_rows<8><<<SG < 1024 ? SG : 1024, KGW_BLK, 0, st>>>(buf->scan_tmp, nl, nb, buf->meta);
        else return KGW_E_RANGE;
        k_ts_scan_tot<<<nl, 1024, 0, st>>>(buf->scan_tmp, nl, nb, nblk, buf->meta);
        k_ts_scatter<<<nblk, KGW_BLK, (size_t)4 * (nb + 1) * sizeof(int), st>>>(A, l0, nl, sh, nb, nbits);
        if (sh <= 12) {                   // four wavefronts per bucket while their counters fit 64 KB of LDS
            // (beside a training step, sampler grid 256: 1 024 blocks here 1.154 - 1.165 ms per step, 512: 1.175, 256: 1.195 --
            // the short launch disturbs the step less than the narrow one)
            k_ts_rows<4><<<nb < 4 * SG ? nb : 4 * SG, 256, (size_t)4 * sizeof(int) << sh, st>>>(A, l0, nl, sh, nb, nblk);
        } else {
            k_ts_rows<1><<<nb < 4 * SG ? nb : 4 * SG, 64, (size_t)sizeof(int) << sh, st>>>(A, l0, nl, sh, nb, nblk);
        }
        k_t_end<<<SG, KGW_BLK, 0, st>>>(A, l0, nl);
        KGW_LAUNCH_CHECK();
    }
    if (buf->meta_host) {
        k_meta_to_host<<<1, KGW_BLK, 0, st>>>(buf->meta, buf->meta_host);
        KGW_LAUNCH_CHECK();
    }
    return KGW_OK;
}

extern "C" int kgw_sample_batch(const KgwGraph* graph, const KgwBatchBuf* buf, const int64_t* seeds,
                                int32_t n_seeds, int32_t seed_type, int32_t full_graph,
                                kgw_stream_t stream_) {
    if (!graph) return KGW_E_NULL;
    if (graph->n_hops < 1 || graph->n_hops > KGW_MAX_LAYERS) return KGW_E_RANGE;
    return kgw_sample_batch_parts(graph, buf, seeds, n_seeds, seed_type, full_graph, 0, 2 * graph->n_hops, stream_);
}

// ---- running totals of a captured training loop (one launch instead of index / cast / add / or framework ops) ----
namespace {
__global__ void k_accumulate_stats(const KgwBatchMeta* __restrict__ M, int n_layers, int n_hops, int64_t* __restrict__ stats,
                                   int32_t* __restrict__ tick) {
    const int t = threadIdx.x;
    if (tick && t == 63) *tick += 1;                                   // (the optimiser's step counter: kgw_adam_notick)
    if (t < n_layers) stats[t] += M->n_edges[t];                       // edges aggregated by layer t+1
    else if (t == n_layers) stats[t] += M->edge_end[n_hops - 1];       // edges sampled
    else if (t == n_layers + 1) stats[t] |= M->error;                  // sticky capacity-overflow mask
}
}  // namespace

extern "C" int kgw_accumulate_stats_tick(const KgwBatchMeta* meta_dev, int32_t n_layers, int32_t n_hops, int64_t* stats,
                                         int32_t* tick, kgw_stream_t stream_) {
    if (!meta_dev || !stats) return KGW_E_NULL;
    if (n_layers < 1 || n_layers > KGW_MAX_LAYERS || n_hops < 1 || n_hops > n_layers) return KGW_E_RANGE;
    k_accumulate_stats<<<1, 64, 0, (hipStream_t)stream_>>>(meta_dev, n_layers, n_hops, stats, tick);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_accumulate_stats(const KgwBatchMeta* meta_dev, int32_t n_layers, int32_t n_hops, int64_t* stats,
                                    kgw_stream_t stream_) {
    return kgw_accumulate_stats_tick(meta_dev, n_layers, n_hops, stats, nullptr, stream_);
}

// ---- a sampled batch kept for later epochs (the loader's batch order is fixed: kgwas/kgwas.py:93-101 builds it without shuffle) ----
// One launch moves every array of a batch the training step reads -- KgwSegCopy: up to KGW_SEGCOPY_MAX (pointer, 16-byte units,
// offset in the slot) segments -- into slot ``*slot_index`` of a resident cache, or back.  The slot index is DEVICE data, so the
// launch is captured once and replayed for any batch.  HBM-bound: ~20 MB each way per 512-seed batch of the benchmark graph.
namespace {
__global__ void __launch_bounds__(KGW_BLK) k_segments_copy(KgwSegCopy P) {
    uint8_t* slot = P.slots + *P.slot_index * P.slot_stride;
    const int64_t tid = (int64_t)blockIdx.x * KGW_BLK + threadIdx.x, nthr = (int64_t)gridDim.x * KGW_BLK;
    for (int j = 0; j < P.n; ++j) {
        int4* a = (int4*)P.ptr[j];
        int4* b = (int4*)(slot + P.slot_off[j]);
        const int64_t n = P.units[j];
        if (P.to_slot) { for (int64_t k = tid; k < n; k += nthr) b[k] = a[k]; }
        else           { for (int64_t k = tid; k < n; k += nthr) a[k] = b[k]; }
    }
}
}  // namespace

extern "C" int kgw_segments_copy(const KgwSegCopy* plan, int32_t grid_blocks, kgw_stream_t stream_) {
    if (!plan || !plan->slots || !plan->slot_index) return KGW_E_NULL;
    if (plan->n < 0 || plan->n > KGW_SEGCOPY_MAX || plan->slot_stride < 0 || (plan->slot_stride & 15)) return KGW_E_RANGE;
    if ((uintptr_t)plan->slots & 15) return KGW_E_UNSUPPORTED;
    int64_t total = 0;
    for (int j = 0; j < plan->n; ++j) {
        if (!plan->ptr[j]) return KGW_E_NULL;
        if (plan->units[j] < 0 || plan->slot_off[j] < 0 || plan->slot_off[j] + 16 * plan->units[j] > plan->slot_stride) return KGW_E_RANGE;
        if (((uintptr_t)plan->ptr[j] | (uintptr_t)plan->slot_off[j]) & 15) return KGW_E_UNSUPPORTED;
        total += plan->units[j];
    }
    if (total == 0) return KGW_OK;
    int64_t g = (total + KGW_BLK - 1) / KGW_BLK;
    const int64_t cap = grid_blocks > 0 ? grid_blocks : KGW_GRID;
    if (g > cap) g = cap;
    k_segments_copy<<<(int)g, KGW_BLK, 0, (hipStream_t)stream_>>>(*plan);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

// ---- x[n_id] feature slicing ---------------------------------------------------------------------
namespace {
constexpr int GATHER_MAX_JOBS = 8;
struct GatherJobs {
    const float* src[GATHER_MAX_JOBS];
    const int32_t* ids[GATHER_MAX_JOBS];
    float* dst[GATHER_MAX_JOBS];
    int64_t first[GATHER_MAX_JOBS + 1];      // prefix sums of the jobs' element counts (float4s, or floats if width % 4)
    int n;
};

// dst_j[i] = src_j[ids_j[i]] for up to eight (src, ids, dst) jobs of one row width in ONE launch.  Threads run over the
// flattened (row, float4-of-the-row) space, so narrow rows (the 20-float SNP features: 5 float4) still fill every lane.
template <int VEC>
__global__ void __launch_bounds__(KGW_BLK) k_gather_rows(GatherJobs J, int wv) {
    const int64_t total = J.first[J.n];
    for (int64_t e = (int64_t)blockIdx.x * KGW_BLK + threadIdx.x; e < total; e += (int64_t)gridDim.x * KGW_BLK) {
        int j = 0;
        while (j + 1 < J.n && e >= J.first[j + 1]) ++j;
        const int64_t k = e - J.first[j];
        const int64_t r = k / wv;
        const int c = (int)(k - r * wv);
        const int64_t s = (int64_t)J.ids[j][r] * wv + c;
        if (VEC == 4) ((float4*)J.dst[j])[k] = ((const float4*)J.src[j])[s];
        else J.dst[j][k] = J.src[j][s];
    }
}

int launch_gather(const GatherJobs& J, int width, hipStream_t st) {
    const int64_t total = J.first[J.n];
    if (total == 0) return KGW_OK;
    int64_t grid = (total + KGW_BLK - 1) / KGW_BLK;
    if (grid > KGW_GRID * 8) grid = KGW_GRID * 8;
    if ((width & 3) == 0) k_gather_rows<4><<<(int)grid, KGW_BLK, 0, st>>>(J, width >> 2);
    else k_gather_rows<1><<<(int)grid, KGW_BLK, 0, st>>>(J, width);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}
}  // namespace

extern "C" int kgw_gather_rows(const float* src, const int32_t* ids, int64_t n_rows, int32_t width,
                               float* dst, kgw_stream_t stream_) {
    if (n_rows == 0) return KGW_OK;
    if (!src || !ids || !dst) return KGW_E_NULL;
    if (width <= 0 || n_rows < 0) return KGW_E_RANGE;
    if ((width & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15)) return KGW_E_UNSUPPORTED;
    GatherJobs J{};
    J.src[0] = src; J.ids[0] = ids; J.dst[0] = dst; J.n = 1;
    J.first[0] = 0; J.first[1] = n_rows * ((width & 3) ? width : width >> 2);
    return launch_gather(J, width, (hipStream_t)stream_);
}

extern "C" int kgw_gather_rows_multi(int32_t n_jobs, const float* const* src, const int32_t* const* ids,
                                     const int64_t* n_rows, int32_t width, float* const* dst, kgw_stream_t stream_) {
    if (n_jobs == 0) return KGW_OK;
    if (!src || !ids || !n_rows || !dst) return KGW_E_NULL;
    if (n_jobs < 0 || n_jobs > GATHER_MAX_JOBS || width <= 0) return KGW_E_RANGE;
    GatherJobs J{};
    J.n = 0; J.first[0] = 0;
    for (int j = 0; j < n_jobs; ++j) {
        if (n_rows[j] < 0) return KGW_E_RANGE;
        if (n_rows[j] == 0) continue;
        if (!src[j] || !ids[j] || !dst[j]) return KGW_E_NULL;
        if ((width & 3) == 0 && (((uintptr_t)src[j] | (uintptr_t)dst[j]) & 15)) return KGW_E_UNSUPPORTED;
        J.src[J.n] = src[j]; J.ids[J.n] = ids[j]; J.dst[J.n] = dst[j];
        J.first[J.n + 1] = J.first[J.n] + n_rows[j] * ((width & 3) ? width : width >> 2);
        ++J.n;
    }
    return launch_gather(J, width, (hipStream_t)stream_);
}
