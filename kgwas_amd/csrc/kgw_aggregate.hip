// kgw_aggregate.hip -- fused per-relation graph-attention aggregate for KGWAS on gfx950 (CDNA4).
//
// Replaces, for ALL relations of one HeteroConv layer in one launch, the >= 14 unfused tensor ops
// per relation that the reference runs through PyG (kgwas/conv.py:150-152,177,182,200-228):
//      a_s[j] = <h_src[j], u_r>                       u_r = W_src^T att_src      (conv.py:150)
//      e_ij   = leaky_relu(a_s[j] + a_d[i], 0.2)                                 (conv.py:205,217)
//      alpha  = softmax_i(e_ij / T)   (max-subtracted, denominator + 1e-16)      (conv.py:223)
//      z_i    = sum_j alpha_ij h_src[j]                                          (conv.py:227,182)
// The per-relation linear map is applied AFTER the aggregation by the caller
// (out_i = W_src z_i + bias, an exact re-association of conv.py:138-142 + :228), so the gathered rows
// are the layer input itself and every relation sharing a source type gathers the same tensor.
//
// HBM roofline kernel.  Algorithmic bytes per edge: 4 (col) + 512 (one 128-float row) + 4 (logit).
// Mapping: one wavefront per chunk (<= 256 edges of one destination row of one relation); each
// 32-lane half-wave owns one edge at a time and reads its 512-byte row as one float4 per lane
// (1 KiB per wave-instruction, 8 rows in flight per half); the dot product with u_r is reduced
// inside the half-wave with 4 DPP butterflies + one v_permlane16_swap; softmax is online
// (running max / sum), rescaled once per group of 8 edges; hub rows span several chunks whose
// partial (max, sum, acc) are merged by a second tiny kernel.  No atomics anywhere.
//
// Backward is split the same way the data is laid out: a dst-major pass (re-gathers h_src rows,
// produces per-edge (alpha, d pre-activation) and d a_dst) and a src-major pass over the transposed
// structure built by the sampler (gathers dZ rows, produces dH and d a_src).
#include "kgw_common.h"
#include <math.h>

namespace {

constexpr int PART_STRIDE = 132;      // floats per partial record: [0]=max [1]=sum [4..131]=acc
constexpr float NEG_BIG = -1.0e30f;

struct LayerTab {
    int32_t n_rels, n_types;
    int32_t src_base[KGW_MAX_RELS];   // first H row of the relation's source type
    int32_t z0[KGW_MAX_RELS];         // z_base[dst type] + slot_dst
    int32_t zstride[KGW_MAX_RELS];    // R_dst[dst type]
    int32_t live[KGW_MAX_RELS];
    // src-major view
    int32_t type_src_base[KGW_MAX_TYPES + 1];
    int32_t type_t_base[KGW_MAX_TYPES + 1];
    int32_t type_R_src[KGW_MAX_TYPES];
    int8_t rel_of_slot[KGW_MAX_TYPES][KGW_MAX_RELS / 2];    // relation id of (source type, slot)
    int8_t rel_src_type[KGW_MAX_RELS];                      // source type of relation r
    int8_t rel_slot_src[KGW_MAX_RELS];                      // its slot among the relations of that source type
    int32_t ld_da;                                          // n_rels rounded up to 4; da_src rows are 2*ld_da wide
    // destination-side view (in-kernel a_d = <h_dst, v_r>): the rows of a destination type are the first rows of
    // its own block of H
    int32_t dst_hbase[KGW_MAX_RELS];                        // first H row of the relation's DESTINATION type
    int8_t rel_dst_type[KGW_MAX_RELS];
    int8_t rel_slot_dst[KGW_MAX_RELS];
    int8_t rel_of_dslot[KGW_MAX_TYPES][KGW_MAX_RELS / 2];   // relation id of (destination type, slot)
    int32_t type_z_base[KGW_MAX_TYPES];
    int32_t type_R_dst[KGW_MAX_TYPES];
    uint64_t partial;                                       // bit r: relation r leaves PARTIAL softmax states (sharded mode)
    int32_t oct_rows;                                       // rows [0, oct_rows) (a multiple of 8): node types of the short-row hint
};

struct AggPtrs {
    const KgwChunk* chunks;
    const int32_t* col_local;
    const float* H;
    const float* a_dst;
    const float* V;               // [n_rels][128] v_r (nullable: then a_dst is read)
    const float* U;
    const float* lbias;           // [n_rels] constant of the pre-activation logit (nullable)
    float* Z;
    float* stat;
    float* e_edge;
    float* part;
    const float* dZ;
    float* adp;
    float* da_dst;
    float* part_da;
    const int32_t* t_ptr;
    const int32_t* t_edge;
    const int32_t* t_zrow;
    const uint8_t* t_rel;         // relation id per src-major entry (nullable: no octet path)
    const int32_t* oct_flags;     // per 8 source rows: processed by the octet path
    float* part_du;               // [n_chunks][128] per-chunk part of d u_r (nullable)
    const int32_t* seg_chptr;     // first chunk of every segment (for the d u_r reduction)
    float* duv_ws;                // [2][n_rels][8][128] level-1 sums of d u_r / d v_r
    int n_duv_hops;
    float* dH;
    float* da_src;
    const int32_t* multi;
    int64_t multi_cap;
    const KgwBatchMeta* meta;     // device: actual counts of the batch
    int layer;
    int raw;                      // forward: raw-logit weights (attention export)
    int relu_in;                  // bwd_src: dH *= (H > 0)
};

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
__device__ __forceinline__ void fma4(float4& acc, float w, const float4& x) {
    acc.x = fmaf(w, x.x, acc.x); acc.y = fmaf(w, x.y, acc.y);
    acc.z = fmaf(w, x.z, acc.z); acc.w = fmaf(w, x.w, acc.w);
}
__device__ __forceinline__ void scale4(float4& a, float s) { a.x *= s; a.y *= s; a.z *= s; a.w *= s; }

__device__ __forceinline__ KgwChunk load_chunk(const KgwChunk* chunks, int c) {
    // c is wave-uniform: read through the scalar path
    const int cu = __builtin_amdgcn_readfirstlane(c);
    return chunks[cu];
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// RAW: the messages are weighted by the leaky_relu logits themselves, no softmax -- what the reference's
// attention export computes (conv.py:221-223 skips the softmax under return_raw_attention_weights and
// message() :227-228 multiplies by that alpha; kgwas/utils.py:446-461).
template <int G, bool RAW>
__device__ __forceinline__ void fwd_group(const float4* __restrict__ Hb4, int colv, int q0, int hn, int nb,
                                          int half, int hl, const float4& u4, float ad, float slope,
                                          float inv_temp, float& m, float& s, float4& acc, float& ev) {
    float4 x[G];
    bool valid[G];
#pragma unroll
    for (int p = 0; p < G; ++p) {
        const int q = q0 + p;
        const int i = half * hn + q;
        valid[p] = (q < hn) && (i < nb);
        const int cj = __shfl(colv, valid[p] ? i : 0, 64);
        x[p] = Hb4[(int64_t)cj * 32 + hl];
    }
    float t[G];
    float mb = -INFINITY;
#pragma unroll
    for (int p = 0; p < G; ++p) {
        float d = kgw_half_allsum(dot4(x[p], u4)) + ad;
        d = d > 0.f ? d : d * slope;
        ev = (hl == q0 + p) ? d : ev;              // lane (half, hl) keeps the logit of edge half*hn+hl
        if (RAW) { t[p] = valid[p] ? d : 0.f; continue; }
        t[p] = valid[p] ? d * inv_temp : -INFINITY;
        mb = fmaxf(mb, t[p]);
    }
    if (RAW) {
#pragma unroll
        for (int p = 0; p < G; ++p) fma4(acc, t[p], x[p]);
        return;
    }
    const float mn = fmaxf(m, mb);
    const float sc = __expf(m - mn);
    s *= sc; scale4(acc, sc); m = mn;
#pragma unroll
    for (int p = 0; p < G; ++p) {
        const float w = __expf(t[p] - mn);
        s += w;
        fma4(acc, w, x[p]);
    }
}

__device__ __forceinline__ void grp8_load(const float4* __restrict__ Hb4, int colv, int q0, int hn, int nb, int half, int hl,
                                          float4 (&x)[8]) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int q = q0 + p, i = half * hn + q;
        const bool ok = (q < hn) && (i < nb);
        const int cj = __shfl(colv, ok ? i : 0, 64);
        x[p] = Hb4[(int64_t)cj * 32 + hl];
    }
}

template <bool RAW>
__device__ __forceinline__ void fwd_grp8_compute(const float4 (&x)[8], int q0, int hn, int nb, int half, int hl,
                                                 const float4& u4, float ad, float slope, float inv_temp, float& m, float& s,
                                                 float4& acc, float& ev) {
    float part[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) part[p] = dot4(x[p], u4);
    const int pm = hl & 7, qm = q0 + pm;
    const bool valid = (qm < hn) && (half * hn + qm < nb);
    float d = kgw_half_reduce8(part, hl) + ad;                 // logit of edge q0 + (hl & 7)
    d = d > 0.f ? d : d * slope;
    ev = ((hl >> 3) == (q0 >> 3) && hl == qm) ? d : ev;        // lane (half, q) keeps the logit of edge half*hn + q
    float w;
    if (RAW) {
        w = valid ? d : 0.f;
    } else {
        const float t = valid ? d * inv_temp : -INFINITY;
        const float mn = fmaxf(m, kgw_max8(t));
        const float sc = __expf(m - mn);
        s *= sc; scale4(acc, sc); m = mn;
        w = __expf(t - mn);                                     // 0 for invalid edges (t = -inf, mn finite: m starts at NEG_BIG)
        s += kgw_sum8(w);
    }
    fma4(acc, kgw_bcast8<0>(w), x[0]); fma4(acc, kgw_bcast8<1>(w), x[1]);
    fma4(acc, kgw_bcast8<2>(w), x[2]); fma4(acc, kgw_bcast8<3>(w), x[3]);
    fma4(acc, kgw_bcast8<4>(w), x[4]); fma4(acc, kgw_bcast8<5>(w), x[5]);
    fma4(acc, kgw_bcast8<6>(w), x[6]); fma4(acc, kgw_bcast8<7>(w), x[7]);
}

// Eight edges per half with the transposed reduction (kgw_half_reduce8): the lane with (hl & 7) == p owns edge q0 + p --
// its logit, its softmax weight -- and hands the weight to the other lanes of its 8-lane group with one swizzle.
template <bool RAW>
__device__ __forceinline__ void fwd_group8(const float4* __restrict__ Hb4, int colv, int q0, int hn, int nb,
                                           int half, int hl, const float4& u4, float ad, float slope,
                                           float inv_temp, float& m, float& s, float4& acc, float& ev) {
    float4 x[8];
    grp8_load(Hb4, colv, q0, hn, nb, half, hl, x);
    fwd_grp8_compute<RAW>(x, q0, hn, nb, half, hl, u4, ad, slope, inv_temp, m, s, acc, ev);
}

// PIPE: software pipelining inside a chunk.  A chunk of n edges used to cost one memory round trip for the column ids of
// every 64 edges plus one per group of 16 rows (a 256-edge chunk of a hub row: 20 dependent round trips, and the longest
// chunks set the kernel's tail -- halving KGW_CHUNK alone took 64 -> 58 us); with PIPE the next block's column ids and
// the next group's rows are requested before the current group is reduced.
template <bool RAW, bool PIPE>
__global__ void __launch_bounds__(KGW_BLK) k_agg_fwd(LayerTab T, AggPtrs P, float slope, float inv_temp) {
    const int lane = kgw_lane(), half = lane >> 5, hl = lane & 31;
    const int nw = gridDim.x * 4;
    const int n_items = P.meta->n_chunks[P.layer - 1];
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < n_items; c += nw) {
        const KgwChunk ck = load_chunk(P.chunks, c);
        const int r = ck.rel;
        if (!T.live[r]) continue;
        const int zrow = T.z0[r] + ck.row * T.zstride[r];
        float ad;
        if (P.V) {          // a_d = <h_dst[row], v_r> (conv.py:150-151 re-associated), both halves compute it
            const float4 hd4 = ((const float4*)(P.H + ((int64_t)T.dst_hbase[r] + ck.row) * KGW_C))[hl];
            const float4 v4 = ((const float4*)(P.V + (int64_t)r * KGW_C))[hl];
            ad = kgw_half_allsum(dot4(hd4, v4));
        } else {
            ad = P.a_dst[zrow];
        }
        if (P.lbias) ad += P.lbias[r];
        const float4 u4 = ((const float4*)(P.U + (int64_t)r * KGW_C))[hl];
        const float4* Hb4 = (const float4*)(P.H + (int64_t)T.src_base[r] * KGW_C);
        float m = NEG_BIG, s = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int n = ck.e1 - ck.e0;
        int colv_next = (PIPE && lane < min(64, n)) ? P.col_local[ck.e0 + lane] : 0;
        for (int b = 0; b < n; b += 64) {
            const int nb = min(64, n - b);
            const int hn = (nb + 1) >> 1;
            int colv;
            if (PIPE) {
                colv = colv_next;
                colv_next = (b + 64 < n && lane < min(64, n - b - 64)) ? P.col_local[ck.e0 + b + 64 + lane] : 0;
            } else {
                colv = (lane < nb) ? P.col_local[ck.e0 + b + lane] : 0;
            }
            float ev = 0.f;
            int q0 = 0;
            if (PIPE && hn > 4) {
                float4 x0[8], x1[8];
                grp8_load(Hb4, colv, 0, hn, nb, half, hl, x0);
                while (true) {
                    const int q1 = q0 + 8;
                    const bool more1 = hn - q1 > 4;
                    if (more1) grp8_load(Hb4, colv, q1, hn, nb, half, hl, x1);
                    fwd_grp8_compute<RAW>(x0, q0, hn, nb, half, hl, u4, ad, slope, inv_temp, m, s, acc, ev);
                    q0 = q1;
                    if (!more1) break;
                    const int q2 = q1 + 8;
                    const bool more2 = hn - q2 > 4;
                    if (more2) grp8_load(Hb4, colv, q2, hn, nb, half, hl, x0);
                    fwd_grp8_compute<RAW>(x1, q1, hn, nb, half, hl, u4, ad, slope, inv_temp, m, s, acc, ev);
                    q0 = q2;
                    if (!more2) break;
                }
            }
            for (; q0 < hn;) {
                const int rem = hn - q0;
                if (rem > 4)      { fwd_group8<RAW>(Hb4, colv, q0, hn, nb, half, hl, u4, ad, slope, inv_temp, m, s, acc, ev); q0 += 8; }
                else if (rem > 2) { fwd_group<4, RAW>(Hb4, colv, q0, hn, nb, half, hl, u4, ad, slope, inv_temp, m, s, acc, ev); q0 += 4; }
                else              { fwd_group<2, RAW>(Hb4, colv, q0, hn, nb, half, hl, u4, ad, slope, inv_temp, m, s, acc, ev); q0 += 2; }
            }
            const int i = half * hn + hl;
            if (hl < hn && i < nb) P.e_edge[ck.e0 + b + i] = ev;
        }
        // merge the two half-wave states
        const float mo = kgw_xhalf(m);
        const float M = RAW ? 0.f : fmaxf(m, mo);
        const float f = RAW ? 1.f : __expf(m - M);
        s *= f; scale4(acc, f);
        const float S = RAW ? 1.f : s + kgw_xhalf(s);
        acc.x += kgw_xhalf(acc.x); acc.y += kgw_xhalf(acc.y);
        acc.z += kgw_xhalf(acc.z); acc.w += kgw_xhalf(acc.w);
        if (ck.nch == 1) {
            // partial state (SNP-sharded mode): unnormalised sum and (max, sum of exponentials), merged across ranks later
            const bool partial = !RAW && ((T.partial >> r) & 1ull);
            const float den = RAW ? 1.f : (partial ? S : S + 1e-16f);
            const float inv = partial ? 1.f : 1.0f / den;
            if (half == 0) {
                scale4(acc, inv);
                ((float4*)(P.Z + (int64_t)zrow * KGW_C))[hl] = acc;
            }
            if (lane == 0) { P.stat[2 * (int64_t)zrow] = M; P.stat[2 * (int64_t)zrow + 1] = den; }
        } else {
            float* pr = P.part + (int64_t)c * PART_STRIDE;
            if (half == 0) ((float4*)(pr + 4))[hl] = acc;
            if (lane == 0) { pr[0] = M; pr[1] = S; }
        }
    }
}

__device__ __forceinline__ float wave_allmax(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_allsum_slow(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one wavefront per multi-chunk segment: merge partial (max, sum, acc)
__global__ void __launch_bounds__(KGW_BLK) k_agg_fwd_combine(LayerTab T, AggPtrs P, int n_hops) {
  for (int hop = 0; hop < n_hops; ++hop) {     // (one launch for the multi-chunk lists of every destination hop)
    const int lane = kgw_lane();
    const int nw = gridDim.x * 4;
    const int n_multi = P.meta->multi_cnt[hop];
    const int32_t* mm = P.multi + (int64_t)hop * P.multi_cap * 4;
    for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < n_multi; k += nw) {
        const int first = mm[4 * k], nch = mm[4 * k + 1], row = mm[4 * k + 2], r = mm[4 * k + 3];
        if (!T.live[r]) continue;
        const int zrow = T.z0[r] + row * T.zstride[r];
        float mx = NEG_BIG;
        for (int c = lane; c < nch; c += 64) mx = fmaxf(mx, P.part[(int64_t)(first + c) * PART_STRIDE]);
        const float M = wave_allmax(mx);
        float S = 0.f;
        float2 acc = make_float2(0.f, 0.f);
        int c = 0;
        for (; c + 4 <= nch; c += 4) {               // four partial records in flight (the loads are independent)
            const float* pr = P.part + (int64_t)(first + c) * PART_STRIDE;
            float mq[4], sq[4];
            float2 aq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mq[q] = pr[q * PART_STRIDE]; sq[q] = pr[q * PART_STRIDE + 1];
                aq[q] = ((const float2*)(pr + q * PART_STRIDE + 4))[lane];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {              // (same order as the plain loop: chunk order)
                const float f = __expf(mq[q] - M);
                S = fmaf(sq[q], f, S);
                acc.x = fmaf(aq[q].x, f, acc.x); acc.y = fmaf(aq[q].y, f, acc.y);
            }
        }
        for (; c < nch; ++c) {
            const float* pr = P.part + (int64_t)(first + c) * PART_STRIDE;
            const float f = __expf(pr[0] - M);
            S = fmaf(pr[1], f, S);
            const float2 a = ((const float2*)(pr + 4))[lane];
            acc.x = fmaf(a.x, f, acc.x); acc.y = fmaf(a.y, f, acc.y);
        }
        // (raw mode: every partial carries max 0 and sum 1, so f == 1 above and the sum of partials is the result)
        const bool partial = !P.raw && ((T.partial >> r) & 1ull);
        const float den = P.raw ? 1.f : (partial ? S : S + 1e-16f), inv = partial ? 1.f : 1.0f / den;
        ((float2*)(P.Z + (int64_t)zrow * KGW_C))[lane] = make_float2(acc.x * inv, acc.y * inv);
        if (lane == 0) { P.stat[2 * (int64_t)zrow] = M; P.stat[2 * (int64_t)zrow + 1] = den; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, dst-major pass
// ------------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ void bwd_group(const float4* __restrict__ Hb4, int colv, float evin, int q0, int hn,
                                          int nb, int half, int hl, const float4& dz4, float cdot, float M,
                                          float inv_den, float slope, float inv_temp, float& av, float& dv,
                                          float& dsum, float4& ua, float& esum, float& bsum, float& ssum) {
    float4 x[G];
    float t[G];
#pragma unroll
    for (int p = 0; p < G; ++p) {
        const int q = q0 + p;
        const int i = half * hn + q;
        const bool valid = (q < hn) && (i < nb);
        const int cj = __shfl(colv, valid ? i : 0, 64);
        x[p] = Hb4[(int64_t)cj * 32 + hl];
        const float e = __shfl(evin, half * 32 + (q & 31), 64);   // logit kept by lane (half, q)
        t[p] = valid ? e : -INFINITY;
    }
#pragma unroll
    for (int p = 0; p < G; ++p) {
        const float dalpha = kgw_half_allsum(dot4(x[p], dz4));
        const float alpha = __expf(t[p] * inv_temp - M) * inv_den;
        const float dlogit = alpha * (dalpha - cdot);
        const float sfac = inv_temp * (t[p] > 0.f ? 1.0f : slope);
        const float dpre = dlogit * sfac;
        av = (hl == q0 + p) ? alpha : av;
        dv = (hl == q0 + p) ? dpre : dv;
        dsum += dpre;
        esum += dlogit; bsum += alpha * sfac; ssum += alpha;       // (the row's consistent d a_dst: see k_agg_bwd_dst)
        fma4(ua, dpre, x[p]);                      // d u_r += d pre-activation * h_src (every lane has the edge's dpre here)
    }
}

// Eight edges per half, transposed reduction (see fwd_group8): the lane with (hl & 7) == p owns edge q0 + p.  `dsum`
// here collects only the lane's OWN edges; the caller folds the 8 residues once per chunk (kgw_sum8).
__device__ __forceinline__ void bwd_grp8_compute(const float4 (&x)[8], float evin, int q0, int hn, int nb, int half, int hl,
                                                 const float4& dz4, float cdot, float M, float inv_den, float slope,
                                                 float inv_temp, float& av, float& dv, float& dsum_own, float4& ua,
                                                 float& esum_own, float& bsum_own, float& ssum_own) {
    float part[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) part[p] = dot4(x[p], dz4);
    const int pm = hl & 7, qm = q0 + pm;
    const bool valid = (qm < hn) && (half * hn + qm < nb);
    const float e = __shfl(evin, half * 32 + (qm & 31), 64);     // logit kept by lane (half, q)
    const float t = valid ? e : -INFINITY;
    const float dalpha = kgw_half_reduce8(part, hl);
    const float alpha = __expf(t * inv_temp - M) * inv_den;
    const float dlogit = alpha * (dalpha - cdot);
    const float sfac = inv_temp * (t > 0.f ? 1.0f : slope);
    const float dpre = dlogit * sfac;
    const bool mine = (hl == qm);
    av = mine ? alpha : av;
    dv = mine ? dpre : dv;
    dsum_own += (hl < 8) ? dpre : 0.f;          // one copy per edge: the 8 residues of the first 8-lane group
    esum_own += (hl < 8) ? dlogit : 0.f;
    bsum_own += (hl < 8) ? alpha * sfac : 0.f;
    ssum_own += (hl < 8) ? alpha : 0.f;
    // d u_r += d pre-activation(edge) * h_src(edge): lane p of the half owns edge q0 + p's value (0 for a padding edge)
    // (two scalar lane reads + a select per edge instead of a cross-lane permute through the LDS crossbar)
    const int dbits = __builtin_bit_cast(int, dpre);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const float d0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(dbits, p));
        const float d1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(dbits, 32 + p));
        fma4(ua, half ? d1 : d0, x[p]);
    }
}

__device__ __forceinline__ void bwd_group8(const float4* __restrict__ Hb4, int colv, float evin, int q0, int hn,
                                           int nb, int half, int hl, const float4& dz4, float cdot, float M,
                                           float inv_den, float slope, float inv_temp, float& av, float& dv,
                                           float& dsum_own, float4& ua, float& esum_own, float& bsum_own, float& ssum_own) {
    float4 x[8];
    grp8_load(Hb4, colv, q0, hn, nb, half, hl, x);
    bwd_grp8_compute(x, evin, q0, hn, nb, half, hl, dz4, cdot, M, inv_den, slope, inv_temp, av, dv, dsum_own, ua,
                     esum_own, bsum_own, ssum_own);
}

template <bool PIPE>
__global__ void __launch_bounds__(KGW_BLK) __attribute__((amdgpu_waves_per_eu(4, 4))) k_agg_bwd_dst(LayerTab T, AggPtrs P, float slope, float inv_temp) {
    const int lane = kgw_lane(), half = lane >> 5, hl = lane & 31;
    const int nw = gridDim.x * 4;
    const int n_items = P.meta->n_chunks[P.layer - 1];
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < n_items; c += nw) {
        const KgwChunk ck = load_chunk(P.chunks, c);
        const int r = ck.rel;
        if (!T.live[r]) continue;
        const int zrow = T.z0[r] + ck.row * T.zstride[r];
        const float4 dz4 = ((const float4*)(P.dZ + (int64_t)zrow * KGW_C))[hl];
        const float4 z4 = ((const float4*)(P.Z + (int64_t)zrow * KGW_C))[hl];
        // sum_k alpha_ik dalpha_ik = <dz_i, z_i>: from the STORED fp32 aggregate -- one rounding away from the sum the edges below
        // would give, which is all that is left of d a_dst = sum_e dpre_e wherever a row's logits sit on one branch of the leaky
        // ReLU (it is 0 in exact arithmetic: the softmax of conv.py:223 does not see a common shift).  Round 5: the per-edge values
        // keep this provisional c0 (they are not cancellation residues), the ROW SUM is made consistent with the edges' own
        // d alpha: with E = sum a_e (dalpha_e - c0), B = sum a_e s_e, S = sum a_e (s_e = leaky slope / T of the edge),
        //     d a_dst = sum_e a_e s_e (dalpha_e - c0) - (E / S) B  =  sum_e a_e s_e (dalpha_e - c*),  c* = sum a_e dalpha_e / sum a_e
        // -- what autograd of the reference's softmax computes, edge by edge, in differences (dalpha_e - c0) that are small.
        const float cdot = kgw_half_allsum(dot4(dz4, z4));
        const float M = P.stat[2 * (int64_t)zrow];
        const float inv_den = 1.0f / P.stat[2 * (int64_t)zrow + 1];
        const float4* Hb4 = (const float4*)(P.H + (int64_t)T.src_base[r] * KGW_C);
        float dsum = 0.f, dsum_own = 0.f;
        float esum = 0.f, bsum = 0.f, ssum = 0.f, esum_own = 0.f, bsum_own = 0.f, ssum_own = 0.f;
        float4 ua = make_float4(0.f, 0.f, 0.f, 0.f);          // this chunk's part of d u_r: sum_e dpre_e h_src(e) (a half's edges)
        const int n = ck.e1 - ck.e0;
        int colv_next = (PIPE && lane < min(64, n)) ? P.col_local[ck.e0 + lane] : 0;
        float ev_next = 0.f;
        if (PIPE) {
            const int nb0 = min(64, n), hn0 = (nb0 + 1) >> 1, i0 = half * hn0 + hl;
            ev_next = (hl < hn0 && i0 < nb0) ? P.e_edge[ck.e0 + i0] : 0.f;
        }
        for (int b = 0; b < n; b += 64) {
            const int nb = min(64, n - b);
            const int hn = (nb + 1) >> 1;
            const int i = half * hn + hl;
            const bool mine = (hl < hn) && (i < nb);
            int colv;
            float evin;
            if (PIPE) {
                colv = colv_next; evin = ev_next;
                const int nb2 = min(64, n - b - 64), hn2 = (nb2 + 1) >> 1, i2 = half * hn2 + hl;
                const bool nxt = b + 64 < n;
                colv_next = (nxt && lane < nb2) ? P.col_local[ck.e0 + b + 64 + lane] : 0;
                ev_next = (nxt && hl < hn2 && i2 < nb2) ? P.e_edge[ck.e0 + b + 64 + i2] : 0.f;
            } else {
                colv = (lane < nb) ? P.col_local[ck.e0 + b + lane] : 0;
                evin = mine ? P.e_edge[ck.e0 + b + i] : 0.f;
            }
            float av = 0.f, dv = 0.f;
            int q0 = 0;
            if (PIPE && hn > 4) {
                float4 x0[8], x1[8];
                grp8_load(Hb4, colv, 0, hn, nb, half, hl, x0);
                while (true) {
                    const int q1 = q0 + 8;
                    const bool more1 = hn - q1 > 4;
                    if (more1) grp8_load(Hb4, colv, q1, hn, nb, half, hl, x1);
                    bwd_grp8_compute(x0, evin, q0, hn, nb, half, hl, dz4, cdot, M, inv_den, slope, inv_temp, av, dv, dsum_own, ua,
                                     esum_own, bsum_own, ssum_own);
                    q0 = q1;
                    if (!more1) break;
                    const int q2 = q1 + 8;
                    const bool more2 = hn - q2 > 4;
                    if (more2) grp8_load(Hb4, colv, q2, hn, nb, half, hl, x0);
                    bwd_grp8_compute(x1, evin, q1, hn, nb, half, hl, dz4, cdot, M, inv_den, slope, inv_temp, av, dv, dsum_own, ua,
                                     esum_own, bsum_own, ssum_own);
                    q0 = q2;
                    if (!more2) break;
                }
            }
            for (; q0 < hn;) {
                const int rem = hn - q0;
                if (rem > 4)      { bwd_group8(Hb4, colv, evin, q0, hn, nb, half, hl, dz4, cdot, M, inv_den, slope, inv_temp, av, dv, dsum_own, ua, esum_own, bsum_own, ssum_own); q0 += 8; }
                else if (rem > 2) { bwd_group<4>(Hb4, colv, evin, q0, hn, nb, half, hl, dz4, cdot, M, inv_den, slope, inv_temp, av, dv, dsum, ua, esum, bsum, ssum); q0 += 4; }
                else              { bwd_group<2>(Hb4, colv, evin, q0, hn, nb, half, hl, dz4, cdot, M, inv_den, slope, inv_temp, av, dv, dsum, ua, esum, bsum, ssum); q0 += 2; }
            }
            if (mine) ((float2*)P.adp)[ck.e0 + b + i] = make_float2(av, dv);
        }
        dsum += kgw_sum8(dsum_own);                            // lanes 0..7 of each half: the edges handled 8 at a time
        if (P.part_du) {
            ua.x += kgw_xhalf(ua.x); ua.y += kgw_xhalf(ua.y); ua.z += kgw_xhalf(ua.z); ua.w += kgw_xhalf(ua.w);
            if (half == 0) ((float4*)(P.part_du + (int64_t)c * KGW_C))[hl] = ua;
        }
        esum += kgw_sum8(esum_own); bsum += kgw_sum8(bsum_own); ssum += kgw_sum8(ssum_own);
        const float tot = dsum + kgw_xhalf(dsum);
        const float te = esum + kgw_xhalf(esum), tb = bsum + kgw_xhalf(bsum), ts = ssum + kgw_xhalf(ssum);
        if (lane == 0) {
            // (a relation whose segments hold only THIS rank's part of the edges -- SNP-sharded mode -- keeps the plain sum: c* is
            //  a property of the whole row, and the ranks' d a_dst are added as they are)
            const bool partial = (T.partial >> r) & 1ull;
            if (ck.nch == 1) P.da_dst[zrow] = partial ? tot : tot - (te / ts) * tb;
            else ((float4*)P.part_da)[c] = make_float4(tot, te, tb, ts);
        }
    }
}

__global__ void __launch_bounds__(KGW_BLK) k_agg_bwd_combine(LayerTab T, AggPtrs P, int n_hops) {
  for (int hop = 0; hop < n_hops; ++hop) {     // (one launch for the multi-chunk lists of every destination hop)
    const int lane = kgw_lane();
    const int nw = gridDim.x * 4;
    const int n_multi = P.meta->multi_cnt[hop];
    const int32_t* mm = P.multi + (int64_t)hop * P.multi_cap * 4;
    for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < n_multi; k += nw) {
        const int first = mm[4 * k], nch = mm[4 * k + 1], row = mm[4 * k + 2], r = mm[4 * k + 3];
        if (!T.live[r]) continue;
        const int zrow = T.z0[r] + row * T.zstride[r];
        // fixed-order tree: lane-strided partial sums, then butterfly -- of the four per-chunk sums (D0, E, B, S) of k_agg_bwd_dst
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = lane; c < nch; c += 64) {
            const float4 p = ((const float4*)P.part_da)[first + c];
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
        s.x = wave_allsum_slow(s.x); s.y = wave_allsum_slow(s.y); s.z = wave_allsum_slow(s.z); s.w = wave_allsum_slow(s.w);
        if (lane == 0) P.da_dst[zrow] = ((T.partial >> r) & 1ull) ? s.x : s.x - (s.y / s.w) * s.z;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, src-major pass: one wavefront per source row
// ------------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ void src_group(const float4* __restrict__ dZ4, int tz, float al, int q0, int hn,
                                          int nb, int half, int hl, float4& acc) {
    float4 x[G];
    float w[G];
#pragma unroll
    for (int p = 0; p < G; ++p) {
        const int q = q0 + p;
        const int i = half * hn + q;
        const bool valid = (q < hn) && (i < nb);
        const int z = __shfl(tz, valid ? i : 0, 64);
        const float a = __shfl(al, valid ? i : 0, 64);
        w[p] = valid ? a : 0.f;
        x[p] = dZ4[(int64_t)z * 32 + hl];
    }
#pragma unroll
    for (int p = 0; p < G; ++p) fma4(acc, w[p], x[p]);
}

// one wavefront, one source row (rows with many entries; rows whose neighbour cannot be paired)
__device__ __forceinline__ void bwd_src_one_row(const LayerTab& T, const AggPtrs& P, int u) {
    const int lane = kgw_lane(), half = lane >> 5, hl = lane & 31;
    const float4* dZ4 = (const float4*)P.dZ;
    {
        int ty = 0;
        while (ty + 1 < T.n_types && u >= T.type_src_base[ty + 1]) ++ty;
        const int j = u - T.type_src_base[ty];
        const int Rs = T.type_R_src[ty];
        const int tb = T.type_t_base[ty] + j * Rs;
        if (j >= P.meta->n_src[P.layer - 1][ty]) {          // padding row of a static layout: no gradient
            if (half == 0) ((float4*)(P.dH + (int64_t)u * KGW_C))[hl] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (P.da_src) for (int c = lane; c < 2 * T.ld_da; c += 64) P.da_src[(int64_t)u * 2 * T.ld_da + c] = 0.f;
            return;
        }
        // all Rs + 1 row pointers of this source in ONE load (lane k holds t_ptr[tb + k]); the per-slot logic below
        // works on shuffles of it instead of a chain of dependent scalar loads
        const int tpv = (lane <= Rs) ? P.t_ptr[tb + lane] : 0;
        const int p0 = __shfl(tpv, 0, 64), p1 = __shfl(tpv, Rs, 64);
        const int tpn = __shfl_down(tpv, 1, 64);
        const unsigned long long slots = __ballot(lane < Rs && tpn > tpv);      // bit k: slot k has entries
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float dasv = 0.f;                                    // lane k holds d a_src of slot k
        for (int pb = p0; pb < p1; pb += 64) {
            const int nb = min(64, p1 - pb);
            const int hn = (nb + 1) >> 1;
            int te = 0, tz = 0;
            float al = 0.f, dp = 0.f;
            if (lane < nb) {
                te = P.t_edge[pb + lane];
                tz = P.t_zrow[pb + lane];
                const float2 a2 = ((const float2*)P.adp)[te];
                al = a2.x; dp = a2.y;
            }
            // per-slot sums of d pre-activation (entries of one source are grouped by slot): ONE segmented inclusive scan
            // over the 64 lanes (six shuffle steps, fixed tree) instead of a full-wave reduction per slot; the segment
            // heads come from the row pointers with scalar work only, and each slot's total is read at its last lane
            unsigned long long heads = 1ull;
            for (unsigned long long left = slots; left; left &= left - 1) {
                const int k = __builtin_ctzll(left);
                const int s0 = __builtin_amdgcn_readlane(tpv, k);
                if (s0 > pb && s0 < pb + nb) heads |= 1ull << (s0 - pb);
            }
            float sv = (lane < nb) ? dp : 0.f;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float o = __shfl_up(sv, d, 64);
                const bool open = lane >= d && ((heads >> (lane - d + 1)) & ((1ull << d) - 1ull)) == 0ull;
                sv += open ? o : 0.f;
            }
            for (unsigned long long left = slots; left; left &= left - 1) {
                const int k = __builtin_ctzll(left);
                const int s0 = __builtin_amdgcn_readlane(tpv, k), s1 = __builtin_amdgcn_readlane(tpv, k + 1);
                if (s1 <= pb || s0 >= pb + nb) continue;
                const int last = min(s1, pb + nb) - 1 - pb;
                const float sk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sv), last));
                dasv += (lane == k) ? sk : 0.f;
            }
            for (int q0 = 0; q0 < hn;) {
                const int rem = hn - q0;
                if (rem > 4)      { src_group<8>(dZ4, tz, al, q0, hn, nb, half, hl, acc); q0 += 8; }
                else if (rem > 2) { src_group<4>(dZ4, tz, al, q0, hn, nb, half, hl, acc); q0 += 4; }
                else              { src_group<2>(dZ4, tz, al, q0, hn, nb, half, hl, acc); q0 += 2; }
            }
        }
        acc.x += kgw_xhalf(acc.x); acc.y += kgw_xhalf(acc.y);
        acc.z += kgw_xhalf(acc.z); acc.w += kgw_xhalf(acc.w);
        if (p1 > p0) {
            // d a_src flows back into h_src through a_s = <h_src, u_r>
            for (unsigned long long left = slots; left; left &= left - 1) {
                const int k = __builtin_ctzll(left);
                const float dk = __shfl(dasv, k, 64);
                if (dk != 0.f) {
                    const int r = T.rel_of_slot[ty][k];
                    const float4 u4 = ((const float4*)(P.U + (int64_t)r * KGW_C))[hl];
                    fma4(acc, dk, u4);
                }
            }
        }
        // destination side of the same node: a_d[i, r] = <h[i], v_r> sends d a_d back into h[i]
        const int Rd = T.type_R_dst[ty];
        const bool is_dst = P.V && Rd > 0 && j < P.meta->n_rows[P.layer - 1][ty];
        const int zb = T.type_z_base[ty] + j * Rd;
        if (is_dst) {
            const float dav = (lane < Rd) ? P.da_dst[zb + lane] : 0.f;
            for (int k = 0; k < Rd; ++k) {
                const float dk = __shfl(dav, k, 64);
                if (dk != 0.f) {
                    const int r = T.rel_of_dslot[ty][k];
                    const float4 v4 = ((const float4*)(P.V + (int64_t)r * KGW_C))[hl];
                    fma4(acc, dk, v4);
                }
            }
        }
        if (P.relu_in) {
            const float4 h4 = ((const float4*)(P.H + (int64_t)u * KGW_C))[hl];
            acc.x = h4.x > 0.f ? acc.x : 0.f; acc.y = h4.y > 0.f ? acc.y : 0.f;
            acc.z = h4.z > 0.f ? acc.z : 0.f; acc.w = h4.w > 0.f ? acc.w : 0.f;
        }
        if (half == 0) ((float4*)(P.dH + (int64_t)u * KGW_C))[hl] = acc;
        // [d a_src | d a_dst] row of this node, one column per RELATION ID (zero for relations of other types): the
        // caller gets d u_r = sum_j d a_src[j, r] H[j] and d v_r = sum_i d a_dst[i, r] H[i] for all relations as
        // ONE tall-skinny product over H  (not written when the caller takes d u_r / d v_r from the launch's riders)
        if (P.da_src) {
            const int ld = T.ld_da;                           // up to 64 relations: a row of up to 128 floats, two per lane
            for (int c0 = 0; c0 < 2 * ld; c0 += 64) {
                const int col = c0 + lane;
                const bool mine = col < T.n_rels && T.rel_src_type[col] == ty;
                const float v = __shfl(dasv, mine ? T.rel_slot_src[col] : 0, 64);
                float w = 0.f;
                const int c = col - ld;
                if (is_dst && c >= 0 && c < T.n_rels && T.rel_dst_type[c] == ty) w = P.da_dst[zb + T.rel_slot_dst[c]];
                if (col < 2 * ld) P.da_src[(int64_t)u * 2 * ld + col] = (col < ld) ? (mine ? v : 0.f) : w;
            }
        }
    }
}


// Two source rows per wavefront, one per 32-lane half.  Most source rows are SNPs with one to three entries: a
// whole wavefront per row spends its (VALU-issue bound) instructions on two live lanes.  Here lane hl of a half
// owns entry hl of ITS row during the gather / per-slot phase and float4 hl of the row's 128-float dH during the
// accumulation; per-slot sums are 32-lane reductions (DPP + one permlane16 swap).  Taken when both rows are real, of
// the same node type and have at most KGW_PAIR_MAX entries each (processed in blocks of 32 per half): that covers the
// SNP rows (mean 2 entries) and 99 % of the gene / GO rows (mean ~35, which hold 70 % of all entries).
constexpr int KGW_PAIR_MAX = 128;
__device__ __forceinline__ bool bwd_src_row_pair(const LayerTab& T, const AggPtrs& P, int u, float* wdp) {
    const int lane = kgw_lane(), half = lane >> 5, hl = lane & 31, hb = half << 5;
    int ty = 0;
    while (ty + 1 < T.n_types && u >= T.type_src_base[ty + 1]) ++ty;
    if (u + 1 >= T.type_src_base[ty + 1]) return false;                  // the neighbour is of another type
    const int j0 = u - T.type_src_base[ty];
    const int n_real = P.meta->n_src[P.layer - 1][ty];
    if (j0 + 1 >= n_real) return false;                                   // padding involved: single-row path
    const int Rs = T.type_R_src[ty];
    if (Rs >= 32) return false;
    const int j = j0 + half, uh = u + half;
    const int tb = T.type_t_base[ty] + j * Rs;
    const int tpv = (hl <= Rs) ? P.t_ptr[tb + hl] : 0;
    const int p0 = __shfl(tpv, hb, 64), p1 = __shfl(tpv, hb + Rs, 64);
    const int n = p1 - p0;
    if (__ballot(n > KGW_PAIR_MAX)) return false;
    const int tpn = __shfl_down(tpv, 1, 64);
    const unsigned long long bal = __ballot(hl < Rs && tpn > tpv);
    const unsigned slots_any = (unsigned)(bal | (bal >> 32));              // slots used by either row
    const float4* dZ4 = (const float4*)P.dZ;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float dasv = 0.f;                                                      // lane hl == k of a half: d a_src of slot k
    const int nall = max(__shfl(n, 0, 64), __shfl(n, 32, 64));
    // the rows' entries in blocks of 32 per half (gene / GO rows average ~35 entries: two blocks).  The load chain
    // entries -> (alpha, d pre-activation) / dZ rows is unrolled across blocks: the NEXT block's entries and this block's
    // alphas are requested before this block's gathers, and the gathers (which need the Z rows only) before the alphas
    // are waited for.
    int te = 0, tz = 0;
    if (hl < min(n, 32)) {
        te = P.t_edge[p0 + hl];
        tz = P.t_zrow[p0 + hl];
    }
    for (int b0 = 0; b0 < nall; b0 += 32) {
        const int nb = min(max(n - b0, 0), 32);                            // entries of this half's row in the block
        float2 a2 = make_float2(0.f, 0.f);
        if (hl < nb) a2 = ((const float2*)P.adp)[te];
        int te_n = 0, tz_n = 0;
        if (hl < min(max(n - b0 - 32, 0), 32)) {
            te_n = P.t_edge[p0 + b0 + 32 + hl];
            tz_n = P.t_zrow[p0 + b0 + 32 + hl];
        }
        // dH row += sum over the block's entries of alpha * dZ[z row], eight (row tails: four) gathers in flight per half
        const int nmax = min(nall - b0, 32);
        int i0 = 0;
        for (; i0 + 4 < nmax; i0 += 8) {
            float4 x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = i0 + q;
                const int z = __shfl(tz, hb + (i < nb ? i : 0), 64);
                x[q] = dZ4[(int64_t)z * 32 + hl];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = i0 + q;
                const float a = __shfl(a2.x, hb + (i < nb ? i : 0), 64);
                fma4(acc, i < nb ? a : 0.f, x[q]);
            }
        }
        for (; i0 < nmax; i0 += 4) {
            float4 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q;
                const int z = __shfl(tz, hb + (i < nb ? i : 0), 64);
                x[q] = dZ4[(int64_t)z * 32 + hl];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q;
                const float a = __shfl(a2.x, hb + (i < nb ? i : 0), 64);
                fma4(acc, i < nb ? a : 0.f, x[q]);
            }
        }
        // per-slot sums of d pre-activation: the block's values go through the wavefront's 64 floats of LDS and lane k
        // of a half adds up slot k's range of ITS row, entries in ascending order -- all slots in parallel, as many steps
        // as the longest slot of the block has entries
        wdp[lane] = (hl < nb) ? a2.y : 0.f;
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (hl < Rs) {
            const int a = max(tpv - (p0 + b0), 0), b = min(tpn - (p0 + b0), nb);
            float sk = 0.f;
            for (int i = a; i < b; ++i) sk += wdp[hb + i];
            dasv += sk;
        }
        __asm__ volatile("" ::: "memory");
        te = te_n; tz = tz_n;
    }
    // d a_src flows back into h_src through a_s = <h_src, u_r>
    for (unsigned left = slots_any; left; left &= left - 1) {
        const int k = __builtin_ctz(left);
        const float dk = __shfl(dasv, hb + k, 64);
        const int r = T.rel_of_slot[ty][k];
        const float4 u4 = ((const float4*)(P.U + (int64_t)r * KGW_C))[hl];
        fma4(acc, dk, u4);
    }
    // destination side of the same node
    const int Rd = T.type_R_dst[ty];
    const bool is_dst = P.V && Rd > 0 && j < P.meta->n_rows[P.layer - 1][ty];
    const int zb = T.type_z_base[ty] + j * Rd;
    if (__ballot(is_dst)) {
        const float dav = (is_dst && hl < Rd) ? P.da_dst[zb + hl] : 0.f;
        for (int k = 0; k < Rd; ++k) {
            const float dk = __shfl(dav, hb + k, 64);
            if (__ballot(dk != 0.f)) {
                const int r = T.rel_of_dslot[ty][k];
                const float4 v4 = ((const float4*)(P.V + (int64_t)r * KGW_C))[hl];
                fma4(acc, dk, v4);
            }
        }
    }
    if (P.relu_in) {
        const float4 h4 = ((const float4*)(P.H + (int64_t)uh * KGW_C))[hl];
        acc.x = h4.x > 0.f ? acc.x : 0.f; acc.y = h4.y > 0.f ? acc.y : 0.f;
        acc.z = h4.z > 0.f ? acc.z : 0.f; acc.w = h4.w > 0.f ? acc.w : 0.f;
    }
    ((float4*)(P.dH + (int64_t)uh * KGW_C))[hl] = acc;
    // [d a_src | d a_dst] row, one column per relation id; each half writes its own row, two columns per lane
    if (P.da_src) {
        const int ld = T.ld_da;
        float* row = P.da_src + (int64_t)uh * 2 * ld;
        for (int c0 = 0; c0 < ld; c0 += 32) {                 // (more than 32 relations: a second round)
            const int col = c0 + hl;
            const bool mine = col < T.n_rels && T.rel_src_type[col] == ty;
            const float v = __shfl(dasv, hb + (mine ? T.rel_slot_src[col] : 0), 64);
            float w = 0.f;
            if (is_dst && col < T.n_rels && T.rel_dst_type[col] == ty) w = P.da_dst[zb + T.rel_slot_dst[col]];
            if (col < ld) { row[col] = mine ? v : 0.f; row[ld + col] = w; }
        }
    }
    return true;
}

// Eight consecutive source rows of one node type per wavefront, 8 lanes per row -- the rows of a type whose nodes have a
// handful of entries (the SNPs of the benchmark graph: ~2; two thirds of all rows).  The pair path above spends a
// wavefront-iteration -- row pointers, ballots, the per-slot LDS pass, ~340 VALU instructions -- on two such rows.  Here
// lane e of a row's group owns entry e (at most 8) and float4s e, e+8, e+16, e+24 of the 128-float row; the relation of
// an entry comes from the sampler (t_rel), so d a_src is "column rel(e) += d pre-activation(e)" and the term through
// a_src is sum_e dpre_e u_rel(e), entry by entry.  Which octets qualify is decided by the sampler (oct_flags: eight real rows
// of one short-row type, no destination row -- the seeds have a d a_dst term --, at most 8 entries each).
__device__ __forceinline__ void bwd_src_octet(const LayerTab& T, const AggPtrs& P, int u0) {
    const int lane = kgw_lane(), g = lane >> 3, gl = lane & 7, gb = g << 3;
    int ty = 0;
    while (ty + 1 < T.n_types && u0 >= T.type_src_base[ty + 1]) ++ty;
    const int Rs = T.type_R_src[ty];
    const int tb = T.type_t_base[ty] + (u0 - T.type_src_base[ty]) * Rs;
    const int pv = (lane <= 8) ? P.t_ptr[tb + lane * Rs] : 0;
    const int p0 = __shfl(pv, g, 64), n = __shfl(pv, g + 1, 64) - p0;
    const int u = u0 + g;
    const float4* dZ4 = (const float4*)P.dZ;
    const float4* U4 = (const float4*)P.U;
    int tz = 0, tr = 0;
    float2 a2 = make_float2(0.f, 0.f);
    if (gl < n) {
        const int te = P.t_edge[p0 + gl];
        tz = P.t_zrow[p0 + gl];
        tr = P.t_rel[p0 + gl];
        a2 = ((const float2*)P.adp)[te];
    }
    // the node's own row of the layer input only matters as a ReLU mask: requested first, folded into 16 bits as soon as it
    // lands (the gathers behind it stay in flight)
    unsigned mk = 0xffffu;
    float4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    bool first = true;
    float4 h4[4];
    if (P.relu_in) {
#pragma unroll
        for (int q = 0; q < 4; ++q) h4[q] = ((const float4*)(P.H + (int64_t)u * KGW_C))[gl + 8 * q];
    }
    // two entries per round: 8 gathers in flight per lane; the (L1-resident) u rows follow one entry at a time
    for (int i = 0; __ballot(i < n); i += 2) {
        float4 x[2][4];
        float al[2], dp[2];
        int rr[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool on = i + e < n;
            const int z = __shfl(tz, gb + (on ? i + e : 0), 64);
            rr[e] = __shfl(tr, gb + (on ? i + e : 0), 64);
            const float a = __shfl(a2.x, gb + (on ? i + e : 0), 64);
            const float d = __shfl(a2.y, gb + (on ? i + e : 0), 64);
            al[e] = on ? a : 0.f; dp[e] = on ? d : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) x[e][q] = dZ4[(int64_t)z * 32 + gl + 8 * q];
        }
        if (first && P.relu_in) {
            mk = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                mk |= ((h4[q].x > 0.f ? 1u : 0u) | (h4[q].y > 0.f ? 2u : 0u) | (h4[q].z > 0.f ? 4u : 0u) | (h4[q].w > 0.f ? 8u : 0u)) << (4 * q);
        }
        first = false;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float4 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = U4[rr[e] * 32 + gl + 8 * q];
#pragma unroll
            for (int q = 0; q < 4; ++q) { fma4(acc[q], al[e], x[e][q]); fma4(acc[q], dp[e], w[q]); }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 v = acc[q];
        const unsigned m4 = mk >> (4 * q);
        v.x = (m4 & 1u) ? v.x : 0.f; v.y = (m4 & 2u) ? v.y : 0.f;
        v.z = (m4 & 4u) ? v.z : 0.f; v.w = (m4 & 8u) ? v.w : 0.f;
        ((float4*)(P.dH + (int64_t)u * KGW_C))[gl + 8 * q] = v;
    }
    // [d a_src | d a_dst] row: column rel(e) of the first half gets d pre-activation(e) (entries of one relation are
    // adjacent and added in entry order, like the general path's slot sums); the d a_dst half is zero (no destination row)
    if (P.da_src) {
        const int n4 = T.ld_da >> 1;                              // float4s per row (2 * ld_da floats)
        float4* row = (float4*)(P.da_src + (int64_t)u * 2 * T.ld_da);
        for (int k0 = 0; k0 < n4; k0 += 8) {                      // (uniform trip count: the shuffles below need every lane)
            const int k = k0 + gl;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; __ballot(i < n); ++i) {
                const bool on = i < n;
                const int r = __shfl(tr, gb + (on ? i : 0), 64);
                const float d = __shfl(a2.y, gb + (on ? i : 0), 64);
                const int c = r - 4 * k;
                if (on && c >= 0 && c < 4) {
                    v[0] += c == 0 ? d : 0.f; v[1] += c == 1 ? d : 0.f;
                    v[2] += c == 2 ? d : 0.f; v[3] += c == 3 ? d : 0.f;
                }
            }
            if (k < n4) row[k] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

constexpr int KGW_DUV_SPLIT = 8;      // level-1 pieces per relation of the d u_r / d v_r sums

// Riders of the src-major launch: blocks [0, n_riders) of the grid, ahead of the row work (they start with the kernel and are
// done long before it ends).
//   [0, NR)                    rel_sums != NULL: relation r's sum of d a_dst over its destination rows (gradient of a
//                              per-relation logit constant, KgwLayerArgs.rel_sums)
//   then NR * 8 blocks         duv_ws != NULL: piece s of  d u_r = sum over the chunks c of relation r of part_du[c]
//                              (k_agg_bwd_dst left, per chunk, sum_e dpre_e h_src(e))
//   then NR * 8 blocks         piece s of  d v_r = sum_i d a_dst[i, r] h_dst[i]  over relation r's destination rows
// all in a fixed order; k_duv_fold adds the eight pieces.  This replaces the [d a_src | d a_dst] row per node that this pass
// used to write (39 MB) and the tall-skinny product over H that consumed it (two launches, 36 us).
__device__ __forceinline__ void bwd_src_rider(const LayerTab& T, const AggPtrs& P, int id, float* rel_sums, float* sm) {
    const int t = threadIdx.x;
    const int NR = T.n_rels;
    if (rel_sums) {
        if (id < NR) {
            const int r = id;
            float sacc = 0.f;
            if (T.live[r]) {
                const int rows = P.meta->n_rows[P.layer - 1][T.rel_dst_type[r]];
                const float* p = P.da_dst + T.z0[r];
                const int st = T.zstride[r];
                for (int i = t; i < rows; i += KGW_BLK) sacc += p[(int64_t)i * st];
            }
            sm[t] = sacc;
            __syncthreads();
            for (int o = KGW_BLK / 2; o > 0; o >>= 1) {
                if (t < o) sm[t] += sm[t + o];
                __syncthreads();
            }
            if (t == 0) rel_sums[r] = sm[0];
            return;
        }
        id -= NR;
    }
    const bool is_v = id >= NR * KGW_DUV_SPLIT;
    if (is_v) id -= NR * KGW_DUV_SPLIT;
    const int r = id / KGW_DUV_SPLIT, sp = id % KGW_DUV_SPLIT;
    const int col = t & (KGW_C - 1), rg = t >> 7;                 // 128 columns x 2 row groups
    // Eight independent partial sums per thread, eight loads in flight: the sums used to be ONE dependent load-add chain per
    // thread (215 iterations for the largest relation of the benchmark's layer 1 at ~240 ns each = 52 us -- hidden under the
    // row work, but the floor of any faster row pass).  Fixed order: partial q takes elements q, q + 8, ...
    // of the thread's list, the partials are added as ((0+1)+(2+3))+((4+5)+(6+7)).
    float pa[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (T.live[r]) {
        if (!is_v) {
            for (int h = 0; h < P.n_duv_hops; ++h) {
                const int c0 = P.seg_chptr[P.meta->seg_off[h][r]];
                const int c1 = (r + 1 < NR) ? P.seg_chptr[P.meta->seg_off[h][r + 1]] : P.meta->chunk_end[h];
                const int n = c1 - c0;
                const int a0 = c0 + (int)((int64_t)n * sp / KGW_DUV_SPLIT), a1 = c0 + (int)((int64_t)n * (sp + 1) / KGW_DUV_SPLIT);
                const float* p = P.part_du + col;
                int c = a0 + rg;
                for (; c + 14 < a1; c += 16) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = p[(int64_t)(c + 2 * q) * KGW_C];
#pragma unroll
                    for (int q = 0; q < 8; ++q) pa[q] += v[q];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (c + 2 * q < a1) pa[q] += p[(int64_t)(c + 2 * q) * KGW_C];
            }
        } else {
            const int rows = P.meta->n_rows[P.layer - 1][T.rel_dst_type[r]];
            const int a0 = (int)((int64_t)rows * sp / KGW_DUV_SPLIT), a1 = (int)((int64_t)rows * (sp + 1) / KGW_DUV_SPLIT);
            const float* dd = P.da_dst + T.z0[r];
            const float* hb = P.H + (int64_t)T.dst_hbase[r] * KGW_C + col;
            const int st = T.zstride[r];
            int i = a0 + rg;
            for (; i + 14 < a1; i += 16) {
                float d[8], hv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { d[q] = dd[(int64_t)(i + 2 * q) * st]; hv[q] = hb[(int64_t)(i + 2 * q) * KGW_C]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) pa[q] = fmaf(d[q], hv[q], pa[q]);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (i + 2 * q < a1) pa[q] = fmaf(dd[(int64_t)(i + 2 * q) * st], hb[(int64_t)(i + 2 * q) * KGW_C], pa[q]);
        }
    }
    const float acc = ((pa[0] + pa[1]) + (pa[2] + pa[3])) + ((pa[4] + pa[5]) + (pa[6] + pa[7]));
    sm[t] = acc;
    __syncthreads();
    if (rg == 0)
        P.duv_ws[((int64_t)(is_v ? NR : 0) + r) * KGW_DUV_SPLIT * KGW_C + (int64_t)sp * KGW_C + col] = sm[t] + sm[t + KGW_C];
}

// d u_r / d v_r from their eight pieces (zero rows for relations the layer does not compute)
__global__ void __launch_bounds__(KGW_C) k_duv_fold(int n_rels, const float* __restrict__ ws, float* __restrict__ dU,
                                                     float* __restrict__ dV) {
    const int r = blockIdx.x, v = blockIdx.y, c = threadIdx.x;
    const float* p = ws + ((int64_t)v * n_rels + r) * KGW_DUV_SPLIT * KGW_C + c;
    const float s = ((p[0] + p[KGW_C]) + (p[2 * KGW_C] + p[3 * KGW_C])) + ((p[4 * KGW_C] + p[5 * KGW_C]) + (p[6 * KGW_C] + p[7 * KGW_C]));
    (v ? dV : dU)[(int64_t)r * KGW_C + c] = s;
}

__global__ void __launch_bounds__(KGW_BLK) k_agg_bwd_src(LayerTab T, AggPtrs P, int n_src_rows, int main_blocks, int n_riders,
                                                         float* __restrict__ rel_sums) {
    __shared__ float s_dp[KGW_BLK];                           // 64 floats per wavefront (bwd_src_row_pair)
    if ((int)blockIdx.x < n_riders) {
        bwd_src_rider(T, P, (int)blockIdx.x, rel_sums, s_dp);
        return;
    }
    float* wdp = s_dp + (threadIdx.x & ~63);
    const int nw = main_blocks * 4;
    // A wavefront takes source rows two at a time, LAST rows first: the layout is type-major with the SNPs (short rows,
    // most of the rows) in front and the genes / GO terms (longer rows) at the end -- the long rows must start at the
    // beginning of the kernel, not in its last round.  Pairs inside an octet that the sampler flagged for the short path
    // are skipped here and taken by the second loop, whose octets are dealt from the LAST wavefront down: with about
    // one pair per wavefront the low-numbered wavefronts hold the gene / GO pairs, the high-numbered ones only skipped.
    // (Two plain loops on purpose: with the octet path inside the pair loop, or an inner loop over an unflagged octet's
    // pairs, the general path's code got worse -- scalar-register spills -- and the kernel slower than without octets.)
    const int npairs = (n_src_rows + 1) >> 1;
    const int w0 = ((int)blockIdx.x - n_riders) * 4 + (threadIdx.x >> 6);
    for (int i0 = w0; i0 < npairs; i0 += nw) {
        const int u = __builtin_amdgcn_readfirstlane(2 * (npairs - 1 - i0));
        if (u < T.oct_rows && P.oct_flags[u >> 3]) continue;
        if (u + 1 < n_src_rows && bwd_src_row_pair(T, P, u, wdp)) continue;
        bwd_src_one_row(T, P, u);
        if (u + 1 < n_src_rows) bwd_src_one_row(T, P, u + 1);
    }
    // (Round 5, measured and dropped: every XCD taking a CONTIGUOUS eighth of the octets -- neighbouring SNPs gather the dZ rows
    //  of the same few genes, so an XCD's private L2 would hold its share instead of every L2 seeing all 11.8 MB of dZ.  Same
    //  values, 104.0 / 104.5 us round-robin against 105.9 / 106.2 with ranges on one box: the short rows are not what waits on
    //  the fabric.)
    for (int o = nw - 1 - w0; o < (T.oct_rows >> 3); o += nw)
        if (P.oct_flags[o]) bwd_src_octet(T, P, 8 * o);
}

// alpha per local edge (attention export)
__global__ void __launch_bounds__(KGW_BLK) k_edge_alpha(LayerTab T, AggPtrs P, float inv_temp, float* out) {
    const int lane = kgw_lane();
    const int nw = gridDim.x * 4;
    const int n_chunks = P.meta->n_chunks[P.layer - 1];
    for (int c = blockIdx.x * 4 + (threadIdx.x >> 6); c < n_chunks; c += nw) {
        const KgwChunk ck = load_chunk(P.chunks, c);
        const int r = ck.rel;
        const int n = ck.e1 - ck.e0;
        if (!T.live[r]) { for (int t = lane; t < n; t += 64) out[ck.e0 + t] = 0.f; continue; }
        const int zrow = T.z0[r] + ck.row * T.zstride[r];
        const float M = P.stat[2 * (int64_t)zrow];
        const float inv_den = 1.0f / P.stat[2 * (int64_t)zrow + 1];
        for (int t = lane; t < n; t += 64)
            out[ck.e0 + t] = __expf(P.e_edge[ck.e0 + t] * inv_temp - M) * inv_den;
    }
}

// ---- host side -----------------------------------------------------------------------------------
int build_tab(const KgwLayerArgs* a, LayerTab* T) {
    const KgwGraph* G = a->graph_host;
    const KgwBatchMeta* M = a->meta_host;
    if (!G || !M) return KGW_E_NULL;
    const int l = a->layer;
    if (l < 1 || l > G->n_layers) return KGW_E_RANGE;
    if (G->n_rels > KGW_MAX_RELS || G->n_types > KGW_MAX_TYPES) return KGW_E_RANGE;
    T->n_rels = G->n_rels;
    T->n_types = G->n_types;
    T->partial = a->partial_rels;
    T->ld_da = (G->n_rels + 3) & ~3;
    for (int r = 0; r < G->n_rels; ++r) {
        const int s = G->rel_src[r], d = G->rel_dst[r];
        T->src_base[r] = M->src_base[l - 1][s];
        T->z0[r] = M->z_base[l - 1][d] + G->rel_slot_dst[r];
        T->zstride[r] = G->R_dst[d];
        T->live[r] = G->rel_live[l - 1][r];
        if (G->rel_slot_src[r] >= KGW_MAX_RELS / 2) return KGW_E_RANGE;
        if (G->rel_slot_dst[r] >= KGW_MAX_RELS / 2) return KGW_E_RANGE;
        T->rel_of_slot[s][G->rel_slot_src[r]] = (int8_t)r;
        T->rel_src_type[r] = (int8_t)s;
        T->rel_slot_src[r] = (int8_t)G->rel_slot_src[r];
        T->rel_dst_type[r] = (int8_t)d;
        T->rel_slot_dst[r] = (int8_t)G->rel_slot_dst[r];
        T->rel_of_dslot[d][G->rel_slot_dst[r]] = (int8_t)r;
        T->dst_hbase[r] = M->src_base[l - 1][d];
    }
    for (int t = 0; t <= G->n_types; ++t) {
        T->type_src_base[t] = M->src_base[l - 1][t];
        T->type_t_base[t] = M->t_base[l - 1][t];
        if (t < G->n_types) {
            T->type_R_src[t] = G->R_src[t];
            T->type_R_dst[t] = G->R_dst[t];
            T->type_z_base[t] = M->z_base[l - 1][t];
        }
    }
    // rows [0, oct_rows): the leading node types that carry the short-row hint -- their octets are looked up in oct_flags
    {
        int t = 0;
        while (t < G->n_types && ((G->short_types >> t) & 1u)) ++t;
        T->oct_rows = (a->t_rel && a->oct_flags) ? (T->type_src_base[t] & ~7) : 0;
    }
    return KGW_OK;
}

AggPtrs build_ptrs(const KgwLayerArgs* a) {
    AggPtrs P;
    P.chunks = a->chunks; P.col_local = a->col_local; P.H = a->H; P.a_dst = a->a_dst; P.V = a->V; P.U = a->U; P.raw = (a->flags & KGW_F_RAW_WEIGHTS) ? 1 : 0; P.relu_in = (a->flags & KGW_F_RELU_INPUT) ? 1 : 0;
    P.Z = a->Z; P.stat = a->stat; P.e_edge = a->e_edge; P.part = a->part; P.dZ = a->dZ; P.adp = a->adp;
    P.da_dst = a->da_dst; P.part_da = a->part_da; P.t_ptr = a->t_ptr; P.t_edge = a->t_edge;
    P.t_zrow = a->t_zrow; P.t_rel = a->t_rel; P.oct_flags = a->oct_flags; P.part_du = a->part_du; P.seg_chptr = a->seg_chptr; P.duv_ws = a->duv_ws;
    P.n_duv_hops = a->n_multi_hops; P.dH = a->dH; P.da_src = a->da_src; P.multi = a->multi; P.multi_cap = a->multi_cap;
    P.meta = a->meta_dev; P.layer = a->layer;
    P.lbias = a->logit_bias;
    return P;
}

inline int grid_for_waves(int64_t n_waves) {
    int64_t g = (n_waves + 3) / 4;
    if (g > KGW_GRID) g = KGW_GRID;
    if (g < 1) g = 1;
    return (int)g;
}

// Grid of the three main aggregate kernels: up to 8192 blocks (32 k wavefronts), i.e. about one work item (chunk /
// source-row pair) per wavefront.  Chunks differ in size by two orders of magnitude; with the 2048-block grid-stride
// launch every wavefront drew ~2 of them and the longest draws set the kernel's tail: measured 74 -> 64 us (k_agg_fwd),
// 72 -> 65 us (k_agg_bwd_dst), 163 -> 154 us (k_agg_bwd_src).  (32768 blocks: k_agg_bwd_src back to 164 us -- the
// per-wavefront prologue shows; exactly one resident round -- occupancy x CUs blocks -- : k_agg_fwd 78 us.)
// Round 4, re-measured with today's kernels (cap 8192 / 12288 / 16384, layer 1): batch 512 flat (forward 56.0 / 56.3 / 56.0 us,
// backward-dst 57.1 / 58.0 / 57.6, backward-src 108.4 / 106.8 / 109.3 -- only the src-major pass reaches the cap there), batch 4096
// (6.2 M edges, 15 k blocks of chunks) 329.6 / 308.1 / 298.7, 343.5 / 326.5 / 325.3, 566.0 / 538.8 / 535.8: 16384.
inline int grid_fine(int64_t n_items) {
    const int64_t cap = 16384;
    int64_t g = (n_items + 3) / 4;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)((g + 7) & ~7ll);             // whole rounds over the 8 XCDs: a grid-stride step keeps a block's XCD class
}

}  // namespace

extern "C" int kgw_gat_aggregate_fwd(const KgwLayerArgs* a, kgw_stream_t stream_) {
    if (!a) return KGW_E_NULL;
    if (a->n_chunks == 0) return KGW_OK;
    if (!a->chunks || !a->col_local || !a->H || (!a->a_dst && !a->V) || !a->U || !a->Z || !a->stat || !a->e_edge || !a->part ||
        !a->meta_dev)
        return KGW_E_NULL;
    LayerTab T;
    int rc = build_tab(a, &T);
    if (rc) return rc;
    AggPtrs P = build_ptrs(a);
    hipStream_t st = (hipStream_t)stream_;
    if (a->ev_before) KGW_HIP(hipEventRecord((hipEvent_t)a->ev_before, st));
    // (software-pipelined chunks -- next group's rows + next block's ids in flight -- measured at KGW_CHUNK = 128 on the layer-1
    //  launch of the benchmark: forward 55.5 us plain / 57.5 pipelined (122 VGPRs cost two wavefronts per SIMD), backward-dst
    //  56.8 plain / 53.2 pipelined => the forward plain, the dst-major backward pipelined)
    if (P.raw) k_agg_fwd<true, false><<<grid_fine(a->n_chunks), KGW_BLK, 0, st>>>(T, P, a->neg_slope, a->inv_temp);
    else       k_agg_fwd<false, false><<<grid_fine(a->n_chunks), KGW_BLK, 0, st>>>(T, P, a->neg_slope, a->inv_temp);
    KGW_LAUNCH_CHECK();
    if (a->ev_after) KGW_HIP(hipEventRecord((hipEvent_t)a->ev_after, st));
    if (a->multi && a->multi_cap > 0) {     // hub rows: the number of multi-chunk segments is read on the device
        k_agg_fwd_combine<<<grid_for_waves(a->multi_cap < 1024 ? a->multi_cap : 1024), KGW_BLK, 0, st>>>(T, P, a->n_multi_hops);
        KGW_LAUNCH_CHECK();
    }
    return KGW_OK;
}

extern "C" int kgw_gat_aggregate_bwd_dst(const KgwLayerArgs* a, kgw_stream_t stream_) {
    if (!a) return KGW_E_NULL;
    if (a->n_chunks == 0) return KGW_OK;
    if (!a->chunks || !a->col_local || !a->H || !a->Z || !a->stat || !a->e_edge || !a->dZ || !a->adp ||
        !a->da_dst || !a->part_da || !a->meta_dev)
        return KGW_E_NULL;
    LayerTab T;
    int rc = build_tab(a, &T);
    if (rc) return rc;
    AggPtrs P = build_ptrs(a);
    hipStream_t st = (hipStream_t)stream_;
    if (a->ev_before) KGW_HIP(hipEventRecord((hipEvent_t)a->ev_before, st));
    k_agg_bwd_dst<true><<<grid_fine(a->n_chunks), KGW_BLK, 0, st>>>(T, P, a->neg_slope, a->inv_temp);
    KGW_LAUNCH_CHECK();
    if (a->ev_after) KGW_HIP(hipEventRecord((hipEvent_t)a->ev_after, st));
    if (a->multi && a->multi_cap > 0) {
        k_agg_bwd_combine<<<grid_for_waves(a->multi_cap < 1024 ? a->multi_cap : 1024), KGW_BLK, 0, st>>>(T, P, a->n_multi_hops);
        KGW_LAUNCH_CHECK();
    }
    return KGW_OK;
}

extern "C" int kgw_gat_aggregate_bwd_src(const KgwLayerArgs* a, kgw_stream_t stream_) {
    if (!a) return KGW_E_NULL;
    if (a->n_src_rows == 0) return KGW_OK;
    if (!a->dZ || !a->adp || !a->t_ptr || !a->t_edge || !a->t_zrow || !a->dH || (!a->da_src && !a->dU) || !a->U || !a->meta_dev)
        return KGW_E_NULL;
    LayerTab T;
    int rc = build_tab(a, &T);
    if (rc) return rc;
    AggPtrs P = build_ptrs(a);
    if (a->ev_before) KGW_HIP(hipEventRecord((hipEvent_t)a->ev_before, (hipStream_t)stream_));
    const int gmain = grid_fine((a->n_src_rows + 1) / 2);
    const bool duv = a->dU && a->dV && a->part_du && a->duv_ws && a->seg_chptr;
    const int n_riders = (a->rel_sums ? T.n_rels : 0) + (duv ? 2 * KGW_DUV_SPLIT * T.n_rels : 0);
    if (!duv) P.duv_ws = nullptr;
    k_agg_bwd_src<<<gmain + n_riders, KGW_BLK, 0, (hipStream_t)stream_>>>(T, P, a->n_src_rows, gmain, n_riders, a->rel_sums);
    KGW_LAUNCH_CHECK();
    if (a->ev_after) KGW_HIP(hipEventRecord((hipEvent_t)a->ev_after, (hipStream_t)stream_));
    if (duv && !(a->flags & KGW_F_DUV_PIECES)) {        // (pieces: the consumer of d u_r / d v_r adds the eight pieces itself)
        k_duv_fold<<<dim3(T.n_rels, 2), KGW_C, 0, (hipStream_t)stream_>>>(T.n_rels, a->duv_ws, a->dU, a->dV);
        KGW_LAUNCH_CHECK();
    }
    return KGW_OK;
}

extern "C" int kgw_edge_alpha(const KgwLayerArgs* a, float* alpha_out, kgw_stream_t stream_) {
    if (!a || !alpha_out) return KGW_E_NULL;
    if (a->n_chunks == 0) return KGW_OK;
    if (!a->chunks || !a->stat || !a->e_edge || !a->meta_dev) return KGW_E_NULL;
    LayerTab T;
    int rc = build_tab(a, &T);
    if (rc) return rc;
    AggPtrs P = build_ptrs(a);
    k_edge_alpha<<<grid_for_waves(a->n_chunks), KGW_BLK, 0, (hipStream_t)stream_>>>(T, P, a->inv_temp, alpha_out);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}


// ---- per-relation sums of a per-segment quantity (d logit constant = column sums of d a_dst) ----------------------
namespace {
__global__ void __launch_bounds__(256) k_relation_sums(LayerTab T, const KgwBatchMeta* __restrict__ meta, int layer,
                                                       const float* __restrict__ x, float* __restrict__ out) {
    __shared__ float sm[256];
    const int r = blockIdx.x, t = threadIdx.x;
    float s = 0.f;
    if (T.live[r]) {
        const int ty = T.rel_dst_type[r];
        const int rows = meta->n_rows[layer - 1][ty];
        const float* p = x + T.z0[r];
        const int st = T.zstride[r];
        for (int i = t; i < rows; i += 256) s += p[(int64_t)i * st];
    }
    sm[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {            // fixed tree: deterministic
        if (t < o) sm[t] += sm[t + o];
        __syncthreads();
    }
    if (t == 0) out[r] = sm[0];
}
}  // namespace

extern "C" int kgw_relation_sums(const KgwLayerArgs* a, const float* x, float* out, kgw_stream_t stream_) {
    if (!a || !x || !out || !a->meta_dev) return KGW_E_NULL;
    LayerTab T;
    int rc = build_tab(a, &T);
    if (rc) return rc;
    k_relation_sums<<<T.n_rels, 256, 0, (hipStream_t)stream_>>>(T, a->meta_dev, a->layer, x, out);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

// ---- SNP-sharded mode: merge of partial softmax states across ranks --------------------------------------------
namespace {
// one half-wave (32 lanes x float4) per segment
__global__ void __launch_bounds__(KGW_BLK) k_softmax_pack(const float* __restrict__ Z, const float* __restrict__ stat,
                                                          const int32_t* __restrict__ seg_zrow, int64_t n_seg,
                                                          float* __restrict__ parts) {
    const int hl = threadIdx.x & 31;
    for (int64_t x = (int64_t)blockIdx.x * (KGW_BLK / 32) + (threadIdx.x >> 5); x < n_seg; x += (int64_t)gridDim.x * (KGW_BLK / 32)) {
        const int64_t z = seg_zrow[x];
        float* pr = parts + x * PART_STRIDE;
        ((float4*)(pr + 4))[hl] = ((const float4*)(Z + z * KGW_C))[hl];
        if (hl == 0) { pr[0] = stat[2 * z]; pr[1] = stat[2 * z + 1]; pr[2] = 0.f; pr[3] = 0.f; }
    }
}

__global__ void __launch_bounds__(KGW_BLK) k_softmax_merge(const float* __restrict__ parts, int n_ranks,
                                                           const int32_t* __restrict__ seg_zrow, int64_t n_seg,
                                                           float* __restrict__ Z, float* __restrict__ stat) {
    const int hl = threadIdx.x & 31;
    for (int64_t x = (int64_t)blockIdx.x * (KGW_BLK / 32) + (threadIdx.x >> 5); x < n_seg; x += (int64_t)gridDim.x * (KGW_BLK / 32)) {
        float M = NEG_BIG;
        for (int p = 0; p < n_ranks; ++p) {
            const float* pr = parts + ((int64_t)p * n_seg + x) * PART_STRIDE;
            if (pr[1] > 0.f) M = fmaxf(M, pr[0]);               // ranks without an edge of this segment left (0, 0)
        }
        float S = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = 0; p < n_ranks; ++p) {                     // fixed rank order: the same bits on every rank
            const float* pr = parts + ((int64_t)p * n_seg + x) * PART_STRIDE;
            const float s = pr[1];
            if (!(s > 0.f)) continue;
            const float f = __expf(pr[0] - M);
            S = fmaf(s, f, S);
            fma4(acc, f, ((const float4*)(pr + 4))[hl]);
        }
        const int64_t z = seg_zrow[x];
        const bool any = S > 0.f;
        const float den = any ? S + 1e-16f : 0.f, inv = any ? 1.0f / den : 0.f;
        scale4(acc, inv);
        ((float4*)(Z + z * KGW_C))[hl] = acc;
        if (hl == 0) { stat[2 * z] = any ? M : 0.f; stat[2 * z + 1] = den; }
    }
}

template <int VEC>
__global__ void __launch_bounds__(KGW_BLK) k_scatter_rows(const float* __restrict__ src, const int32_t* __restrict__ ids,
                                                          int64_t total, int wv, float* __restrict__ dst) {
    for (int64_t e = (int64_t)blockIdx.x * KGW_BLK + threadIdx.x; e < total; e += (int64_t)gridDim.x * KGW_BLK) {
        const int64_t r = e / wv;
        const int c = (int)(e - r * wv);
        const int64_t d = (int64_t)ids[r] * wv + c;
        if (VEC == 4) ((float4*)dst)[d] = ((const float4*)src)[e];
        else dst[d] = src[e];
    }
}
}  // namespace

extern "C" int kgw_softmax_pack(const float* Z, const float* stat, const int32_t* seg_zrow, int64_t n_seg, float* parts,
                                kgw_stream_t stream_) {
    if (n_seg == 0) return KGW_OK;
    if (!Z || !stat || !seg_zrow || !parts) return KGW_E_NULL;
    if (n_seg < 0) return KGW_E_RANGE;
    int64_t g = (n_seg + KGW_BLK / 32 - 1) / (KGW_BLK / 32);
    if (g > KGW_GRID) g = KGW_GRID;
    k_softmax_pack<<<(int)g, KGW_BLK, 0, (hipStream_t)stream_>>>(Z, stat, seg_zrow, n_seg, parts);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_softmax_merge(const float* parts, int32_t n_ranks, const int32_t* seg_zrow, int64_t n_seg, float* Z,
                                 float* stat, kgw_stream_t stream_) {
    if (n_seg == 0) return KGW_OK;
    if (!parts || !seg_zrow || !Z || !stat) return KGW_E_NULL;
    if (n_seg < 0 || n_ranks < 1) return KGW_E_RANGE;
    int64_t g = (n_seg + KGW_BLK / 32 - 1) / (KGW_BLK / 32);
    if (g > KGW_GRID) g = KGW_GRID;
    k_softmax_merge<<<(int)g, KGW_BLK, 0, (hipStream_t)stream_>>>(parts, n_ranks, seg_zrow, n_seg, Z, stat);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_scatter_rows(const float* src, const int32_t* ids, int64_t n_rows, int32_t width, float* dst,
                                kgw_stream_t stream_) {
    if (n_rows == 0) return KGW_OK;
    if (!src || !ids || !dst) return KGW_E_NULL;
    if (width <= 0 || n_rows < 0) return KGW_E_RANGE;
    const bool v4 = (width & 3) == 0 && !(((uintptr_t)src | (uintptr_t)dst) & 15);
    const int wv = v4 ? width >> 2 : width;
    const int64_t total = n_rows * wv;
    int64_t g = (total + KGW_BLK - 1) / KGW_BLK;
    if (g > KGW_GRID * 8) g = KGW_GRID * 8;
    if (v4) k_scatter_rows<4><<<(int)g, KGW_BLK, 0, (hipStream_t)stream_>>>(src, ids, total, wv, dst);
    else k_scatter_rows<1><<<(int)g, KGW_BLK, 0, (hipStream_t)stream_>>>(src, ids, total, wv, dst);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

// ---- misc entry points -----------------------------------------------------------------------------
extern "C" int kgw_version(void) { return KGW_VERSION; }

extern "C" const char* kgw_status_string(int status) {
    switch (status) {
        case KGW_OK: return "ok";
        case KGW_E_NULL: return "null pointer argument";
        case KGW_E_RANGE: return "argument out of range";
        case KGW_E_UNSUPPORTED: return "unsupported configuration";
        default: return status > 0 ? hipGetErrorString((hipError_t)status) : "unknown error";
    }
}

extern "C" int kgw_struct_sizes(int64_t* out, int n) {
    if (!out) return KGW_E_NULL;
    const int64_t v[7] = {(int64_t)sizeof(KgwGraph), (int64_t)sizeof(KgwBatchMeta), (int64_t)sizeof(KgwChunk),
                          (int64_t)sizeof(KgwBatchBuf), (int64_t)sizeof(KgwLayerArgs), (int64_t)sizeof(KgwTnJob),
                          (int64_t)sizeof(KgwGradSrc)};
    for (int i = 0; i < n && i < 7; ++i) out[i] = v[i];
    return KGW_OK;
}

// ---- self-test of the cross-lane primitives (tests/test_gpu_primitives.py) -------------------------
namespace {
__global__ void k_debug_reduce(const float* in, float* out_half, float* out_wave, float* out_steps) {
    const int lane = threadIdx.x;
    float v = in[lane];
    out_half[lane] = kgw_half_allsum(v);
    out_wave[lane] = kgw_wave_allsum(v);
    float a = v + kgw_dpp<0xB1>(v);
    out_steps[lane] = a;
    float b = a + kgw_dpp<0x4E>(a);
    out_steps[64 + lane] = b;
    float c = b + kgw_dpp<0x141>(b);
    out_steps[128 + lane] = c;
    float d = c + kgw_dpp<0x140>(c);
    out_steps[192 + lane] = d;
    // raw semantics of the two swap instructions on distinguishable inputs
    int a16 = lane, b16 = 100 + lane;
    auto r16 = __builtin_amdgcn_permlane16_swap(a16, b16, false, false);
    out_steps[256 + lane] = (float)r16[0];
    out_steps[320 + lane] = (float)r16[1];
    int a32 = lane, b32 = 100 + lane;
    auto r32 = __builtin_amdgcn_permlane32_swap(a32, b32, false, false);
    out_steps[384 + lane] = (float)r32[0];
    out_steps[448 + lane] = (float)r32[1];
}
}  // namespace

namespace {
__global__ void k_debug_reduce8(const float* in, float* out) {
    const int lane = threadIdx.x, hl = lane & 31;
    float v[8];
    for (int p = 0; p < 8; ++p) v[p] = in[lane * 8 + p];
    out[lane] = kgw_half_reduce8(v, hl);              // lane with (hl & 7) == p: sum over its half of v[p]
    out[64 + lane] = kgw_max8(in[lane * 8]);          // over the lanes of the 8-lane group
    out[128 + lane] = kgw_sum8(in[lane * 8]);
    out[192 + lane] = kgw_bcast8<3>(in[lane * 8]);    // value of lane (lane & ~7) | 3
    out[256 + lane] = kgw_xor4(in[lane * 8]);
    out[320 + lane] = kgw_xor8(in[lane * 8]);
}
}  // namespace

extern "C" int kgw_debug_reduce8(const float* in, float* out, kgw_stream_t stream_) {
    if (!in || !out) return KGW_E_NULL;
    k_debug_reduce8<<<1, 64, 0, (hipStream_t)stream_>>>(in, out);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_debug_reduce(const float* in, float* out_half, float* out_wave, float* out_steps,
                                kgw_stream_t stream_) {
    if (!in || !out_half || !out_wave || !out_steps) return KGW_E_NULL;
    k_debug_reduce<<<1, 64, 0, (hipStream_t)stream_>>>(in, out_half, out_wave, out_steps);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}
