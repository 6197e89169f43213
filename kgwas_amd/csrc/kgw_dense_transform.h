// kgw_dense_transform.h -- part of kgw_dense.hip (ONE translation unit, split by kernel family in round 6; include order matters:
// later families use device functions of earlier ones): the per-relation transform on few rows (kgw_linear_splitk*), its backward as one launch (kgw_transform_bwd) and the d gamma column sums.
#pragma once

// ======================================================================================================
// kgw_linear_splitk: Y[rows, N] = act(X[rows, K] * Wop + bias) for FEW rows (hundreds to a few thousand) when one of
// K, N is 128 and the other a multiple of 128 -- the per-relation transform of a layer after aggregate-then-transform
// (kgwas/conv.py:138-144 + bias :190 + HeteroConv sum model.py:74 + ReLU :75 as ONE product [N_dst, R*128] x [R*128, 128])
// and its dZ twin [N_dst, 128] x [128, R*128], at the shapes a 512-seed batch has: ~1.2 k gene rows x R = 17 relations,
// 512 SNP rows x R = 6.  A 128-row-tile kernel puts such a product on ten workgroups.
//
// Here the long dimension is cut into 128-wide SLABS (= relations) and a 4-wavefront block owns (slab, a strided group of
// 32-row tiles).  The slab's 128 x 128 weight block is STATIONARY IN REGISTERS: wavefront w holds, as MFMA B operands, the
// 64 values W(k = 64 lk + j, column 32 w + li) of its lanes for the whole block (v_mfma_f32_32x32x2_f32, k order inside
// the slab permuted so that a lane's A values are contiguous), loaded once -- coalesced for the [K, N] form (the packed
// per-relation weights).  Row tiles stream through a double-buffered LDS tile (coalesced 512-byte row reads, row stride
// 132 floats: the ds_read_b128 of the A operand is conflict free), one barrier per tile, 64 MFMAs per wavefront and tile.
//   K > 128 (forward transform): slab = K range; a block writes its partial [rows, 128] to the workspace and a second
//     launch adds the slabs in order (deterministic, no atomics) with bias / ReLU;
//   K == 128 (dZ twin): slab = column range; results are final, written directly.
// ======================================================================================================
namespace {

constexpr int SK_LD = 132;      // LDS row stride of the X tile (floats)

struct SplitKArgs {
    const float* X; int64_t ldx;
    const float* W; int64_t ldw;
    const float* bias;
    float* Y; int64_t ldy;
    float* ws;                 // [KS][rows][128] partial products (KS > 1)
    int64_t rows; int K, N;
    int relu, w_kn;
    int RT, KS, NS, G;         // row tiles; K slabs; column slabs; row-tile groups per slab
    const int32_t* rows_dev;
    const float* seg_stat;     // optional: (max, denominator) pairs of the KS segments of every row; with gamma [KS][128]
    const float* gamma;
};

__device__ __forceinline__ int64_t sk_rows_eff(const SplitKArgs& a) {
    if (!a.rows_dev) return a.rows;
    const int64_t r = *a.rows_dev;
    return r < 0 ? 0 : (r < a.rows ? r : a.rows);
}

// Up to four products of one kind per launch (kgw_linear_splitk_multi): the grid's x dimension is the concatenation of the jobs'
// blocks; a single product is a table of one.
constexpr int SK_MAX_JOBS = 4;
struct SplitKJobs { SplitKArgs j[SK_MAX_JOBS]; int blk0[SK_MAX_JOBS + 1]; int n; };
struct ColsumJobs { const float* seg_stat[SK_MAX_JOBS]; const float* dY[SK_MAX_JOBS]; float* dgamma[SK_MAX_JOBS];
                    int64_t ldy[SK_MAX_JOBS], rows[SK_MAX_JOBS]; int R[SK_MAX_JOBS]; int blk0[SK_MAX_JOBS + 1]; int n; };

// (the body of k_linear_splitk for block ``bxg`` of the jobs' concatenated grid; Xs: 2 x 32 x SK_LD floats of LDS; also inlined
//  into k_transform_bwd)
template <bool WKN>
__device__ __forceinline__ void splitk_block(const SplitKJobs& J, const int bxg, float (*Xs)[32 * SK_LD]) {
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lk = lane >> 5, w = tid >> 6;
    int jq = 0;
    while (jq + 1 < J.n && bxg >= J.blk0[jq + 1]) ++jq;
    const SplitKArgs& a = J.j[jq];
    const bool ksplit = a.KS > 1;
    const int nslab_ = ksplit ? a.KS : a.NS;
    const int bx = bxg - J.blk0[jq];
    const int slab = bx % nslab_, g = bx / nslab_;
    const int64_t rows_eff = sk_rows_eff(a);
    const int kx0 = ksplit ? slab * 128 : 0;           // first K column of the X tiles
    const int n0 = (ksplit ? 0 : slab * 128) + 32 * w; // first output column of this wavefront
    const int ntile = (int)((rows_eff + 31) / 32);
    if (g >= ntile) return;

    // B operand, stationary: W(k = kx0 + 64 lk + j, n = n0 + li)
    float bw[64];
    if (WKN) {
        const float* p = a.W + (int64_t)(kx0 + 64 * lk) * a.ldw + n0 + li;
#pragma unroll
        for (int j = 0; j < 64; ++j) bw[j] = p[(int64_t)j * a.ldw];
    } else {
        const float4* p = (const float4*)(a.W + (int64_t)(n0 + li) * a.ldw + kx0 + 64 * lk);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 v = p[q];
            bw[4 * q] = v.x; bw[4 * q + 1] = v.y; bw[4 * q + 2] = v.z; bw[4 * q + 3] = v.w;
        }
    }
    // X tile of row tile rt: thread t moves float4 (row = idx / 32, column 4 (idx % 32)), idx = t + 256 i
    float4 xr[4];
    auto fetch = [&](int rt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int64_t r = (int64_t)rt * 32 + (idx >> 5);
            const bool ok = r < rows_eff;
            const float4 v = *(const float4*)(a.X + (ok ? r : 0) * a.ldx + kx0 + 4 * (idx & 31));
            xr[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            *(float4*)(&Xs[buf][(idx >> 5) * SK_LD + 4 * (idx & 31)]) = xr[i];
        }
    };
    fetch(g);
    stage(0);
    __syncthreads();
    const float bv = (!ksplit && a.bias) ? a.bias[n0 + li] : 0.f;
    int buf = 0;
    for (int rt = g; rt < ntile; rt += a.G, buf ^= 1) {
        const bool more = rt + a.G < ntile;
        if (more) fetch(rt + a.G);                         // in flight under the MFMAs below
        float xa[64];
        const float4* px = (const float4*)(&Xs[buf][li * SK_LD + 64 * lk]);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 v = px[q];
            xa[4 * q] = v.x; xa[4 * q + 1] = v.y; xa[4 * q + 2] = v.z; xa[4 * q + 3] = v.w;
        }
        // two interleaved accumulators (even / odd k steps): no back-to-back dependent MFMAs
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
        for (int j = 0; j < 64; j += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j], bw[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j + 1], bw[j + 1], acc1, 0, 0, 0);
        }
        // accumulator element e of a lane: row (e & 3) + 8 (e >> 2) + 4 lk, column li
        const int64_t r0 = (int64_t)rt * 32;
        if (ksplit) {
            float* out = a.ws + ((int64_t)slab * a.rows + r0) * 128 + n0 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (r0 + row < rows_eff) out[(int64_t)row * 128] = acc0[e] + acc1[e];
            }
        } else {
            float* out = a.Y + r0 * a.ldy + n0 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (r0 + row >= rows_eff) continue;
                float v = acc0[e] + acc1[e] + bv;
                if (a.relu) v = fmaxf(v, 0.f);
                out[(int64_t)row * a.ldy] = v;
            }
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
    }
}

template <bool WKN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_linear_splitk(SplitKJobs J) {
    __shared__ __attribute__((aligned(16))) float Xs[2][32 * SK_LD];
    splitk_block<WKN>(J, (int)blockIdx.x, Xs);
}

// Forward transform in ONE launch (K = R * 128 > 128, N == 128, packed [K, N] weights): a block of EIGHT wavefronts owns a
// (32-row, 32-column) output tile; wavefront w multiplies the K slabs (= relations) w, w + 8, ... into its own accumulator --
// the slab's X tile goes through a wavefront-private LDS buffer (coalesced 512-byte row reads, then the per-row operand
// reads of the MFMA layout; no block barrier: a wavefront's LDS operations execute in order), the weights come straight
// from global memory (coalesced) -- and the eight accumulators are added through LDS in wavefront order, with bias, the
// per-segment constants of a folded FC_output and ReLU applied on the way out.  No partial buffer, no second launch.
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_linear_splitk_fused(SplitKJobs J) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lk = lane >> 5, w = tid >> 6;
    int jq = 0;
    while (jq + 1 < J.n && (int)blockIdx.x >= J.blk0[jq + 1]) ++jq;
    const SplitKArgs& a = J.j[jq];
    const int bx = (int)blockIdx.x - J.blk0[jq];
    const int rt = bx % a.RT, cb = bx / a.RT;
    const int64_t rows_eff = sk_rows_eff(a);
    const int64_t r0 = (int64_t)rt * 32;
    float* my = lds + w * (32 * SK_LD);
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    if (r0 < rows_eff) {
        for (int ks = w; ks < a.KS; ks += 8) {
            float4 xr[16];
            float bw[64];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t r = r0 + 2 * q + lk;
                const bool ok = r < rows_eff;
                const float4 v = *(const float4*)(a.X + (ok ? r : r0) * a.ldx + ks * 128 + 4 * li);
                xr[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const float* pw = a.W + (int64_t)(ks * 128 + 64 * lk) * a.ldw + cb * 32 + li;
#pragma unroll
            for (int j = 0; j < 64; ++j) bw[j] = pw[(int64_t)j * a.ldw];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 16; ++q) *(float4*)(&my[(2 * q + lk) * SK_LD + 4 * li]) = xr[q];
            float xa[64];
            const float4* px = (const float4*)(&my[li * SK_LD + 64 * lk]);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float4 v = px[q];
                xa[4 * q] = v.x; xa[4 * q + 1] = v.y; xa[4 * q + 2] = v.z; xa[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < 64; j += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j], bw[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j + 1], bw[j + 1], acc1, 0, 0, 0);
            }
        }
    }
    // each wavefront's 32 x 32 partial into the head of its own buffer, then the sum in wavefront order
#pragma unroll
    for (int e = 0; e < 16; ++e) my[((e & 3) + 8 * (e >> 2) + 4 * lk) * 32 + li] = acc0[e] + acc1[e];
    __syncthreads();
    const int col = cb * 32 + (tid & 31);
    const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = (tid >> 5) + 16 * h;
        const int64_t r = r0 + row;
        if (r >= a.rows) continue;
        float v = 0.f;
        if (r < rows_eff) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v += lds[q * (32 * SK_LD) + row * 32 + (tid & 31)];
            v += bv;
            if (a.seg_stat) {
                const float* st = a.seg_stat + 2 * r * a.KS + 1;
                for (int ks = 0; ks < a.KS; ++ks)
                    if (st[2 * ks] > 0.f) v += a.gamma[ks * 128 + col];
            }
            if (a.relu) v = fmaxf(v, 0.f);
        }
        a.Y[r * a.ldy + col] = v;
    }
}

// K-split: Y = act(sum over slabs of ws + bias); always: rows beyond the batch's own count (static capacity) get zeros
__global__ void __launch_bounds__(256) k_linear_splitk_finish(SplitKArgs a) {
    const int64_t rows_eff = sk_rows_eff(a);
    const int n4 = a.N >> 2;
    const bool ksplit = a.KS > 1;
    const int64_t first = ksplit ? 0 : rows_eff;
    const int64_t total = (a.rows - first) * n4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t r = first + q / n4;
        const int c = (int)(q % n4) * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows_eff) {
            for (int ks = 0; ks < a.KS; ++ks) {
                const float4 v = *(const float4*)(a.ws + ((int64_t)ks * a.rows + r) * 128 + c);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            if (a.bias) { const float4 b = *(const float4*)(a.bias + c); s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w; }
            if (a.seg_stat) {         // + gamma[slot] for every non-empty (row, slot) segment, slots in order
                const float* st = a.seg_stat + 2 * r * a.KS + 1;
                for (int ks = 0; ks < a.KS; ++ks) {
                    if (st[2 * ks] > 0.f) {
                        const float4 gm = *(const float4*)(a.gamma + ks * 128 + c);
                        s.x += gm.x; s.y += gm.y; s.z += gm.z; s.w += gm.w;
                    }
                }
            }
            if (a.relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
        }
        *(float4*)(a.Y + r * a.ldy + c) = s;
    }
}

}  // namespace

extern "C" int64_t kgw_linear_splitk_workspace_floats(int64_t rows, int32_t K, int32_t N) {
    if (rows <= 0 || K <= 128 || N != 128) return 0;
    return (int64_t)(K / 128) * rows * 128;
}

namespace {
__global__ void __launch_bounds__(1024) k_ind_colsum(ColsumJobs J) {
    // block = (relation slot r, group of 32 columns); thread = (row phase 0..31, column): rows ph, ph + 32, ... added in
    // order, four independent loads in flight per thread; the phases are folded through LDS in phase order (deterministic)
    __shared__ float sm[32][32];
    int jq = 0;
    while (jq + 1 < J.n && (int)blockIdx.x >= J.blk0[jq + 1]) ++jq;
    const float* __restrict__ seg_stat = J.seg_stat[jq];
    const float* __restrict__ dY = J.dY[jq];
    float* __restrict__ dgamma = J.dgamma[jq];
    const int64_t ldy = J.ldy[jq], rows = J.rows[jq];
    const int R = J.R[jq];
    const int bx = (int)blockIdx.x - J.blk0[jq];
    const int r = bx >> 2, c = (bx & 3) * 32 + (threadIdx.x & 31), ph = threadIdx.x >> 5;
    float s = 0.f;
    int64_t i = ph;
    for (; i + 96 < rows; i += 128) {
        float d[4], v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { d[q] = seg_stat[2 * ((i + 32 * q) * R + r) + 1]; v[q] = dY[(i + 32 * q) * ldy + c]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) s += d[q] > 0.f ? v[q] : 0.f;
    }
    for (; i < rows; i += 32)
        if (seg_stat[2 * (i * R + r) + 1] > 0.f) s += dY[i * ldy + c];
    sm[ph][threadIdx.x & 31] = s;
    __syncthreads();
    if (ph == 0) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) tot += sm[q][threadIdx.x];
        dgamma[r * 128 + c] = tot;
    }
}

// k_ind_colsum's work of block ``bxg`` on 256 threads: thread (phase group pg = 0..7, column): the four phases pg, pg + 8, pg + 16,
// pg + 24 one after the other, each exactly as a thread of k_ind_colsum adds it; the 32 phase sums folded in the same order -- the
// same bits
__device__ __forceinline__ void ind_colsum_block256(const ColsumJobs& J, const int bxg, float* lds) {
    float (*sm)[32] = (float (*)[32])lds;
    int jq = 0;
    while (jq + 1 < J.n && bxg >= J.blk0[jq + 1]) ++jq;
    const float* __restrict__ seg_stat = J.seg_stat[jq];
    const float* __restrict__ dY = J.dY[jq];
    float* __restrict__ dgamma = J.dgamma[jq];
    const int64_t ldy = J.ldy[jq], rows = J.rows[jq];
    const int R = J.R[jq];
    const int bx = bxg - J.blk0[jq];
    const int cl = threadIdx.x & 31, r = bx >> 2, c = (bx & 3) * 32 + cl, pg = threadIdx.x >> 5;
    // (the four phases advance together, sixteen loads in flight; every phase still adds its own rows in its own order)
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t t = 0;
    for (; t + pg + 96 < rows; t += 128) {               // (phase pg, the thread's first, has the longest main loop)
        float d[4][4], v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = t + pg + 8 * u;
            const bool on = i + 96 < rows;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t row = on ? i + 32 * q : 0;
                d[u][q] = seg_stat[2 * (row * R + r) + 1]; v[u][q] = dY[row * ldy + c];
                if (!on) d[u][q] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) s[u] += d[u][q] > 0.f ? v[u][q] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        // where phase ph's main loop stopped: the first multiple of 128 (from ph) with i + 96 >= rows
        const int ph = pg + 8 * u;
        int64_t i = ph;
        if (rows > ph + 96) i = ph + ((rows - ph - 97) / 128 + 1) * 128;
        for (; i < rows; i += 32)
            if (seg_stat[2 * (i * R + r) + 1] > 0.f) s[u] += dY[i * ldy + c];
        sm[ph][cl] = s[u];
    }
    __syncthreads();
    if (pg == 0) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) tot += sm[q][cl];
        dgamma[r * 128 + c] = tot;
    }
}

// The backward of a layer's relation transform in ONE launch: everything that is a function of d(output) alone --
//   the weight / bias gradients  dW^T = Z^T dY  (k_tn_gemm<2,2>'s row-block partials; k_tn_reduce follows as before),
//   the dZ twins                 dZ = dY W^T    (k_linear_splitk<false>),
//   the d gamma sums of a folded layer          (k_ind_colsum)
// -- as blocks of one grid.  They are independent of each other, each is a few hundred latency-bound blocks at two per CU, and as
// three launches one after the other each waits for the last block of the one before it.  Same code per block, same values.
// k_readout_train_fold (one block of 1 024 threads: thread = (c, g), 129 x 7) on a 256-thread block that walks the same (c, g) pairs:
// the step's read-out fold as ONE MORE block of the launch that follows it (kgw_transform_bwd_ex's fold_in).  lds: >= 1.5 k floats
__device__ __forceinline__ void readout_train_fold_block256(const KgwReadoutFold& F, float* lds) {
    float (*sm)[KGW_C + 1] = (float (*)[KGW_C + 1])lds;            // [7][129]
    double* sd = (double*)(lds + 1024);                            // [256] (8-byte aligned: the LDS base is 16-byte aligned)
    const float* __restrict__ part = F.scratch;
    const int nb = F.nb, n = F.n;
    {   // the thread's (up to) four (c, g) pairs side by side: 16 loads in flight instead of 4 (one pair after the other made this
        // block's latency 3 x the 1 024-thread kernel's -- longer than the launch it rides in)
        int cc[4], gg[4];
        bool ok[4];
        float a0[4], a1[4], a2[4], a3[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = threadIdx.x + 256 * k;
            ok[k] = idx < 7 * (KGW_C + 1);
            cc[k] = ok[k] ? idx % (KGW_C + 1) : 0; gg[k] = ok[k] ? idx / (KGW_C + 1) : 0;
            a0[k] = a1[k] = a2[k] = a3[k] = 0.f;
        }
        // (every pair walks q = g, g + 7, ...: the trip counts differ by at most one between the groups -- the common part unrolled
        //  over the four pairs, the rest pair by pair, every accumulator in the 1 024-thread kernel's order)
        int qn = 0;                                            // full rounds of 28 every pair has
        while (6 + 28 * qn + 21 < nb) ++qn;
        for (int r = 0; r < qn; ++r) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = gg[k] + 28 * r;
                const float* p = part + (int64_t)q * (KGW_C + 1) + cc[k];
                a0[k] += p[0]; a1[k] += p[7 * (KGW_C + 1)]; a2[k] += p[14 * (KGW_C + 1)]; a3[k] += p[21 * (KGW_C + 1)];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int q = gg[k] + 28 * qn;
            for (; q + 21 < nb; q += 28) {
                a0[k] += part[(int64_t)q * (KGW_C + 1) + cc[k]];        a1[k] += part[(int64_t)(q + 7) * (KGW_C + 1) + cc[k]];
                a2[k] += part[(int64_t)(q + 14) * (KGW_C + 1) + cc[k]]; a3[k] += part[(int64_t)(q + 21) * (KGW_C + 1) + cc[k]];
            }
            for (; q < nb; q += 7) a0[k] += part[(int64_t)q * (KGW_C + 1) + cc[k]];
            if (ok[k]) sm[gg[k]][cc[k]] = (a0[k] + a1[k]) + (a2[k] + a3[k]);
        }
    }
    {
        double acc = 0.0;
        for (int q = threadIdx.x; q < n; q += 256) acc += F.terms[q];
        sd[threadIdx.x] = acc;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sd[threadIdx.x] += sd[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) F.loss[0] = sd[0] / (double)n;
    if (threadIdx.x <= KGW_C) {
        const int c = threadIdx.x;
        const float t = ((sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c])) + ((sm[4][c] + sm[5][c]) + sm[6][c]);
        if (c < KGW_C) F.dw_lin[c] = t; else F.db_lin[0] = t;
    }
}

// (round 5: + the reduce blocks of an EARLIER product group whose second launch was left pending -- JR, X.n_rd: the last blocks)
struct TransformBwdIdx { int tn_flat0[TN_MAX_JOBS + 1]; int n_sk, n_tn, n_cs, n_rd, rd_gy, rd_gz, has_fold; KgwReadoutFold fold; };
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_transform_bwd(TnJobs JT, SplitKJobs JS, ColsumJobs JC, TnJobs JR, TransformBwdIdx X) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (the column-sum blocks first: few, and the longest -- a row walk per block; then the products' row blocks, then the twins)
    const int b = (int)blockIdx.x;
    if (b >= X.n_cs + X.n_tn + X.n_sk + X.n_rd) {
        readout_train_fold_block256(X.fold, lds);             // (the very last block, when there is one)
    } else if (b >= X.n_cs + X.n_tn + X.n_sk) {
        tn_reduce_plan_block(JR, X.rd_gy, X.rd_gz, b - X.n_cs - X.n_tn - X.n_sk, lds);
    } else if (b < X.n_cs) {
        ind_colsum_block256(JC, b, lds);
    } else if (b < X.n_cs + X.n_tn) {
        const int t = b - X.n_cs;
        int jq = 0;
        while (jq + 1 < JT.n && t >= X.tn_flat0[jq + 1]) ++jq;
        const TnJob& T = JT.j[jq];
        const int l = t - X.tn_flat0[jq];
        const int bx = l % T.nblk, rest = l / T.nblk;
        tn_gemm_block<2, 2>(T, bx, rest % T.gy, rest / T.gy, lds);
    } else {
        splitk_block<false>(JS, b - X.n_cs - X.n_tn, (float (*)[32 * SK_LD])lds);
    }
}

int splitk_launch(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy, int64_t rows,
                  int32_t K, int32_t N, int32_t relu, int32_t w_is_kn, float* workspace, int64_t workspace_floats,
                  const int32_t* rows_dev, const float* seg_stat, const float* gamma, kgw_stream_t stream_);
}  // namespace

extern "C" int kgw_ind_colsum_multi(int32_t n_jobs, const KgwSplitKJob* jobs, kgw_stream_t stream_) {
    if (n_jobs <= 0) return KGW_OK;
    if (!jobs) return KGW_E_NULL;
    if (n_jobs > SK_MAX_JOBS) return KGW_E_RANGE;
    ColsumJobs J{};
    int blk = 0;
    for (int q = 0; q < n_jobs; ++q) {
        const KgwSplitKJob& D = jobs[q];
        if (!D.seg_stat || !D.Y || !D.dgamma) return KGW_E_NULL;
        if (D.rows < 0 || D.K <= 0 || (D.K & 127)) return KGW_E_RANGE;
        J.seg_stat[q] = D.seg_stat; J.dY[q] = D.Y; J.dgamma[q] = D.dgamma; J.ldy[q] = D.ldy; J.rows[q] = D.rows; J.R[q] = D.K / 128;
        J.blk0[q] = blk;
        blk += 4 * (D.K / 128);
    }
    J.blk0[n_jobs] = blk; J.n = n_jobs;
    k_ind_colsum<<<blk, 1024, 0, (hipStream_t)stream_>>>(J);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_tn_reduce_launch(const KgwTnReducePlan* plan, kgw_stream_t stream_) {
    if (!plan) return KGW_E_NULL;
    const TnReducePlan& R = *(const TnReducePlan*)plan;
    if (!R.valid) return KGW_OK;
    k_tn_reduce<2, 2><<<dim3(TN22_FRAG / 64, R.gy_max, R.gz_max * R.n), 256, 0, (hipStream_t)stream_>>>(R.J, R.gz_max);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_transform_bwd_ex(int32_t n_tn, const KgwTnJob* tn_jobs, int32_t n_sk, const KgwSplitKJob* sk_jobs, int32_t n_cs,
                                    const KgwSplitKJob* cs_jobs, const KgwTnReducePlan* ride_in, KgwTnReducePlan* defer_out,
                                    const KgwReadoutFold* fold_in, kgw_stream_t stream_) {
    if (defer_out) ((TnReducePlan*)defer_out)->valid = 0;
    if (fold_in && (!fold_in->scratch || !fold_in->terms || !fold_in->dw_lin || !fold_in->db_lin || !fold_in->loss)) return KGW_E_NULL;
    if (fold_in && (fold_in->n <= 0 || fold_in->nb <= 0)) return KGW_E_RANGE;
    if (n_tn < 0 || n_sk < 0 || n_cs < 0 || n_tn > TN_MAX_JOBS || n_sk > SK_MAX_JOBS || n_cs > SK_MAX_JOBS) return KGW_E_RANGE;
    if ((n_tn && !tn_jobs) || (n_sk && !sk_jobs) || (n_cs && !cs_jobs)) return KGW_E_NULL;
    const TnReducePlan* RI = (const TnReducePlan*)ride_in;
    if (RI && !RI->valid) RI = nullptr;
    if (n_tn + n_sk + n_cs == 0) {
        if (fold_in) { const int rc = kgw_readout_train_fold(fold_in, stream_); if (rc) return rc; }
        return RI ? kgw_tn_reduce_launch(ride_in, stream_) : KGW_OK;
    }
    hipStream_t st = (hipStream_t)stream_;
    auto aligned8 = [](const void* p) { return ((uintptr_t)p & 7) == 0; };
    // weight-gradient products: kgw_tn_gemm_multi's checks and plan (64 x 64-per-wavefront tiling)
    TnPlan P{};
    if (n_tn) {
        TnDesc d[TN_MAX_JOBS];
        for (int q = 0; q < n_tn; ++q) {
            const KgwTnJob& j = tn_jobs[q];
            if (!j.A || !j.B || !j.C || !j.workspace) return KGW_E_NULL;
            if (j.M <= 0 || j.N <= 0 || j.rows <= 0 || j.lda < j.M || j.ldb < j.N || j.ldc < (j.c_transposed ? j.M : j.N)) return KGW_E_RANGE;
            if (j.colsum_a && (j.colsum_repeat < 1 || (j.colsum_repeat > 1 && j.colsum_ld < j.M))) return KGW_E_RANGE;
            if ((j.M & 1) || (j.lda & 1) || !aligned8(j.A) || (j.N & 1) || (j.ldb & 1) || !aligned8(j.B)) return KGW_E_UNSUPPORTED;
            d[q] = TnDesc{j.A, j.lda, j.M, j.B, j.ldb, j.N, j.rows, j.C, j.ldc, j.c_transposed != 0, j.colsum_a,
                          j.colsum_a ? j.colsum_repeat : 0, j.colsum_ld, j.workspace, j.workspace_floats, j.rows_dev};
        }
        const int rc = launch_tn_jobs<2, 2>(d, n_tn, st, nullptr, &P);
        if (rc != KGW_OK) return rc;
    }
    TransformBwdIdx X{};
    X.tn_flat0[0] = 0;
    for (int q = 0; q < n_tn; ++q) X.tn_flat0[q + 1] = X.tn_flat0[q] + P.J.j[q].nblk * P.J.j[q].gy * P.J.j[q].gz;
    X.n_tn = X.tn_flat0[n_tn];
    // dZ twins: kgw_linear_splitk_multi's K == 128 kind with [N, K] weights
    SplitKJobs JS{};
    int blk = 0, n = 0;
    for (int q = 0; q < n_sk; ++q) {
        const KgwSplitKJob& D = sk_jobs[q];
        if (D.rows == 0) continue;
        if (!D.X || !D.W || !D.Y) return KGW_E_NULL;
        if (D.rows < 0 || D.N <= 0) return KGW_E_RANGE;
        if (D.K != 128 || D.seg_stat || D.w_is_kn || (D.N & 127) || (D.ldx & 3) || (D.ldw & 3) || (D.ldy & 3) || !aligned16(D.X) ||
            !aligned16(D.W) || !aligned16(D.Y) || (D.bias && !aligned16(D.bias)))
            return KGW_E_UNSUPPORTED;
        SplitKArgs a{D.X, D.ldx, D.W, D.ldw, D.bias, D.Y, D.ldy, nullptr, D.rows, D.K, D.N, D.relu, D.w_is_kn,
                     (int)((D.rows + 31) / 32), D.K / 128, D.N / 128, 1, nullptr, nullptr, nullptr};
        const int target = 512;
        int G = (target + a.NS - 1) / a.NS;
        if (G > a.RT) G = a.RT;
        if (G < 1) G = 1;
        a.G = G;
        JS.blk0[n] = blk;
        blk += a.NS * G;
        JS.j[n++] = a;
    }
    JS.blk0[n] = blk; JS.n = n;
    X.n_sk = blk;
    ColsumJobs JC{};
    blk = 0;
    for (int q = 0; q < n_cs; ++q) {
        const KgwSplitKJob& D = cs_jobs[q];
        if (!D.seg_stat || !D.Y || !D.dgamma) return KGW_E_NULL;
        if (D.rows < 0 || D.K <= 0 || (D.K & 127)) return KGW_E_RANGE;
        JC.seg_stat[q] = D.seg_stat; JC.dY[q] = D.Y; JC.dgamma[q] = D.dgamma; JC.ldy[q] = D.ldy; JC.rows[q] = D.rows; JC.R[q] = D.K / 128;
        JC.blk0[q] = blk;
        blk += 4 * (D.K / 128);
    }
    JC.blk0[n_cs] = blk; JC.n = n_cs;
    X.n_cs = blk;
    constexpr int FRAG = 2 * 2 * 16 * 64;
    constexpr size_t lds_bytes = (size_t)(2 * FRAG + 4 * 32 * 2) * sizeof(float);
    static_assert(lds_bytes >= 2 * 32 * SK_LD * sizeof(float) && lds_bytes >= 32 * 32 * sizeof(float), "one LDS buffer serves the three block kinds");
    TnJobs JRd{};
    if (RI) { JRd = RI->J; X.n_rd = tn_reduce_plan_blocks(*RI); X.rd_gy = RI->gy_max; X.rd_gz = RI->gz_max; }
    if (fold_in) { X.has_fold = 1; X.fold = *fold_in; }
    const int total = X.n_sk + X.n_tn + X.n_cs + X.n_rd + X.has_fold;
    if (total > 0) {
        k_transform_bwd<<<total, 256, lds_bytes, st>>>(P.J, JS, JC, JRd, X);
        KGW_LAUNCH_CHECK();
    }
    if (n_tn && !P.all_direct) {
        if (defer_out) {
            TnReducePlan& R = *(TnReducePlan*)defer_out;
            R.valid = 1; R.gy_max = P.gy_max; R.gz_max = P.gz_max; R.n = n_tn; R.J = P.J;
            R.blocks = tn_reduce_plan_blocks(R);
        } else {
            k_tn_reduce<2, 2><<<dim3(FRAG / 64, P.gy_max, P.gz_max * n_tn), 256, 0, st>>>(P.J, P.gz_max);
            KGW_LAUNCH_CHECK();
        }
    }
    return KGW_OK;
}

extern "C" int kgw_transform_bwd(int32_t n_tn, const KgwTnJob* tn_jobs, int32_t n_sk, const KgwSplitKJob* sk_jobs, int32_t n_cs,
                                 const KgwSplitKJob* cs_jobs, kgw_stream_t stream_) {
    return kgw_transform_bwd_ex(n_tn, tn_jobs, n_sk, sk_jobs, n_cs, cs_jobs, nullptr, nullptr, nullptr, stream_);
}

extern "C" int kgw_ind_colsum(const float* seg_stat, const float* dY, int64_t ldy, int64_t rows, int32_t R, float* dgamma,
                              kgw_stream_t stream_) {
    if (R <= 0) return KGW_OK;
    KgwSplitKJob j{};
    j.seg_stat = seg_stat; j.Y = const_cast<float*>(dY); j.ldy = ldy; j.rows = rows; j.K = R * 128; j.dgamma = dgamma;
    return kgw_ind_colsum_multi(1, &j, stream_);
}

extern "C" int kgw_linear_splitk(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y,
                                 int64_t ldy, int64_t rows, int32_t K, int32_t N, int32_t relu, int32_t w_is_kn,
                                 float* workspace, int64_t workspace_floats, const int32_t* rows_dev,
                                 kgw_stream_t stream_) {
    return splitk_launch(X, ldx, W, ldw, bias, Y, ldy, rows, K, N, relu, w_is_kn, workspace, workspace_floats, rows_dev,
                         nullptr, nullptr, stream_);
}

extern "C" int kgw_linear_splitk_ind(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y,
                                     int64_t ldy, int64_t rows, int32_t K, int32_t relu, const float* seg_stat,
                                     const float* gamma, float* workspace, int64_t workspace_floats,
                                     const int32_t* rows_dev, kgw_stream_t stream_) {
    if (!seg_stat || !gamma) return KGW_E_NULL;
    if (K <= 128 || !aligned16(gamma)) return KGW_E_UNSUPPORTED;
    return splitk_launch(X, ldx, W, ldw, bias, Y, ldy, rows, K, 128, relu, 1, workspace, workspace_floats, rows_dev, seg_stat,
                         gamma, stream_);
}

namespace {
int splitk_fused_launch(const SplitKJobs& J, hipStream_t st) {
    const size_t lds_bytes = (size_t)8 * 32 * SK_LD * sizeof(float);
    static KgwPerDevice attr_once;
    if (attr_once.need()) {
        KGW_HIP(hipFuncSetAttribute((const void*)k_linear_splitk_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    k_linear_splitk_fused<<<J.blk0[J.n], 512, lds_bytes, st>>>(J);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

int splitk_launch(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy, int64_t rows,
                  int32_t K, int32_t N, int32_t relu, int32_t w_is_kn, float* workspace, int64_t workspace_floats,
                  const int32_t* rows_dev, const float* seg_stat, const float* gamma, kgw_stream_t stream_) {
    if (rows == 0) return KGW_OK;
    if (!X || !W || !Y) return KGW_E_NULL;
    if (rows < 0 || K <= 0 || N <= 0) return KGW_E_RANGE;
    if ((K & 127) || (N & 127) || (K != 128 && N != 128) || (ldx & 3) || (ldw & 3) || (ldy & 3) || !aligned16(X) ||
        !aligned16(W) || !aligned16(Y) || (bias && !aligned16(bias)))
        return KGW_E_UNSUPPORTED;
    SplitKArgs a{X, ldx, W, ldw, bias, Y, ldy, workspace, rows, K, N, relu, w_is_kn, (int)((rows + 31) / 32),
                 K / 128, N / 128, 1, rows_dev, seg_stat, gamma};
    const int nslab = a.KS > 1 ? a.KS : a.NS;
    if (a.KS > 1 && w_is_kn && N == 128) {        // forward transform: one launch
        SplitKJobs J{};
        J.j[0] = a; J.n = 1; J.blk0[0] = 0; J.blk0[1] = a.RT * 4;
        return splitk_fused_launch(J, (hipStream_t)stream_);
    }
    if (a.KS > 1 && (!workspace || workspace_floats < kgw_linear_splitk_workspace_floats(rows, K, N))) return KGW_E_NULL;
    // row-tile groups per slab: about two blocks per CU in total, at most one tile... at least one tile per block
    const int target = 512;
    int G = (target + nslab - 1) / nslab;
    if (G > a.RT) G = a.RT;
    if (G < 1) G = 1;
    a.G = G;
    hipStream_t st = (hipStream_t)stream_;
    SplitKJobs J{};
    J.j[0] = a; J.n = 1; J.blk0[0] = 0; J.blk0[1] = nslab * G;
    if (w_is_kn) k_linear_splitk<true><<<nslab * G, 256, 0, st>>>(J);
    else k_linear_splitk<false><<<nslab * G, 256, 0, st>>>(J);
    KGW_LAUNCH_CHECK();
    if (a.KS > 1 || rows_dev) {
        int64_t g = (rows * (N / 4) + 255) / 256;
        if (g > KGW_GRID) g = KGW_GRID;
        k_linear_splitk_finish<<<(int)g, 256, 0, st>>>(a);
        KGW_LAUNCH_CHECK();
    }
    return KGW_OK;
}
}  // namespace

extern "C" int kgw_linear_splitk_multi(int32_t n_jobs, const KgwSplitKJob* jobs, kgw_stream_t stream_) {
    if (n_jobs <= 0) return KGW_OK;
    if (!jobs) return KGW_E_NULL;
    if (n_jobs > SK_MAX_JOBS) return KGW_E_RANGE;
    SplitKJobs J{};
    int blk = 0, kind = -1, n = 0;                       // kind 0: forward transform (fused kernel); 1: dZ twin (K == 128)
    for (int q = 0; q < n_jobs; ++q) {
        const KgwSplitKJob& D = jobs[q];
        if (D.rows == 0) continue;
        if (!D.X || !D.W || !D.Y) return KGW_E_NULL;
        if (D.rows < 0 || D.K <= 0 || D.N <= 0) return KGW_E_RANGE;
        if ((D.K & 127) || (D.N & 127) || (D.ldx & 3) || (D.ldw & 3) || (D.ldy & 3) || !aligned16(D.X) || !aligned16(D.W) ||
            !aligned16(D.Y) || (D.bias && !aligned16(D.bias)) || (D.gamma && !aligned16(D.gamma)))
            return KGW_E_UNSUPPORTED;
        const int k = (D.K > 128 && D.N == 128 && D.w_is_kn) ? 0 : ((D.K == 128 && !D.seg_stat) ? 1 : -1);
        if (k < 0 || (kind >= 0 && k != kind) || (n > 0 && (D.w_is_kn != 0) != (J.j[0].w_kn != 0))) return KGW_E_UNSUPPORTED;
        if (D.seg_stat && !D.gamma) return KGW_E_NULL;
        kind = k;
        SplitKArgs a{D.X, D.ldx, D.W, D.ldw, D.bias, D.Y, D.ldy, nullptr, D.rows, D.K, D.N, D.relu, D.w_is_kn,
                     (int)((D.rows + 31) / 32), D.K / 128, D.N / 128, 1, nullptr, D.seg_stat, D.gamma};
        J.blk0[n] = blk;
        if (k == 0) {
            blk += a.RT * 4;
        } else {
            // row-tile groups per slab: the jobs together aim at about two blocks per CU
            const int target = 512;
            int G = (target + a.NS - 1) / a.NS;
            if (G > a.RT) G = a.RT;
            if (G < 1) G = 1;
            a.G = G;
            blk += a.NS * G;
        }
        J.j[n++] = a;
    }
    if (n == 0) return KGW_OK;
    J.blk0[n] = blk; J.n = n;
    hipStream_t st = (hipStream_t)stream_;
    if (kind == 0) return splitk_fused_launch(J, st);
    if (J.j[0].w_kn) k_linear_splitk<true><<<blk, 256, 0, st>>>(J);
    else k_linear_splitk<false><<<blk, 256, 0, st>>>(J);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}
