// kgw_dense_optim.h -- part of kgw_dense.hip (ONE translation unit, split by kernel family in round 6; include order matters:
// later families use device functions of earlier ones): backward of the resident gene layer rows (kgw_scatter_relu_rows), Adam (kgw_adam) and the fused optimiser launch (kgw_adam_fused, kgw_grad_finish).
#pragma once

// ======================================================================================================
// kgw_scatter_relu_rows: backward of "rows ids of relu(X W^T + b) computed on a RESIDENT matrix" (the 5120-wide gene
// layer runs on all N genes and the batch takes its rows): dz[row] = g[g2l[row]] * (h[row] > 0) for every row of the
// resident matrix (zero where the node is not in the batch), and colsum[c] = sum_row dz[row][c] -- the framework's
// zero fill + index_add + ReLU mask + column reduction (5 launches) in 2.  Deterministic (fixed partial layout).
// ======================================================================================================
namespace {
__global__ void __launch_bounds__(256) k_scatter_relu_rows(const float* __restrict__ g, const int32_t* __restrict__ g2l,
                                                           const float* __restrict__ h, int64_t n_rows,
                                                           float* __restrict__ dz, float* __restrict__ part) {
    __shared__ float4 red[8][32];
    const int r8 = threadIdx.x >> 5, c4 = threadIdx.x & 31;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r = (int64_t)blockIdx.x * 8 + r8; r < n_rows; r += (int64_t)gridDim.x * 8) {
        const int pos = g2l[r];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pos >= 0) {
            const float4 gv = ((const float4*)g)[(int64_t)pos * 32 + c4];
            const float4 hv = ((const float4*)h)[r * 32 + c4];
            v.x = hv.x > 0.f ? gv.x : 0.f; v.y = hv.y > 0.f ? gv.y : 0.f;
            v.z = hv.z > 0.f ? gv.z : 0.f; v.w = hv.w > 0.f ? gv.w : 0.f;
        }
        ((float4*)dz)[r * 32 + c4] = v;
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    red[r8][c4] = acc;
    __syncthreads();
    if (r8 == 0) {
#pragma unroll
        for (int k = 1; k < 8; ++k) { const float4 o = red[k][c4]; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        ((float4*)part)[(int64_t)blockIdx.x * 32 + c4] = acc;
    }
}

// 128 columns x 8 groups of partial rows, eight independent loads in flight per thread, fixed order
__global__ void __launch_bounds__(1024) k_colsum_fold(const float* __restrict__ part, int nblk, float* __restrict__ out) {
    __shared__ float sm[8][128];
    const int c = threadIdx.x & 127, gq = threadIdx.x >> 7;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = gq;
    for (; b + 56 < nblk; b += 64) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += part[(int64_t)(b + 8 * k) * 128 + c];
    }
    for (int k = 0; b < nblk; b += 8, ++k) s[k & 7] += part[(int64_t)b * 128 + c];
    sm[gq][c] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (gq == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sm[k][c];
        out[c] = t;
    }
}
}  // namespace

extern "C" int64_t kgw_scatter_relu_rows_workspace_floats(int64_t n_rows) {
    int64_t nblk = (n_rows + 7) / 8;
    if (nblk > 256) nblk = 256;
    return (nblk > 0 ? nblk : 1) * 128;
}

extern "C" int kgw_scatter_relu_rows(const float* g, const int32_t* g2l, const float* h, int64_t n_rows, float* dz,
                                     float* colsum, float* workspace, kgw_stream_t stream_) {
    if (!g2l || !h || !dz || !colsum || !workspace) return KGW_E_NULL;
    if (n_rows <= 0) return KGW_E_RANGE;
    if (!aligned16(h) || !aligned16(dz) || !aligned16(workspace) || (g && !aligned16(g))) return KGW_E_UNSUPPORTED;
    int64_t nblk = (n_rows + 7) / 8;
    if (nblk > 256) nblk = 256;
    k_scatter_relu_rows<<<(int)nblk, 256, 0, (hipStream_t)stream_>>>(g, g2l, h, n_rows, dz, workspace);
    KGW_LAUNCH_CHECK();
    k_colsum_fold<<<1, 1024, 0, (hipStream_t)stream_>>>(workspace, (int)nblk, colsum);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

// ======================================================================================================
// kgw_adam: torch.optim.Adam(lr, betas, eps, weight_decay as L2) of kgwas/kgwas.py:116,151 for ALL parameter
// tensors in one launch (the framework's capturable Adam issues ~100 small launches per step).  Same update
// order as torch: g += wd*p ; m = lerp(m, g, 1-b1) ; v = v*b2 + (1-b2)*g*g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).
// The step counter lives on the device so the call can sit inside a captured HIP graph.
// ======================================================================================================
namespace {

constexpr int ADAM_MAX = 64;
struct AdamTab {
    float* p[ADAM_MAX]; const float* g[ADAM_MAX]; float* m[ADAM_MAX]; float* v[ADAM_MAX];
    int64_t off[ADAM_MAX + 1];      // prefix sums of element counts
    int64_t coff[ADAM_MAX + 1];     // prefix sums of 1024-element work units
    unsigned char vec[ADAM_MAX];    // all four pointers 16-byte aligned: float4 path
    int n;
};

// One element's update.  Contraction is switched off and the one fused multiply-add written out, so that every place this is
// inlined (vector and scalar paths of k_adam, both paths of k_adam_fused) rounds identically: the fused launch must leave the
// same bits as the unfused one.
__device__ __forceinline__ void adam_update(float& p, float g0, float& m, float& v, float wd, float b1, float b2, float eps,
                                            float step_size, float bc2_sqrt) {
#pragma clang fp contract(off)
    const float g = fmaf(wd, p, g0);
    m = m + (1.0f - b1) * (g - m);
    v = v * b2 + (1.0f - b2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

// Work unit = 1024 consecutive elements of ONE tensor (256 threads x float4); the tensor of a unit is found once per
// unit with wave-uniform (scalar) comparisons, not per element.
__global__ void __launch_bounds__(256) k_adam(AdamTab T, int32_t* step, float lr, float b1, float b2, float eps, float wd) {
    const int t_now = *step + 1;                       // every thread reads the same pre-increment value
    const float bc1 = 1.0f - powf(b1, (float)t_now);
    const float bc2 = 1.0f - powf(b2, (float)t_now);
    const float step_size = lr / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    const int64_t units = T.coff[T.n];
    for (int64_t c = blockIdx.x; c < units; c += gridDim.x) {
        int lo = 0, hi = T.n;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (T.coff[mid] <= c) lo = mid; else hi = mid; }
        const int64_t n = T.off[lo + 1] - T.off[lo];
        const int64_t j = (c - T.coff[lo]) * 1024 + (int64_t)threadIdx.x * 4;
        float* __restrict__ P = T.p[lo];
        const float* __restrict__ G = T.g[lo];
        float* __restrict__ M = T.m[lo];
        float* __restrict__ V = T.v[lo];
        if (j + 4 <= n && T.vec[lo]) {
            float4 p = *(float4*)(P + j), m = *(float4*)(M + j), v = *(float4*)(V + j);
            const float4 g0 = *(const float4*)(G + j);
            float pe[4] = {p.x, p.y, p.z, p.w}, ge[4] = {g0.x, g0.y, g0.z, g0.w}, me[4] = {m.x, m.y, m.z, m.w}, ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) adam_update(pe[e], ge[e], me[e], ve[e], wd, b1, b2, eps, step_size, bc2_sqrt);
            *(float4*)(M + j) = make_float4(me[0], me[1], me[2], me[3]);
            *(float4*)(V + j) = make_float4(ve[0], ve[1], ve[2], ve[3]);
            *(float4*)(P + j) = make_float4(pe[0], pe[1], pe[2], pe[3]);
        } else {
            for (int64_t i = j; i < n && i < j + 4; ++i) {
                float p = P[i], m = M[i], v = V[i];
                adam_update(p, G[i], m, v, wd, b1, b2, eps, step_size, bc2_sqrt);
                M[i] = m; V[i] = v; P[i] = p;
            }
        }
    }
}

__global__ void k_adam_tick(int32_t* step) { if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1; }

}  // namespace

static int adam_launch(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* numel, int32_t* step_dev, float lr, float beta1,
                       float beta2, float eps, float weight_decay, bool tick, kgw_stream_t stream_) {
    if (n_tensors == 0) return KGW_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !step_dev) return KGW_E_NULL;
    if (n_tensors < 0 || n_tensors > ADAM_MAX) return KGW_E_RANGE;
    AdamTab T;
    T.n = n_tensors;
    T.off[0] = 0;
    T.coff[0] = 0;
    for (int i = 0; i < n_tensors; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 0) return KGW_E_NULL;
        T.p[i] = params[i]; T.g[i] = grads[i]; T.m[i] = exp_avg[i]; T.v[i] = exp_avg_sq[i];
        T.off[i + 1] = T.off[i] + numel[i];
        T.coff[i + 1] = T.coff[i] + (numel[i] + 1023) / 1024;
        T.vec[i] = (((uintptr_t)params[i] | (uintptr_t)grads[i] | (uintptr_t)exp_avg[i] | (uintptr_t)exp_avg_sq[i]) & 15) == 0;
    }
    hipStream_t st = (hipStream_t)stream_;
    int64_t g = T.coff[n_tensors];
    if (g > 4 * KGW_GRID) g = 4 * KGW_GRID;
    if (g < 1) g = 1;
    k_adam<<<(int)g, 256, 0, st>>>(T, step_dev, lr, beta1, beta2, eps, weight_decay);
    KGW_LAUNCH_CHECK();
    if (tick) {
        k_adam_tick<<<1, 64, 0, st>>>(step_dev);
        KGW_LAUNCH_CHECK();
    }
    return KGW_OK;
}

extern "C" int kgw_adam(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, const int64_t* numel, int32_t* step_dev, float lr, float beta1,
                        float beta2, float eps, float weight_decay, kgw_stream_t stream_) {
    return adam_launch(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, step_dev, lr, beta1, beta2, eps, weight_decay, true,
                       stream_);
}

// the same without the launch that advances *step_dev: the caller does that later in the step (kgw_accumulate_stats_tick)
extern "C" int kgw_adam_notick(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* numel, int32_t* step_dev, float lr, float beta1,
                               float beta2, float eps, float weight_decay, kgw_stream_t stream_) {
    return adam_launch(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, step_dev, lr, beta1, beta2, eps, weight_decay, false,
                       stream_);
}

// ======================================================================================================
// kgw_adam_fused: the optimiser launch of a captured step.  On top of k_adam:
//   * gradients whose producers stopped after their first launch (kgw_tn_gemm_partial, kgw_mlp2_bwd_first_partial) are
//     finished here: a work unit of such a tensor is 64 of its elements x 4 groups of partial records, added in the SAME order
//     as k_tn_reduce / k_mlp2_bwd_fold (bit-identical gradients), then updated by the lanes that hold the sums.  Five
//     ~7 us launches of the 47-launch step disappear into this one;
//   * the block that finishes last (a device counter) advances the step counter and accumulates the running totals of
//     kgw_accumulate_stats_tick -- every other block has read the counter by then.
// ======================================================================================================
namespace {

struct AdamFTab {
    float* p[KGW_ADAM_FUSED_MAX]; float* g[KGW_ADAM_FUSED_MAX]; float* m[KGW_ADAM_FUSED_MAX]; float* v[KGW_ADAM_FUSED_MAX];
    int64_t off[KGW_ADAM_FUSED_MAX + 1];
    int64_t coff[KGW_ADAM_FUSED_MAX + 1];     // prefix sums of work units (1024 elements of a direct tensor, 64 of a sourced one)
    unsigned char vec[KGW_ADAM_FUSED_MAX];
    unsigned char src_of[KGW_ADAM_FUSED_MAX]; // index into src, 255 = the gradient tensor holds the gradient
    KgwGradSrc src[KGW_ADAM_FUSED_SRC];
    int n;
};
struct AdamTail { const KgwBatchMeta* meta; int64_t* stats; int32_t* done; int n_layers, n_hops; };

// sum over the partial records of element i of a sourced gradient; every thread of the block calls it (fl = element within the
// unit, G = group of records); the value is returned to the threads with G == 0
__device__ __forceinline__ float adam_src_sum(const KgwGradSrc& S, int64_t i, bool valid, int fl, int G, float* sm) {
    const int nblk = S.nblk;
    if (S.kind == KGW_GRAD_TN) {
        const int MT = S.MT, NT = S.NT;
        const int64_t FRAG = (int64_t)MT * NT * 1024;
        int m, n;
        if (!S.c_transposed) { m = (int)(i / S.N); n = (int)(i - (int64_t)m * S.N); }
        else                 { n = (int)(i / S.M); m = (int)(i - (int64_t)n * S.M); }
        const int by = m / (32 * MT), rm = m - by * 32 * MT, ta = rm % MT, ti = rm / MT;
        const int bz = n / (32 * NT), rn = n - bz * 32 * NT, tb = rn % NT, tj = rn / NT;
        const int lane = tj + 32 * ((ti >> 2) & 1), e = (ti & 3) + 4 * (ti >> 3);
        const float* p = S.ws + ((int64_t)bz * S.gy + by) * nblk * FRAG + ((ta * NT + tb) * 16 + e) * 64 + lane;
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (valid) {
            int b = G;
            for (; b + 28 < nblk; b += 32) {
#pragma unroll
                for (int q = 0; q < 8; ++q) s8[q] += p[(int64_t)(b + 4 * q) * FRAG];
            }
            for (int q = 0; b < nblk; b += 4, ++q) s8[q & 7] += p[(int64_t)b * FRAG];
        }
        sm[G * 64 + fl] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        __syncthreads();
        return (sm[fl] + sm[64 + fl]) + (sm[128 + fl] + sm[192 + fl]);
    }
    if (S.kind == KGW_GRAD_TN_COLSUM) {
        const int NC = 32 * S.MT, NG = 256 / NC;
        const int m = (int)i, by = m / NC, c = m - by * NC;
        const float* p = S.ws + (int64_t)by * nblk * NC + c;
        for (int gq = G; gq < NG; gq += 4) {
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                int b = gq;
                for (; b + 3 * NG < nblk; b += 4 * NG) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) s4[q] += p[(int64_t)(b + q * NG) * NC];
                }
                for (int q = 0; b < nblk; b += NG, ++q) s4[q & 3] += p[(int64_t)b * NC];
            }
            sm[gq * 64 + fl] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
        __syncthreads();
        float t = 0.f;
        for (int q = 0; q < NG; ++q) t += sm[q * 64 + fl];
        return t;
    }
    // KGW_GRAD_MLP2_W / _B: fragment (t * 16 + e) * 64 + lane of a block's 4096-float record holds
    // C[k = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)][col = 32 t + (lane & 31)]; d W1[col][k] for k < K1, d b1[col] at k == K1
    {
        int col, k;
        if (S.kind == KGW_GRAD_MLP2_W) { col = (int)(i / S.K1); k = (int)(i - (int64_t)col * S.K1); }
        else                           { col = (int)i; k = S.K1; }
        const int t = col >> 5, lane = (col & 31) + 32 * ((k >> 2) & 1), e = (k & 3) + 4 * (k >> 3);
        const float* p = S.ws + (t * 16 + e) * 64 + lane;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gg = G + 4 * u;
            // (k_mlp2_bwd_fold's order -- accumulator j takes records gg + 16 j, gg + 16 (j + 4), ... -- with the four loads of a
            //  round independent of each other: written as ``s4[q & 3]`` the loop is one dependent load after the other)
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                int b = gg;
                for (; b + 48 < nblk; b += 64) {
                    float x[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = p[(int64_t)(b + 16 * j) * 4096];
#pragma unroll
                    for (int j = 0; j < 4; ++j) s4[j] += x[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (b + 16 * j < nblk) s4[j] += p[(int64_t)(b + 16 * j) * 4096];
            }
            sm[gg * 64 + fl] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
        __syncthreads();
        float sv = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < 16; k4 += 4) sv += (sm[k4 * 64 + fl] + sm[(k4 + 1) * 64 + fl]) + (sm[(k4 + 2) * 64 + fl] + sm[(k4 + 3) * 64 + fl]);
        return sv;
    }
}

// UPD = false (kgw_grad_finish): no update -- the finished gradient goes to T.p[i], here the tensor's slot in a flat all-reduce
// bucket (and into the gradient tensor itself where it was a sum of partial records); no counters, nothing read from ``step``.
template <bool UPD>
__global__ void __launch_bounds__(256) k_adam_fused(AdamFTab T, AdamTail Z, int32_t* step, float lr, float b1, float b2, float eps,
                                                    float wd) {
    __shared__ float sm[32 * 33];                       // 16 x 64 partial sums of a sourced unit / one 32 x 33 tile (G3T)
    const int t_now = UPD ? *step + 1 : 1;             // (the counter moves only after every block has arrived at the end)
    const float bc1 = 1.0f - powf(b1, (float)t_now);
    const float bc2 = 1.0f - powf(b2, (float)t_now);
    const float step_size = lr / bc1;
    const float bc2_sqrt = sqrtf(bc2);
    const int64_t units = T.coff[T.n];
    const int fl = threadIdx.x & 63, G = threadIdx.x >> 6;
    for (int64_t c = blockIdx.x; c < units; c += gridDim.x) {
        int lo = 0, hi = T.n;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (T.coff[mid] <= c) lo = mid; else hi = mid; }
        const int64_t n = T.off[lo + 1] - T.off[lo];
        float* __restrict__ P = T.p[lo];
        float* __restrict__ Gr = T.g[lo];
        float* __restrict__ M = T.m[lo];
        float* __restrict__ V = T.v[lo];
        const int si = T.src_of[lo];
        if (si != 255 && T.src[si].kind == KGW_GRAD_G3T) {
            // the weight gradient of the first gene Linear, out[col][row] = sum over K ranges of ws[s][row][col]: one 32 x 32 tile
            // per unit through LDS exactly like k_g3_reduce_t (same order), then the update of the tile's 1024 parameters, and --
            // S.packed -- the three bf16 pieces of the UPDATED values in kgw_gemm3's operand image (k_g3_pack<false>'s layout):
            // the next forward product finds its B operand ready
            const KgwGradSrc& S = T.src[si];
            const int64_t u = c - T.coff[lo];
            const int64_t Mr = S.M, r0 = (u >> 2) * 32;
            const int cb = (int)(u & 3) * 32;
            float (*tl)[33] = (float (*)[33])sm;
            {
                const int r = threadIdx.x >> 3, c4 = threadIdx.x & 7;
                const float4* w = (const float4*)S.ws + (r0 + r) * 32 + (cb >> 2) + c4;
                // (k_g3_reduce_t's order, K range after K range; four loads in flight)
                const int64_t ks = Mr * 32;
                float4 a4 = w[0];
                int k = 1;
                for (; k + 3 < S.nblk; k += 4) {
                    float4 x[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = w[(int64_t)(k + q) * ks];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { a4.x += x[q].x; a4.y += x[q].y; a4.z += x[q].z; a4.w += x[q].w; }
                }
                for (; k < S.nblk; ++k) {
                    const float4 x = w[(int64_t)k * ks];
                    a4.x += x.x; a4.y += x.y; a4.z += x.z; a4.w += x.w;
                }
                tl[r][4 * c4] = a4.x; tl[r][4 * c4 + 1] = a4.y; tl[r][4 * c4 + 2] = a4.z; tl[r][4 * c4 + 3] = a4.w;
            }
            __syncthreads();
            {
                const int r = threadIdx.x & 31;
                float pq[4], mq[4], vq[4], gq[4];
                if constexpr (!UPD) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int col = (threadIdx.x >> 5) + 8 * q;
                        const int64_t i = (int64_t)(cb + col) * Mr + r0 + r;
                        const float g = tl[r][col];
                        Gr[i] = g; P[i] = g;
                    }
                    __syncthreads();
                    continue;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {                  // (all loads of the thread's four elements first)
                    const int col = (threadIdx.x >> 5) + 8 * q;
                    const int64_t i = (int64_t)(cb + col) * Mr + r0 + r;
                    gq[q] = tl[r][col];
                    pq[q] = P[i]; mq[q] = M[i]; vq[q] = V[i];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = (threadIdx.x >> 5) + 8 * q;
                    const int64_t i = (int64_t)(cb + col) * Mr + r0 + r;
                    adam_update(pq[q], gq[q], mq[q], vq[q], wd, b1, b2, eps, step_size, bc2_sqrt);
                    Gr[i] = gq[q]; M[i] = mq[q]; V[i] = vq[q]; P[i] = pq[q];
                    tl[r][col] = pq[q];
                }
            }
            __syncthreads();
            if (S.packed && threadIdx.x < 128) {
                // image index (((c * 2 + j) * 3 + piece) * 4 + nt) * 64 + lane: the eight bf16 of a piece for
                // k = 32 c + 16 j + 8 (lane >> 5) + i, column 32 nt + (lane & 31); here k = the tile's rows, column = its columns
                const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
                const int64_t ch = r0 >> 5;
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = tl[16 * j + 8 * (lane >> 5) + e][lane & 31];
                if (S.flip && ((ch / S.flip) & 1)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = -x[e];
                }
                uint4 p1, p2, p3;
                kgw_split3x8(x, p1, p2, p3);
                uint4* o = (uint4*)S.packed + ((ch * 2 + j) * 3 * 4 + (cb >> 5)) * 64 + lane;
                o[0] = p1; o[4 * 64] = p2; o[8 * 64] = p3;
            }
            __syncthreads();
            continue;
        }
        if (si != 255) {
            const int64_t i = (c - T.coff[lo]) * 64 + fl;
            const bool valid = i < n;
            const float gs = adam_src_sum(T.src[si], i, valid, fl, G, sm);
            if (G == 0 && valid) {
                Gr[i] = gs;
                if constexpr (UPD) {
                    float p = P[i], m = M[i], v = V[i];
                    adam_update(p, gs, m, v, wd, b1, b2, eps, step_size, bc2_sqrt);
                    M[i] = m; V[i] = v; P[i] = p;
                } else {
                    P[i] = gs;
                }
            }
            __syncthreads();                           // (sm is reused by the block's next unit)
            continue;
        }
        const int64_t j = (c - T.coff[lo]) * 1024 + (int64_t)threadIdx.x * 4;
        if constexpr (!UPD) {                          // a complete gradient: copied to its slot
            if (j + 4 <= n && T.vec[lo]) *(float4*)(P + j) = *(const float4*)(Gr + j);
            else for (int64_t i = j; i < n && i < j + 4; ++i) P[i] = Gr[i];
            continue;
        }
        if (j + 4 <= n && T.vec[lo]) {
            float4 p = *(float4*)(P + j), m = *(float4*)(M + j), v = *(float4*)(V + j);
            const float4 g0 = *(const float4*)(Gr + j);
            float pe[4] = {p.x, p.y, p.z, p.w}, ge[4] = {g0.x, g0.y, g0.z, g0.w}, me[4] = {m.x, m.y, m.z, m.w}, ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) adam_update(pe[e], ge[e], me[e], ve[e], wd, b1, b2, eps, step_size, bc2_sqrt);
            *(float4*)(M + j) = make_float4(me[0], me[1], me[2], me[3]);
            *(float4*)(V + j) = make_float4(ve[0], ve[1], ve[2], ve[3]);
            *(float4*)(P + j) = make_float4(pe[0], pe[1], pe[2], pe[3]);
        } else {
            for (int64_t i = j; i < n && i < j + 4; ++i) {
                float p = P[i], m = M[i], v = V[i];
                adam_update(p, Gr[i], m, v, wd, b1, b2, eps, step_size, bc2_sqrt);
                M[i] = m; V[i] = v; P[i] = p;
            }
        }
    }
    // the last block to get here: step counter + running totals (k_accumulate_stats).  Two levels of counters, each on a
    // 128-byte line of its own: same-address device atomics are served one at a time (~20 ns each: 2 600 blocks on ONE counter
    // made this launch 55 us long), 64 first-level counters take <= grid / 64 arrivals each and the block that completes one moves
    // on to the top counter (64 arrivals).  Relaxed on purpose -- an acquire / release at agent scope is an L2 write-back +
    // invalidate per block on this multi-die part (135 us for the launch); nothing is published through the counters: the only
    // ordering needed is "every block has READ *step before the last one writes it", each block's read was consumed before its
    // atomic is issued, and the last block's store depends on the values its atomics return.
    if constexpr (!UPD) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int slot = (int)(blockIdx.x & 63), in_slot = ((int)gridDim.x - slot + 63) >> 6;
        const int n_slots = (int)gridDim.x < 64 ? (int)gridDim.x : 64;
        int32_t* c1 = Z.done + 32 * (1 + slot);
        if (__hip_atomic_fetch_add(c1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_slot - 1) {
            __hip_atomic_store(c1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(Z.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_slots - 1) {
                __hip_atomic_store(Z.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *step = t_now;
                if (Z.meta) {
                    const KgwBatchMeta* Mt = Z.meta;
                    for (int t = 0; t < Z.n_layers; ++t) Z.stats[t] += Mt->n_edges[t];
                    Z.stats[Z.n_layers] += Mt->edge_end[Z.n_hops - 1];
                    Z.stats[Z.n_layers + 1] |= Mt->error;
                }
            }
        }
    }
}

}  // namespace

static int adam_fused_table(AdamFTab& T, int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                            float* const* exp_avg_sq, const int64_t* numel, const KgwGradSrc* src);

extern "C" int kgw_adam_fused(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const int64_t* numel, const KgwGradSrc* src, int32_t* step_dev, float lr,
                              float beta1, float beta2, float eps, float weight_decay, const KgwBatchMeta* meta_dev,
                              int32_t n_layers, int32_t n_hops, int64_t* stats, int32_t* done_counter, kgw_stream_t stream_) {
    if (n_tensors < 0 || n_tensors > KGW_ADAM_FUSED_MAX) return KGW_E_RANGE;
    if (!step_dev || !done_counter) return KGW_E_NULL;
    if (n_tensors > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel)) return KGW_E_NULL;
    if (meta_dev && (!stats || n_layers < 1 || n_layers > KGW_MAX_LAYERS || n_hops < 1 || n_hops > n_layers)) return KGW_E_RANGE;
    AdamFTab T;
    const int rc = adam_fused_table(T, n_tensors, params, grads, exp_avg, exp_avg_sq, numel, src);
    if (rc != KGW_OK) return rc;
    AdamTail Z{meta_dev, stats, done_counter, n_layers, n_hops};
    int64_t g = T.coff[n_tensors];
    if (g > 4 * KGW_GRID) g = 4 * KGW_GRID;
    if (g < 1) g = 1;
    k_adam_fused<true><<<(int)g, 256, 0, (hipStream_t)stream_>>>(T, Z, step_dev, lr, beta1, beta2, eps, weight_decay);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

// The gradients of a multi-GPU step on their way into the all-reduce bucket: dst[i] = the finished gradient of tensor i -- a copy of
// grads[i] where that is complete, the sum of its producer's partial records where src[i] says so (also stored into grads[i]) --
// in ONE launch: k_adam_fused's work units without the update.
extern "C" int kgw_grad_finish(int32_t n_tensors, float* const* dst, float* const* grads, const int64_t* numel, const KgwGradSrc* src,
                               kgw_stream_t stream_) {
    if (n_tensors < 0 || n_tensors > KGW_ADAM_FUSED_MAX) return KGW_E_RANGE;
    if (n_tensors == 0) return KGW_OK;
    if (!dst || !grads || !numel) return KGW_E_NULL;
    AdamFTab T;
    const int rc = adam_fused_table(T, n_tensors, dst, grads, dst, dst, numel, src);
    if (rc != KGW_OK) return rc;
    for (int i = 0; i < n_tensors; ++i)
        if (src && src[i].kind == KGW_GRAD_G3T && src[i].packed) return KGW_E_UNSUPPORTED;      // (no update, no image)
    AdamTail Z{};
    int64_t g = T.coff[n_tensors];
    if (g > 4 * KGW_GRID) g = 4 * KGW_GRID;
    k_adam_fused<false><<<(int)g, 256, 0, (hipStream_t)stream_>>>(T, Z, nullptr, 0.f, 0.f, 0.f, 0.f, 0.f);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

static int adam_fused_table(AdamFTab& T, int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                            float* const* exp_avg_sq, const int64_t* numel, const KgwGradSrc* src) {
    T.n = n_tensors;
    T.off[0] = 0;
    T.coff[0] = 0;
    int nsrc = 0;
    for (int i = 0; i < n_tensors; ++i) {
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || numel[i] < 0) return KGW_E_NULL;
        T.p[i] = params[i]; T.g[i] = grads[i]; T.m[i] = exp_avg[i]; T.v[i] = exp_avg_sq[i];
        T.src_of[i] = 255;
        int64_t per = 1024;
        if (src && src[i].kind != KGW_GRAD_DIRECT) {
            const KgwGradSrc& S = src[i];
            if (nsrc >= KGW_ADAM_FUSED_SRC) return KGW_E_RANGE;
            if (!S.ws || S.nblk < 1) return KGW_E_NULL;
            // the record must describe exactly this tensor
            if (S.kind == KGW_GRAD_TN) {
                if (S.MT < 1 || S.NT < 1 || S.MT * S.NT > 16 || (256 % (32 * S.MT)) || (int64_t)S.M * S.N != numel[i]) return KGW_E_RANGE;
            } else if (S.kind == KGW_GRAD_TN_COLSUM) {
                if (S.MT < 1 || (256 % (32 * S.MT)) || S.M != numel[i]) return KGW_E_RANGE;
            } else if (S.kind == KGW_GRAD_MLP2_W) {
                if (S.K1 < 1 || S.K1 > 31 || (int64_t)128 * S.K1 != numel[i]) return KGW_E_RANGE;
            } else if (S.kind == KGW_GRAD_MLP2_B) {
                if (S.K1 < 0 || S.K1 > 31 || numel[i] != 128) return KGW_E_RANGE;
            } else if (S.kind == KGW_GRAD_G3T) {
                if (S.M < 32 || (S.M & 31) || (int64_t)128 * S.M != numel[i] || S.flip < 0 || (S.flip & (S.flip - 1))) return KGW_E_RANGE;
                if (((uintptr_t)S.ws | (uintptr_t)S.packed) & 15) return KGW_E_UNSUPPORTED;
            } else {
                return KGW_E_RANGE;
            }
            T.src[nsrc] = S;
            T.src_of[i] = (unsigned char)nsrc++;
            per = S.kind == KGW_GRAD_G3T ? 1024 : 64;          // (a 32 x 32 tile per unit)
        }
        T.off[i + 1] = T.off[i] + numel[i];
        T.coff[i + 1] = T.coff[i] + (numel[i] + per - 1) / per;
        T.vec[i] = (((uintptr_t)params[i] | (uintptr_t)grads[i] | (uintptr_t)exp_avg[i] | (uintptr_t)exp_avg_sq[i]) & 15) == 0;
    }
    return KGW_OK;
}
