// kgw_fold.hip -- FC_output of the feature MLPs folded into the layer-1 relation parameters (gfx950).
//
// A node's feature-MLP output H = h2 T + c (T = FC_output.weight^T, c = FC_output.bias; kgwas/model.py:15,21) enters the
// first GATConv layer only linearly -- as the message sum_j alpha_ij H_j and through the logit projections <H_j, u_r>,
// <H_i, v_r> (kgwas/conv.py:150-152,227-228; GATConv has no root term).  So layer 1 can run on the hidden state h2 with
//      U'_r  = T_src U_r                 V'_r = T_dst V_r                 kappa_r = <c_src, U_r> + <c_dst, V_r>
//      W'_r  = T_src W_r^T               gamma_r = c_src W_r^T            (the packed [in, out] weights)
// (src / dst = the MLP of the relation's source / destination node type), which removes a 128 x 128 Linear -- forward, dX and
// dW -- over EVERY sampled node (~160 k rows per 512-seed batch) for a few 128^3 products per relation.  Exact: the same
// sums re-associated, like aggregate-then-transform.
//
// kgw_fold_fwd:  one launch.  Blocks [0, 4 n): the n products W'_i = T W_i^T on fp32 MFMA, one wavefront per 32 x 32 tile
//                (operands straight from global memory, both coalesced: the k order inside the product is permuted like
//                in kgw_linear_splitk).  Blocks [4 n, 4 n + n_rels): one per relation id -- U', V', kappa (zeros for
//                relations outside the pack) and gamma.
// kgw_fold_bwd:  one launch, every output element written, fixed summation orders (deterministic):
//                blocks A [0, 4 n):          d W_i^T (fold part) = T^T dW'_i + c (x) dgamma_i            (MFMA tiles)
//                blocks B [.., + 16 n_mlp):  d FC_output.weight_m = sum over the relations whose source MLP is m of
//                                            (dW'_i W_i)^T, eight wavefronts share a tile's relations and add their
//                                            accumulators through LDS in wavefront order, + the rank-1 terms dU'_i (x) U_i,
//                                            dV'_i (x) V_i; the blocks of column tile 0 also produce d FC_output.bias_m (the
//                                            A operand w_i is already in their registers)
//                blocks C [.., + n_rels):    dU_r = T^T dU'_r + dkappa_r c_src,  dV_r likewise   (then kgw_relvec_bwd)
#include "kgw_common.h"
#include <cstdlib>
#include "kgw_fold_common.h"

namespace {

__device__ __forceinline__ float block128_sum(float v, float* sm) {      // sum over threads 0..127 of a 256-thread block
    const int t = threadIdx.x;
    if (t < 128) sm[t] = v;
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
        if (t < o) sm[t] += sm[t + o];
        __syncthreads();
    }
    return sm[0];
}

__global__ void __launch_bounds__(256) k_fold_fwd(FoldTab T, FoldPtrs P) {
    __shared__ float sm[128];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5, wave = threadIdx.x >> 6;
    if (b < 4 * T.n) {
        const int i = b >> 2, tile = (b & 3) * 4 + wave, tm = tile >> 2, tn = tile & 3;
        const float* fw = P.fcw[T.src_m[i]];
        const float* w = P.w_src_t + (int64_t)i * FC * FC;
        // W'[k][o] = sum_c T[k][c] w[c][o];  A(m = k, c) = fcw[c][k], B(c, n = o) = w[c][o]
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        tile_mma<false, false>(fw + 32 * tm, 1, FC, w + 32 * tn, FC, 1, li, lk, acc0, acc1);
        float* out = P.Wp + (int64_t)i * FC * FC + (int64_t)(32 * tm) * FC + 32 * tn + li;
#pragma unroll
        for (int e = 0; e < 16; ++e) out[((e & 3) + 8 * (e >> 2) + 4 * lk) * FC] = acc0[e] + acc1[e];
        return;
    }
    const int r = b - 4 * T.n;
    const int i = T.live_of[r];
    const int t = threadIdx.x;
    if (i < 0) {                                   // relation outside the pack: zero rows (the aggregate reads by relation id)
        if (t < 128) { P.Up[r * FC + t] = 0.f; P.Vp[r * FC + t] = 0.f; }
        if (t == 0) P.kappa[r] = 0.f;
        return;
    }
    const int ms = T.src_m[i], md = T.dst_m[i];
    const float* u = P.U + r * FC;
    const float* v = P.V + r * FC;
    float kp = 0.f;
    if (t < 128) {
        const float* ws = P.fcw[ms];
        const float* wd = P.fcw[md];
        const float* cb = P.fcb[ms];
        const float* w = P.w_src_t + (int64_t)i * FC * FC;
        float up = 0.f, vp = 0.f, gm = 0.f;
        for (int c = 0; c < FC; ++c) {
            up = fmaf(ws[c * FC + t], u[c], up);          // U'[k = t] = sum_c T[k][c] U[c]
            vp = fmaf(wd[c * FC + t], v[c], vp);
            gm = fmaf(cb[c], w[c * FC + t], gm);          // gamma[o = t] = sum_c c[c] w[c][o]
        }
        P.Up[r * FC + t] = up; P.Vp[r * FC + t] = vp; P.gamma[i * FC + t] = gm;
        kp = fmaf(P.fcb[ms][t], u[t], P.fcb[md][t] * v[t]);      // (written out: the riders' copy, kgw_riders.h, rounds the same way)
    }
    const float ksum = block128_sum(kp, sm);
    if (t == 0) P.kappa[r] = ksum;
}

__global__ void __launch_bounds__(512) k_fold_bwd(FoldTab T, FoldPtrs P, int parts) {
    __shared__ float red[8][32 * 32];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5, wave = threadIdx.x >> 6;
    const int t = threadIdx.x;
    const int nA = 2 * T.n, nB = 16 * T.n_mlp, nC = T.n_rels;
    if (b < nA) {
        if (!(parts & 1)) return;
        // ---- A: dws[i][h][o] = sum_k T[k][h] dW'[k][o] + c[h] dgamma[o];  A(m = h, k) = fcw[h][k], B(k, n = o) = dW'[k][o]
        const int i = b >> 1, tile = (b & 1) * 8 + wave, tm = tile >> 2, tn = tile & 3;
        const int ms = T.src_m[i];
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        tile_mma<true, false>(P.fcw[ms] + (int64_t)(32 * tm) * FC, FC, 1, P.dWp + (int64_t)i * FC * FC + 32 * tn, FC, 1, li, lk, acc0, acc1);
        const float dg = P.dgamma[i * FC + 32 * tn + li];
        float* out = P.dws + (int64_t)i * FC * FC + (int64_t)(32 * tm) * FC + 32 * tn + li;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * lk;
            out[row * FC] = fmaf(P.fcb[ms][32 * tm + row], dg, acc0[e] + acc1[e]);      // (written out: k_param_tail's copy rounds the same way)
        }
        return;
    }
    if (b < nA + nB) {
        if (!(parts & 2)) return;
        // ---- B: dfcw_m[h][k] = sum_{i: src_m = m} ( sum_o w_i[h][o] dW'_i[k][o] + U_i[h] dU'_i[k] ) + sum_{i: dst_m = m} V_i[h] dV'_i[k]
        //         (column-tile 0 also:) dfcb_m[h] = sum_{i: src_m = m} ( sum_o w_i[h][o] dgamma_i[o] + dkappa_i U_i[h] )
        //                                          + sum_{i: dst_m = m} dkappa_i V_i[h]
        const int q = b - nA, m = q >> 4, tile = q & 15, tm = tile >> 2, tn = tile & 3;
        // d U'_i[k], d V'_i[k] of the tile's 32 columns for every relation, once per block (requested here, read by the rank-1 loop
        // after the products): with KGW_F_DUV_PIECES each value is eight pieces -- added on the way in, in k_duv_fold's order
        __shared__ float dus[KGW_MAX_RELS][32], dvs[KGW_MAX_RELS][32];
        for (int idx = t; idx < T.n * 32; idx += 512) {
            const int i = idx >> 5, c = idx & 31, r = T.rel_id[i], kk = 32 * tn + c;
            dus[i][c] = P.duv_pieces ? kgw_duv_sum8(P.dUp + r * 8 * FC + kk) : P.dUp[r * FC + kk];
            dvs[i][c] = P.duv_pieces ? kgw_duv_sum8(P.dVp + r * 8 * FC + kk) : P.dVp[r * FC + kk];
        }
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        float qb = 0.f;                                    // this lane's share of sum_o w_i[h = 32 tm + li][o] dgamma_i[o]
        int seen = 0;
        for (int i = 0; i < T.n; ++i) {                    // the m-th MLP's relations, dealt to the 8 wavefronts in order
            if (T.src_m[i] != m) continue;
            if ((seen++ & 7) != wave) continue;
            // A(m = h, o) = w_i[h][o], B(o, n = k) = dW'_i[k][o]: both contiguous along the contraction index o
            float av[64], bv[64];
            const float4* qa = (const float4*)(P.w_src_t + (int64_t)i * FC * FC + (int64_t)(32 * tm + li) * FC + 64 * lk);
            const float4* qk = (const float4*)(P.dWp + (int64_t)i * FC * FC + (int64_t)(32 * tn + li) * FC + 64 * lk);
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                const float4 v = qa[x]; av[4 * x] = v.x; av[4 * x + 1] = v.y; av[4 * x + 2] = v.z; av[4 * x + 3] = v.w;
                const float4 u = qk[x]; bv[4 * x] = u.x; bv[4 * x + 1] = u.y; bv[4 * x + 2] = u.z; bv[4 * x + 3] = u.w;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 64; j += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j + 1], bv[j + 1], acc1, 0, 0, 0);
            }
            if (tn == 0) {                                 // the bias gradient reuses the A operand already in registers
                const float4* dg = (const float4*)(P.dgamma + i * FC + 64 * lk);
                float s = 0.f;
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const float4 g4 = dg[x];
                    s = fmaf(av[4 * x], g4.x, s); s = fmaf(av[4 * x + 1], g4.y, s);
                    s = fmaf(av[4 * x + 2], g4.z, s); s = fmaf(av[4 * x + 3], g4.w, s);
                }
                qb += s;
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave][((e & 3) + 8 * (e >> 2) + 4 * lk) * 32 + li] = acc0[e] + acc1[e];
        __shared__ float redb[8][64];
        redb[wave][lane] = qb;
        __syncthreads();
        const int col = t & 31;                            // k inside the tile
        float vs[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int row = (t >> 5) + 16 * h2;            // h inside the tile
            float vsum = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) vsum += red[w8][row * 32 + col];
            vs[h2] = vsum;
        }
        const int k = 32 * tn + col, h0 = 32 * tm + (t >> 5);
#pragma unroll 8
        for (int i = 0; i < T.n; ++i) {                    // rank-1 terms, relation order; loads unconditional (independent)
            const int r = T.rel_id[i];
            const float fs = T.src_m[i] == m ? 1.f : 0.f, fd = T.dst_m[i] == m ? 1.f : 0.f;
            const float du = dus[i][col] * fs, dv = dvs[i][col] * fd;
            vs[0] = fmaf(P.U[r * FC + h0], du, vs[0]);      vs[0] = fmaf(P.V[r * FC + h0], dv, vs[0]);
            vs[1] = fmaf(P.U[r * FC + h0 + 16], du, vs[1]); vs[1] = fmaf(P.V[r * FC + h0 + 16], dv, vs[1]);
        }
        P.dfcw[m][h0 * FC + k] = vs[0];
        P.dfcw[m][(h0 + 16) * FC + k] = vs[1];
        if (tn == 0 && t < 32) {                           // bias gradient of rows h = 32 tm + t
            float s = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) s += redb[w8][t] + redb[w8][32 + t];      // the two k halves of every wavefront
            const int h = 32 * tm + t;
#pragma unroll 8
            for (int i = 0; i < T.n; ++i) {
                const int r = T.rel_id[i];
                const float dk = P.dkappa[r];
                s = fmaf(P.U[r * FC + h], T.src_m[i] == m ? dk : 0.f, s);
                s = fmaf(P.V[r * FC + h], T.dst_m[i] == m ? dk : 0.f, s);
            }
            P.dfcb[m][h] = s;
        }
        return;
    }
    {
        // ---- C: dU[r][c] = sum_k T[k][c] dU'[r][k] + dkappa_r c_src[c]   (fcw[c][k]: thread c reads its own row)
        if (!(parts & 4)) return;
        // (round 4: the rows of FC_output.weight are read ROW by row, a wavefront per 16 rows, two floats per lane -- 512 contiguous
        //  bytes per load, 32 loads in flight, one wave-wide sum per row; with one thread per row every load touched 64 rows)
        const int r = b - nA - nB;
        const int i = T.live_of[r];
        if (i < 0) {
            if (t < 128) { P.dU[r * FC + t] = 0.f; P.dV[r * FC + t] = 0.f; }
            return;
        }
        const int ms = T.src_m[i], md = T.dst_m[i];
        float2 du2, dv2;
        if (P.duv_pieces) {
            du2 = make_float2(kgw_duv_sum8(P.dUp + r * 8 * FC + 2 * lane), kgw_duv_sum8(P.dUp + r * 8 * FC + 2 * lane + 1));
            dv2 = make_float2(kgw_duv_sum8(P.dVp + r * 8 * FC + 2 * lane), kgw_duv_sum8(P.dVp + r * 8 * FC + 2 * lane + 1));
        } else {
            du2 = ((const float2*)(P.dUp + r * FC))[lane]; dv2 = ((const float2*)(P.dVp + r * FC))[lane];
        }
        const float dk = P.dkappa[r];
        const int c0 = wave * 16;
        float2 a[16], bq[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            a[q] = ((const float2*)(P.fcw[ms] + (c0 + q) * FC))[lane];
            bq[q] = ((const float2*)(P.fcw[md] + (c0 + q) * FC))[lane];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float su = kgw_wave_allsum(fmaf(a[q].x, du2.x, a[q].y * du2.y));
            const float sv = kgw_wave_allsum(fmaf(bq[q].x, dv2.x, bq[q].y * dv2.y));
            if (lane == 0) {
                P.dU[r * FC + c0 + q] = fmaf(dk, P.fcb[ms][c0 + q], su);
                P.dV[r * FC + c0 + q] = fmaf(dk, P.fcb[md][c0 + q], sv);
            }
        }
    }
}

}  // namespace

extern "C" int kgw_fold_fwd(const KgwFoldArgs* a, kgw_stream_t stream_) {
    FoldTab T; FoldPtrs P;
    int rc = build(a, &T, &P);
    if (rc) return rc;
    if (!P.Up || !P.Vp || !P.kappa || !P.Wp || !P.gamma) return KGW_E_NULL;
    k_fold_fwd<<<4 * T.n + T.n_rels, 256, 0, (hipStream_t)stream_>>>(T, P);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_fold_bwd(const KgwFoldArgs* a, kgw_stream_t stream_) {
    FoldTab T; FoldPtrs P;
    int rc = build(a, &T, &P);
    if (rc) return rc;
    if (!P.dUp || !P.dVp || !P.dkappa || !P.dWp || !P.dgamma || !P.dU || !P.dV || !P.dws) return KGW_E_NULL;
    for (int m = 0; m < T.n_mlp; ++m)
        if (!P.dfcw[m] || !P.dfcb[m]) return KGW_E_NULL;
    const int parts = 15;                             // (all four block classes)
    k_fold_bwd<<<2 * T.n + 16 * T.n_mlp + T.n_rels, 512, 0, (hipStream_t)stream_>>>(T, P, parts);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}
