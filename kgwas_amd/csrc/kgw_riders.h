// kgw_riders.h -- parameter-only forward work of a training step as RIDER blocks of the first gene Linear's kgw_gemm3 launch.
//
// The forward k_g3_gemm of the benchmark graph is 240 blocks of one CU each on a 256-CU chip, for ~145 us: sixteen compute units
// idle.  Measured (round 5, KGW_G3_SPIN_*): sixteen extra blocks that only wait 100 us cost the step 3 - 8 us for both of its
// kgw_gemm3 launches together, thirty-two cost their whole duration -- exactly the idle CUs are free.  What runs there now is
// the work of a step that depends on the PARAMETERS only and used to sit on the step's critical path as two latency-bound
// launches of its own (k_relvec_fwd 9.4 us + k_fold_fwd 9.8 us on the benchmark):
//   * the attention vectors of every relation of every layer, u_r = W_src^T att_src, v_r = W_dst^T att_dst (kgwas/conv.py:138-151;
//     kgw_relvec_fwd_multi), the summed biases of the layers' destination blocks, the zero fill of the aggregates' workspaces;
//   * FC_output folded into the layer-1 relation parameters (kgw_fold_fwd: U', V', kappa, W', gamma; kgwas/model.py:15,21).
// Every task is ONE WAVEFRONT's (no block barrier, any number of rider blocks): a relation's two 128 x 128 slabs row by row with
// the arithmetic of k_relvec_fwd (two floats per lane, one wave-wide sum per row), then -- the vectors still in registers -- the
// relation's fold vectors with the arithmetic of k_fold_fwd (one output column per "thread", i.e. two per lane, the same fmaf
// chains; kappa's 128-term tree as shuffles); a 32 x 32 tile of W' = T W^T per task through the same tile_mma.  The values are
// bit-identical to the stand-alone launches' (tests/test_gpu_riders.py).
#pragma once
#include "kgw_fold_common.h"

namespace {

struct RiderRv {                    // one layer's kgw_relvec_fwd job (v_by_rel = 1)
    int NR, n_live, n_blk, fold;    // fold != 0: the layer whose relations the FC_output fold takes
    const int32_t* live_of_rel; const int32_t* bip_pos;
    const float* wsT; const float* wdT; const float* att_src; const float* att_dst;
    float* U_full; float* V; const float* bias; const int32_t* blk_of_live; float* bsum; float* zero_buf; int64_t zero_f4;
};

struct G3Riders {
    int n_blocks;                   // rider blocks of the launch (0: none)
    int n_rv;                       // relvec jobs (layers)
    int has_fold, pad_;
    RiderRv rv[KGW_MAX_LAYERS];
    FoldTab FT; FoldPtrs FP;
};

// fold vectors of relation id r (packed slot i >= 0) from u, v held as (lane -> columns lane, lane + 64)
__device__ __forceinline__ void rider_fold_vectors(const FoldTab& T, const FoldPtrs& P, int r, int i, float u_lo, float u_hi,
                                                   float v_lo, float v_hi, int lane) {
    const int ms = T.src_m[i], md = T.dst_m[i];
    const float* __restrict__ ws = P.fcw[ms];
    const float* __restrict__ wd = P.fcw[md];
    const float* __restrict__ cb = P.fcb[ms];
    const float* __restrict__ w = P.w_src_t + (int64_t)i * FC * FC;
    float up[2] = {0.f, 0.f}, vp[2] = {0.f, 0.f}, gm[2] = {0.f, 0.f};
#pragma unroll 4
    for (int c = 0; c < FC; ++c) {
        const float uc = __shfl(c < 64 ? u_lo : u_hi, c & 63, 64);
        const float vc = __shfl(c < 64 ? v_lo : v_hi, c & 63, 64);
        const float cc = cb[c];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = lane + 64 * h;
            up[h] = fmaf(ws[c * FC + t], uc, up[h]);          // U'[k = t] = sum_c T[k][c] U[c]
            vp[h] = fmaf(wd[c * FC + t], vc, vp[h]);
            gm[h] = fmaf(cc, w[c * FC + t], gm[h]);           // gamma[o = t] = sum_c c[c] w[c][o]
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = lane + 64 * h;
        P.Up[r * FC + t] = up[h]; P.Vp[r * FC + t] = vp[h]; P.gamma[i * FC + t] = gm[h];
    }
    // kappa = sum_t fcb_src[t] u[t] + fcb_dst[t] v[t], added in block128_sum's tree: (t, t + 64), then t + 32, 16, 8, 4, 2, 1
    const float kp_lo = fmaf(P.fcb[ms][lane], u_lo, P.fcb[md][lane] * v_lo);
    const float kp_hi = fmaf(P.fcb[ms][lane + 64], u_hi, P.fcb[md][lane + 64] * v_hi);
    float kp = kp_lo + kp_hi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kp += __shfl_down(kp, o, 64);
    if (lane == 0) P.kappa[r] = kp;
}

// relation id r of layer job J: u_r, v_r (k_relvec_fwd's arithmetic), then its fold vectors
__device__ __forceinline__ void rider_relation(const G3Riders& R, const RiderRv& J, int r, int lane) {
    const int i = J.live_of_rel[r];
    const bool fold = J.fold && R.has_fold;
    if (i < 0) {
        for (int h = 0; h < 2; ++h) {
            J.U_full[(int64_t)r * KGW_C + lane + 64 * h] = 0.f;
            J.V[(int64_t)r * KGW_C + lane + 64 * h] = 0.f;
        }
        if (fold && R.FT.live_of[r] < 0) {
            for (int h = 0; h < 2; ++h) { R.FP.Up[r * FC + lane + 64 * h] = 0.f; R.FP.Vp[r * FC + lane + 64 * h] = 0.f; }
            if (lane == 0) R.FP.kappa[r] = 0.f;
        }
        return;
    }
    const float2 as2 = ((const float2*)(J.att_src + (int64_t)i * KGW_C))[lane];
    const float2 ad2 = ((const float2*)(J.att_dst + (int64_t)i * KGW_C))[lane];
    const int j = J.bip_pos[i];
    const float* ws = J.wsT + (int64_t)i * KGW_C * KGW_C;
    const float* wd = j >= 0 ? J.wdT + (int64_t)j * KGW_C * KGW_C : ws;
    float u_lo = 0.f, u_hi = 0.f, v_lo = 0.f, v_hi = 0.f;        // lane L keeps rows L and L + 64
    for (int rb = 0; rb < 16; ++rb) {                             // (a wavefront of k_relvec_fwd = 8 rows)
        float2 a[8], b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            a[q] = ((const float2*)(ws + (rb * 8 + q) * KGW_C))[lane];
            b[q] = ((const float2*)(wd + (rb * 8 + q) * KGW_C))[lane];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float u = kgw_wave_allsum(fmaf(a[q].x, as2.x, a[q].y * as2.y));
            const float v = kgw_wave_allsum(fmaf(b[q].x, ad2.x, b[q].y * ad2.y));
            const int row = rb * 8 + q;
            if (lane == (row & 63)) {
                if (row < 64) { u_lo = u; v_lo = v; } else { u_hi = u; v_hi = v; }
            }
        }
    }
    J.U_full[(int64_t)r * KGW_C + lane] = u_lo; J.U_full[(int64_t)r * KGW_C + lane + 64] = u_hi;
    J.V[(int64_t)r * KGW_C + lane] = v_lo; J.V[(int64_t)r * KGW_C + lane + 64] = v_hi;
    if (fold) {
        const int fi = R.FT.live_of[r];
        if (fi >= 0) rider_fold_vectors(R.FT, R.FP, r, fi, u_lo, u_hi, v_lo, v_hi, lane);
        else {
            for (int h = 0; h < 2; ++h) { R.FP.Up[r * FC + lane + 64 * h] = 0.f; R.FP.Vp[r * FC + lane + 64 * h] = 0.f; }
            if (lane == 0) R.FP.kappa[r] = 0.f;
        }
    }
}

// summed bias of every destination block of a layer (k_relvec_fwd's extra block), two columns per lane
__device__ __forceinline__ void rider_bias_sums(const RiderRv& J, int lane) {
    for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        float acc[KGW_MAX_TYPES];
#pragma unroll
        for (int b = 0; b < KGW_MAX_TYPES; ++b) acc[b] = 0.f;
#pragma unroll 8
        for (int i = 0; i < J.n_live; ++i) {
            const float v = J.bias[(int64_t)i * KGW_C + k];
            const int bi = J.blk_of_live[i];
#pragma unroll
            for (int b = 0; b < KGW_MAX_TYPES; ++b) acc[b] += (bi == b) ? v : 0.f;
        }
#pragma unroll
        for (int b = 0; b < KGW_MAX_TYPES; ++b)
            if (b < J.n_blk) J.bsum[(int64_t)b * KGW_C + k] = acc[b];
    }
}

// W'_i tile (tm, tn) = T W_i^T on one wavefront: k_fold_fwd's first kind of block
__device__ __forceinline__ void rider_fold_tile(const FoldTab& T, const FoldPtrs& P, int i, int tile, int lane) {
    const int li = lane & 31, lk = lane >> 5, tm = tile >> 2, tn = tile & 3;
    const float* fw = P.fcw[T.src_m[i]];
    const float* w = P.w_src_t + (int64_t)i * FC * FC;
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    tile_mma<false, false>(fw + 32 * tm, 1, FC, w + 32 * tn, FC, 1, li, lk, acc0, acc1);
    float* out = P.Wp + (int64_t)i * FC * FC + (int64_t)(32 * tm) * FC + 32 * tn + li;
#pragma unroll
    for (int e = 0; e < 16; ++e) out[((e & 3) + 8 * (e >> 2) + 4 * lk) * FC] = acc0[e] + acc1[e];
}

// rider block rb of R.n_blocks, NW wavefronts: the tasks are dealt to the launch's rider wavefronts in the order
// [relations of every layer (the long ones), bias sums, W' tiles], then everybody clears the aggregates' workspaces
template <int NW>
__device__ __attribute__((noinline)) void g3_param_riders(const G3Riders& R, int rb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwv = R.n_blocks * NW, w0 = rb * NW + wave;
    int n_rel = 0;
    for (int q = 0; q < R.n_rv; ++q) n_rel += R.rv[q].NR;
    const int n_bias = R.n_rv, n_tiles = R.has_fold ? 16 * R.FT.n : 0;
    const int total = n_rel + n_bias + n_tiles;
    for (int t = w0; t < total; t += nwv) {
        if (t < n_rel) {
            int q = 0, r = t;
            while (r >= R.rv[q].NR) { r -= R.rv[q].NR; ++q; }
            rider_relation(R, R.rv[q], r, lane);
        } else if (t < n_rel + n_bias) {
            const RiderRv& J = R.rv[t - n_rel];
            if (J.n_blk > 0) rider_bias_sums(J, lane);
        } else {
            const int k = t - n_rel - n_bias;
            rider_fold_tile(R.FT, R.FP, k >> 4, k & 15, lane);
        }
    }
    const int64_t nthr = (int64_t)R.n_blocks * NW * 64, gt = (int64_t)rb * NW * 64 + threadIdx.x;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < R.n_rv; ++q) {
        float4* zb = (float4*)R.rv[q].zero_buf;
        for (int64_t i = gt; i < R.rv[q].zero_f4; i += nthr) zb[i] = z4;
    }
}

// host: the ABI's job records -> G3Riders (same argument checks as kgw_relvec_fwd_multi / kgw_fold_fwd)
inline int g3_riders_build(int n_rv, const KgwRelvecJob* jobs, const KgwFoldArgs* fold, int fold_job, int n_blocks, G3Riders* R) {
    *R = G3Riders{};
    if (n_rv < 0 || n_rv > KGW_MAX_LAYERS || (n_rv && !jobs)) return KGW_E_RANGE;
    for (int q = 0; q < n_rv; ++q) {
        const KgwRelvecJob& D = jobs[q];
        if (D.n_rels_total <= 0 || D.n_rels_total > KGW_MAX_RELS) return KGW_E_RANGE;
        if (!D.live_of_rel || !D.bip_pos || !D.w_src_t || !D.att_src || !D.att_dst || !D.U_full || !D.V) return KGW_E_NULL;
        if (D.zero_buf && ((D.zero_floats & 3) || D.zero_floats < 0 || ((uintptr_t)D.zero_buf & 15))) return KGW_E_UNSUPPORTED;
        const bool with_bias = D.bias && D.blk_of_live && D.bias_sum && D.n_blk > 0 && D.n_blk <= KGW_MAX_TYPES;
        RiderRv& T = R->rv[q];
        T.NR = D.n_rels_total; T.n_live = D.n_live; T.n_blk = with_bias ? D.n_blk : 0; T.fold = (fold && q == fold_job) ? 1 : 0;
        T.live_of_rel = D.live_of_rel; T.bip_pos = D.bip_pos; T.wsT = D.w_src_t; T.wdT = D.w_dst_t; T.att_src = D.att_src;
        T.att_dst = D.att_dst; T.U_full = D.U_full; T.V = D.V; T.bias = D.bias; T.blk_of_live = D.blk_of_live; T.bsum = D.bias_sum;
        T.zero_buf = D.zero_buf; T.zero_f4 = D.zero_buf ? D.zero_floats / 4 : 0;
    }
    R->n_rv = n_rv;
    if (fold) {
        if (fold_job < 0 || fold_job >= n_rv) return KGW_E_RANGE;
        int rc = build(fold, &R->FT, &R->FP);
        if (rc) return rc;
        if (!R->FP.Up || !R->FP.Vp || !R->FP.kappa || !R->FP.Wp || !R->FP.gamma) return KGW_E_NULL;
        if (R->FT.n_rels != R->rv[fold_job].NR) return KGW_E_RANGE;
        // the fold reads u_r / v_r where this launch's relation task leaves them
        if (R->FP.U != R->rv[fold_job].U_full || R->FP.V != R->rv[fold_job].V) return KGW_E_UNSUPPORTED;
        R->has_fold = 1;
    }
    R->n_blocks = (n_rv || fold) ? n_blocks : 0;
    return KGW_OK;
}

}  // namespace
