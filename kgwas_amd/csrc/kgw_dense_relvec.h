// kgw_dense_relvec.h -- part of kgw_dense.hip (ONE translation unit, split by kernel family in round 6; include order matters:
// later families use device functions of earlier ones): attention vectors of a layer (kgw_relvec*) and the parameter-only end of the backward pass as one launch (kgw_param_tail).
#pragma once

// ======================================================================================================
// kgw_relvec: the attention vectors of every relation of a layer in one launch.
//   u_r = W_src^T att_src , v_r = W_dst^T att_dst (W_src^T att_dst for same-type relations)   conv.py:138-151
// Weights are stored transposed/packed: wT[i][k][c] = W_i[c][k].  Forward: U_full[r] (zeros for relations the
// layer does not compute) and V[i].  Backward: d wT, d att from (dU_full, dV).
// ======================================================================================================
namespace {

struct RvFwdJob {
    int NR, n_live, n_blk, n_main, blk0, nblk;
    const int32_t* live_of_rel; const int32_t* bip_pos;
    const float* wsT; const float* wdT; const float* att_src; const float* att_dst;
    float* U_full; float* V; const float* bias; const int32_t* blk_of_live; float* bsum; float* zero_buf; int64_t zero_f4;
};
struct RvFwdJobs { RvFwdJob j[KGW_MAX_LAYERS]; int n; };
struct RvBwdJob {
    int blk0, v_by_rel, pieces, pad_;
    const int32_t* rel_ids; const int32_t* bip_pos;
    const float* wsT; const float* wdT; const float* att_src; const float* att_dst; const float* dU_full; const float* dV;
    float* dwsT; float* dwdT; float* datt_src; float* datt_dst; const float* dws_acc;
};
struct RvBwdJobs { RvBwdJob j[KGW_MAX_LAYERS]; int blk_end; int n; };

// (round 4: 1 024 threads per block.  A relation's two 128 x 128 slabs are read ROW by row, a wavefront per 8 rows, two floats per
//  lane -- 512 contiguous bytes per load, 16 loads in flight, one wave-wide sum per row; with one thread per row every load touched
//  64 different rows and the 29-block launch took 12 - 18 us for 6 MB)
__global__ void __launch_bounds__(1024) k_relvec_fwd(RvFwdJobs J, int v_by_rel) {
    int jq = 0;
    while (jq + 1 < J.n && (int)blockIdx.x >= J.j[jq + 1].blk0) ++jq;
    const RvFwdJob& T = J.j[jq];
    const int NR = T.NR, n_live = T.n_live, n_blk = T.n_blk, n_main = T.n_main;
    const int32_t* __restrict__ live_of_rel = T.live_of_rel;
    const int32_t* __restrict__ bip_pos = T.bip_pos;
    const float* __restrict__ wsT = T.wsT;
    const float* __restrict__ wdT = T.wdT;
    const float* __restrict__ att_src = T.att_src;
    const float* __restrict__ att_dst = T.att_dst;
    float* __restrict__ U_full = T.U_full;
    float* __restrict__ V = T.V;
    const float* __restrict__ bias = T.bias;
    const int32_t* __restrict__ blk_of_live = T.blk_of_live;
    float* __restrict__ bsum = T.bsum;
    float* __restrict__ zero_buf = T.zero_buf;
    const int64_t zero_f4 = T.zero_f4;
    const int r = (int)blockIdx.x - T.blk0, k = threadIdx.x;
    if (r >= n_main) {      // extra blocks: clear the aggregate's workspace (Z, stat, d a_dst) in this launch
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t i = (int64_t)(r - n_main) * 1024 + k; i < zero_f4; i += (int64_t)(T.nblk - n_main) * 1024)
            ((float4*)zero_buf)[i] = z4;
        return;
    }
    if (r == NR) {          // extra block: bias of every relation into a destination type, summed in packed order
        if (k >= KGW_C) return;
        float acc[KGW_MAX_TYPES];
#pragma unroll
        for (int b = 0; b < KGW_MAX_TYPES; ++b) acc[b] = 0.f;
#pragma unroll 8
        for (int i = 0; i < n_live; ++i) {                 // independent loads: all in flight together
            const float v = bias[(int64_t)i * KGW_C + k];
            const int bi = blk_of_live[i];
#pragma unroll
            for (int b = 0; b < KGW_MAX_TYPES; ++b) acc[b] += (bi == b) ? v : 0.f;
        }
#pragma unroll
        for (int b = 0; b < KGW_MAX_TYPES; ++b)
            if (b < n_blk) bsum[(int64_t)b * KGW_C + k] = acc[b];
        return;
    }
    const int i = live_of_rel[r];
    if (i < 0) {
        if (k < KGW_C) {
            U_full[(int64_t)r * KGW_C + k] = 0.f;
            if (v_by_rel) V[(int64_t)r * KGW_C + k] = 0.f;
        }
        return;
    }
    const int lane = k & 63, wave = k >> 6;
    const float2 as2 = ((const float2*)(att_src + (int64_t)i * KGW_C))[lane];
    const float2 ad2 = ((const float2*)(att_dst + (int64_t)i * KGW_C))[lane];
    const int j = bip_pos[i];
    const float* ws = wsT + ((int64_t)i * KGW_C + wave * 8) * KGW_C;
    const float* wd = j >= 0 ? wdT + ((int64_t)j * KGW_C + wave * 8) * KGW_C : ws;
    float2 a[8], b[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        a[q] = ((const float2*)(ws + q * KGW_C))[lane];
        b[q] = ((const float2*)(wd + q * KGW_C))[lane];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float u = kgw_wave_allsum(fmaf(a[q].x, as2.x, a[q].y * as2.y));
        const float v = kgw_wave_allsum(fmaf(b[q].x, ad2.x, b[q].y * ad2.y));
        if (lane == 0) {
            const int row = wave * 8 + q;
            U_full[(int64_t)r * KGW_C + row] = u;
            V[(int64_t)(v_by_rel ? r : i) * KGW_C + row] = v;
        }
    }
}

// one block per live relation i; thread c owns column c of the [k][c] matrices
// One block per packed relation, 8 x 128 threads: thread (q, c) takes the rows k = q, q + 8, ... of the 128 x 128 weight
// slab (16 independent iterations, loads batched eight at a time -- with one thread per column the 128 iterations of
// dependent-latency loads made this 23-block launch take 14-38 us); the eight partial sums of d att are added in a fixed order.
__global__ void __launch_bounds__(1024) k_relvec_bwd(RvBwdJobs J) {
    int jq = 0;
    while (jq + 1 < J.n && (int)blockIdx.x >= J.j[jq + 1].blk0) ++jq;
    const RvBwdJob& T = J.j[jq];
    const int32_t* __restrict__ rel_ids = T.rel_ids;
    const int32_t* __restrict__ bip_pos = T.bip_pos;
    const float* __restrict__ wsT = T.wsT;
    const float* __restrict__ wdT = T.wdT;
    const float* __restrict__ att_src = T.att_src;
    const float* __restrict__ att_dst = T.att_dst;
    const float* __restrict__ dU_full = T.dU_full;
    const float* __restrict__ dV = T.dV;
    float* __restrict__ dwsT = T.dwsT;
    float* __restrict__ dwdT = T.dwdT;
    float* __restrict__ datt_src = T.datt_src;
    float* __restrict__ datt_dst = T.datt_dst;
    const float* __restrict__ dws_acc = T.dws_acc;
    const int v_by_rel = T.v_by_rel;
    // (round 4: FOUR blocks per relation, each owns 32 of the 128 columns -- the sums over k are per column, so the split needs no
    //  combine -- 52 relations then fill 208 CUs instead of 52; thread (q, c): rows k = q, q + 32, q + 64, q + 96)
    __shared__ float du[KGW_C], dv[KGW_C];
    __shared__ float ps[32][33], pd[32][33];
    const int bx = (int)blockIdx.x - T.blk0;
    const int i = bx >> 2, cl = threadIdx.x & 31, c = (bx & 3) * 32 + cl, q = threadIdx.x >> 5;
    const int r = rel_ids[i], j = bip_pos[i];
    if (threadIdx.x < KGW_C) {
        if (T.pieces) {             // (d u_r / d v_r as the aggregate's riders left them: eight pieces per value)
            du[threadIdx.x] = dU_full ? kgw_duv_sum8(dU_full + (int64_t)r * 8 * KGW_C + threadIdx.x) : 0.f;
            dv[threadIdx.x] = dV ? kgw_duv_sum8(dV + (int64_t)(v_by_rel ? r : i) * 8 * KGW_C + threadIdx.x) : 0.f;
        } else {
            du[threadIdx.x] = dU_full ? dU_full[(int64_t)r * KGW_C + threadIdx.x] : 0.f;
            dv[threadIdx.x] = dV ? dV[(int64_t)(v_by_rel ? r : i) * KGW_C + threadIdx.x] : 0.f;
        }
    }
    __syncthreads();
    const float as = att_src[(int64_t)i * KGW_C + c], ad = att_dst[(int64_t)i * KGW_C + c];
    const float* ws = wsT + (int64_t)i * KGW_C * KGW_C;
    float* dws = dwsT + (int64_t)i * KGW_C * KGW_C;
    // dws_acc: a gradient of w_src_t that arrived by another path (the layer's transform / the FC_output fold), added here
    // instead of by a separate framework launch
    const float* acc = dws_acc ? dws_acc + (int64_t)i * KGW_C * KGW_C : nullptr;
    const float* wd = j >= 0 ? wdT + (int64_t)j * KGW_C * KGW_C : nullptr;
    float* dwd = j >= 0 ? dwdT + (int64_t)j * KGW_C * KGW_C : nullptr;
    float gs = 0.f, gd = 0.f;
    float w[4], w2[4], a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = q + 32 * t;
        w[t] = ws[k * KGW_C + c];
        w2[t] = wd ? wd[k * KGW_C + c] : w[t];
        a[t] = acc ? acc[k * KGW_C + c] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = q + 32 * t;
        gs = fmaf(w[t], du[k], gs);
        gd = fmaf(w2[t], dv[k], gd);
        if (wd) {
            dws[k * KGW_C + c] = fmaf(du[k], as, a[t]);
            dwd[k * KGW_C + c] = dv[k] * ad;
        } else {
            dws[k * KGW_C + c] = fmaf(du[k], as, dv[k] * ad) + a[t];
        }
    }
    ps[q][cl] = gs; pd[q][cl] = gd;
    __syncthreads();
    if (q == 0) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
        for (int g = 0; g < 32; g += 4) {
            s0 += ps[g][cl]; s1 += ps[g + 1][cl]; s2 += ps[g + 2][cl]; s3 += ps[g + 3][cl];
            d0 += pd[g][cl]; d1 += pd[g + 1][cl]; d2 += pd[g + 2][cl]; d3 += pd[g + 3][cl];
        }
        datt_src[(int64_t)i * KGW_C + c] = (s0 + s1) + (s2 + s3);
        datt_dst[(int64_t)i * KGW_C + c] = (d0 + d1) + (d2 + d3);
    }
}

}  // namespace

namespace {
int relvec_fwd_launch(int n_jobs, const KgwRelvecJob* jobs, int v_by_rel, hipStream_t st) {
    RvFwdJobs J{};
    int blk = 0, n = 0;
    for (int q = 0; q < n_jobs; ++q) {
        const KgwRelvecJob& D = jobs[q];
        if (D.n_rels_total <= 0) continue;
        if (!D.live_of_rel || !D.bip_pos || !D.w_src_t || !D.att_src || !D.att_dst || !D.U_full || !D.V) return KGW_E_NULL;
        if (D.zero_buf && ((D.zero_floats & 3) || D.zero_floats < 0 || !aligned16(D.zero_buf))) return KGW_E_UNSUPPORTED;
        const bool with_bias = D.bias && D.blk_of_live && D.bias_sum && D.n_blk > 0 && D.n_blk <= KGW_MAX_TYPES;
        RvFwdJob& T = J.j[n++];
        T.NR = D.n_rels_total; T.n_live = D.n_live; T.n_blk = with_bias ? D.n_blk : 0;
        T.n_main = D.n_rels_total + (with_bias ? 1 : 0);
        T.zero_f4 = D.zero_buf ? D.zero_floats / 4 : 0;
        int64_t zblk = (T.zero_f4 + 1024 * 4 - 1) / (1024 * 4);        // ~4 float4 per thread
        if (zblk > 1024) zblk = 1024;
        T.blk0 = blk; T.nblk = T.n_main + (int)zblk;
        blk += T.nblk;
        T.live_of_rel = D.live_of_rel; T.bip_pos = D.bip_pos; T.wsT = D.w_src_t; T.wdT = D.w_dst_t; T.att_src = D.att_src;
        T.att_dst = D.att_dst; T.U_full = D.U_full; T.V = D.V; T.bias = D.bias; T.blk_of_live = D.blk_of_live; T.bsum = D.bias_sum;
        T.zero_buf = D.zero_buf;
    }
    if (n == 0) return KGW_OK;
    J.n = n;
    k_relvec_fwd<<<blk, 1024, 0, st>>>(J, v_by_rel);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

int relvec_bwd_launch(int n_jobs, const KgwRelvecJob* jobs, int v_by_rel, hipStream_t st) {
    RvBwdJobs J{};
    int blk = 0, n = 0;
    for (int q = 0; q < n_jobs; ++q) {
        const KgwRelvecJob& D = jobs[q];
        if (D.n_live <= 0) continue;
        if (!D.rel_ids || !D.bip_pos || !D.w_src_t || !D.att_src || !D.att_dst || !D.dw_src_t || !D.datt_src || !D.datt_dst)
            return KGW_E_NULL;
        RvBwdJob& T = J.j[n++];
        T.blk0 = blk; T.v_by_rel = v_by_rel; T.pieces = D.duv_pieces;
        blk += 4 * D.n_live;
        T.rel_ids = D.rel_ids; T.bip_pos = D.bip_pos; T.wsT = D.w_src_t; T.wdT = D.w_dst_t; T.att_src = D.att_src;
        T.att_dst = D.att_dst; T.dU_full = D.dU_full; T.dV = D.dV; T.dwsT = D.dw_src_t; T.dwdT = D.dw_dst_t;
        T.datt_src = D.datt_src; T.datt_dst = D.datt_dst; T.dws_acc = D.dw_src_acc;
    }
    if (n == 0) return KGW_OK;
    J.n = n; J.blk_end = blk;
    k_relvec_bwd<<<blk, 1024, 0, st>>>(J);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}
}  // namespace


// ======================================================================================================
// kgw_param_tail: the END of a captured step's backward pass as ONE launch (round 5).  What is left when the last layer-1 kernel
// has run is parameter-only work that nothing but the optimiser waits for:
//   the deferred weight-gradient products of the MLPs   (kgw_tn_gemm_multi_partial: k_tn_gemm<2,2>'s row blocks),
//   the backward of the FC_output fold                  (kgw_fold_bwd: 123 blocks, 28 us on its own),
//   the backward of the relation vectors of every layer (kgw_relvec_bwd_multi: 208 blocks, 9 us; the fold's layer reads the
//                                                        fold's d U, d V and d W share),
// three launches of which none fills the chip and each waits for the last block of the one before.  Here they are the blocks
// of one grid:
//   kind B  k_fold_bwd's B part (d FC_output.weight / .bias), a 256-thread block playing the 512-thread block's eight
//           wavefronts two at a time;
//   kind F  one block per (relation of the fold's layer, group of 32 columns): k_fold_bwd's C part for the relation (d U_r, d V_r
//           -- recomputed by each of the four column groups: 256 dot products), its A part for the group's four 32 x 32 tiles,
//           and k_relvec_bwd's block for the same columns on top of them, through LDS instead of through d U / d V / dws in HBM
//           (which are not written: nothing else reads them);
//   kind R  k_relvec_bwd's blocks of the other layers, 256 threads each;
//   kind T  tn_gemm_block<2,2>, as in k_tn_gemm.
// Every value is computed with the expressions, in the order, of the kernel it comes from: bit-identical results.
// ======================================================================================================
namespace {

struct TailIdx { int n_B, n_F, n_R, n_tn, fold_job, pad_; int tn_flat0[TN_MAX_JOBS + 1]; };
constexpr int TAIL_LDS_FLOATS = 8 * 1024 + 2 * KGW_MAX_RELS * 32 + 8 * 64;      // kind B: red | dus | dvs | redb

// k_relvec_bwd's block (relation slot i, columns 32 cg ..) on 256 threads: thread (qq, cl) plays the 1024-thread block's threads
// (qq + 8 qs, cl), qs = 0..3.  du / dv: d U_r / d V_r in LDS; acc_tile: the other gradient of w_src_t for these columns as a
// [128][33] LDS tile, or null (then T.dws_acc in HBM, or none)
__device__ __forceinline__ void relvec_bwd_cols256(const RvBwdJob& T, int i, int cg, const float* du, const float* dv,
                                                   const float* acc_tile, float* ps, float* pd) {
    const int cl = threadIdx.x & 31, c = cg * 32 + cl, qq = threadIdx.x >> 5;
    const int j = T.bip_pos[i];
    const float as = T.att_src[(int64_t)i * KGW_C + c], ad = T.att_dst[(int64_t)i * KGW_C + c];
    const float* __restrict__ ws = T.wsT + (int64_t)i * KGW_C * KGW_C;
    float* __restrict__ dws = T.dwsT + (int64_t)i * KGW_C * KGW_C;
    const float* __restrict__ acc = (!acc_tile && T.dws_acc) ? T.dws_acc + (int64_t)i * KGW_C * KGW_C : nullptr;
    const float* __restrict__ wd = j >= 0 ? T.wdT + (int64_t)j * KGW_C * KGW_C : nullptr;
    float* __restrict__ dwd = j >= 0 ? T.dwdT + (int64_t)j * KGW_C * KGW_C : nullptr;
#pragma unroll
    for (int qs = 0; qs < 4; ++qs) {
        const int q = qq + 8 * qs;
        float gs = 0.f, gd = 0.f;
        float w[4], w2[4], a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = q + 32 * t;
            w[t] = ws[k * KGW_C + c];
            w2[t] = wd ? wd[k * KGW_C + c] : w[t];
            a[t] = acc_tile ? acc_tile[k * 33 + cl] : (acc ? acc[k * KGW_C + c] : 0.f);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = q + 32 * t;
            gs = fmaf(w[t], du[k], gs);
            gd = fmaf(w2[t], dv[k], gd);
            if (wd) {
                dws[k * KGW_C + c] = fmaf(du[k], as, a[t]);
                dwd[k * KGW_C + c] = dv[k] * ad;
            } else {
                dws[k * KGW_C + c] = fmaf(du[k], as, dv[k] * ad) + a[t];
            }
        }
        ps[q * 33 + cl] = gs; pd[q * 33 + cl] = gd;
    }
    __syncthreads();
    if (qq == 0) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
        for (int g = 0; g < 32; g += 4) {
            s0 += ps[g * 33 + cl]; s1 += ps[(g + 1) * 33 + cl]; s2 += ps[(g + 2) * 33 + cl]; s3 += ps[(g + 3) * 33 + cl];
            d0 += pd[g * 33 + cl]; d1 += pd[(g + 1) * 33 + cl]; d2 += pd[(g + 2) * 33 + cl]; d3 += pd[(g + 3) * 33 + cl];
        }
        T.datt_src[(int64_t)i * KGW_C + c] = (s0 + s1) + (s2 + s3);
        T.datt_dst[(int64_t)i * KGW_C + c] = (d0 + d1) + (d2 + d3);
    }
}

// kind R
__device__ __forceinline__ void tail_relvec_block(const RvBwdJob& T, int bx, float* lds) {
    float* du = lds; float* dv = lds + KGW_C; float* ps = lds + 2 * KGW_C; float* pd = ps + 32 * 33;
    const int i = bx >> 2, r = T.rel_ids[i];
    if (threadIdx.x < KGW_C) {
        const int t = threadIdx.x;
        if (T.pieces) {
            du[t] = T.dU_full ? kgw_duv_sum8(T.dU_full + (int64_t)r * 8 * KGW_C + t) : 0.f;
            dv[t] = T.dV ? kgw_duv_sum8(T.dV + (int64_t)(T.v_by_rel ? r : i) * 8 * KGW_C + t) : 0.f;
        } else {
            du[t] = T.dU_full ? T.dU_full[(int64_t)r * KGW_C + t] : 0.f;
            dv[t] = T.dV ? T.dV[(int64_t)(T.v_by_rel ? r : i) * KGW_C + t] : 0.f;
        }
    }
    __syncthreads();
    relvec_bwd_cols256(T, i, bx & 3, du, dv, nullptr, ps, pd);
}

// kind F
__device__ __forceinline__ void tail_fold_rel_block(const FoldTab& T, const FoldPtrs& P, const RvBwdJob& J, int bx, float* lds) {
    float* du = lds; float* dv = lds + FC; float* tile = lds + 2 * FC;           // tile [128][33]
    float* ps = tile + FC * 33; float* pd = ps + 32 * 33;
    const int i = bx >> 2, cg = bx & 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lk = lane >> 5;
    const int r = T.rel_id[i], ms = T.src_m[i], md = T.dst_m[i];
    {   // k_fold_bwd C: d U_r[c] = <fcw_src[c][:], d U'_r> + d kappa_r fcb_src[c] (and V), the wavefront's 32 rows 16 at a time
        float2 du2, dv2;
        if (P.duv_pieces) {
            du2 = make_float2(kgw_duv_sum8(P.dUp + r * 8 * FC + 2 * lane), kgw_duv_sum8(P.dUp + r * 8 * FC + 2 * lane + 1));
            dv2 = make_float2(kgw_duv_sum8(P.dVp + r * 8 * FC + 2 * lane), kgw_duv_sum8(P.dVp + r * 8 * FC + 2 * lane + 1));
        } else {
            du2 = ((const float2*)(P.dUp + r * FC))[lane]; dv2 = ((const float2*)(P.dVp + r * FC))[lane];
        }
        const float dk = P.dkappa[r];
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int c0 = wave * 32 + half * 16;
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
                    du[c0 + q] = fmaf(dk, P.fcb[ms][c0 + q], su);
                    dv[c0 + q] = fmaf(dk, P.fcb[md][c0 + q], sv);
                }
            }
        }
    }
    {   // k_fold_bwd A: the tile (tm = wavefront, tn = column group) of dws_i
        const int tm = wave, tn = cg;
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        tile_mma<true, false>(P.fcw[ms] + (int64_t)(32 * tm) * FC, FC, 1, P.dWp + (int64_t)i * FC * FC + 32 * tn, FC, 1, li, lk, acc0, acc1);
        const float dg = P.dgamma[i * FC + 32 * tn + li];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * lk;
            tile[(32 * tm + row) * 33 + li] = fmaf(P.fcb[ms][32 * tm + row], dg, acc0[e] + acc1[e]);
        }
    }
    __syncthreads();
    relvec_bwd_cols256(J, i, cg, du, dv, tile, ps, pd);
}

// kind B: k_fold_bwd's B part.  q = 16 m + tile
__device__ __forceinline__ void tail_fold_B_block(const FoldTab& T, const FoldPtrs& P, int q, float* lds) {
    float (*red)[32 * 32] = (float (*)[32 * 32])lds;
    float (*dus)[32] = (float (*)[32])(lds + 8 * 1024);
    float (*dvs)[32] = (float (*)[32])(lds + 8 * 1024 + KGW_MAX_RELS * 32);
    float (*redb)[64] = (float (*)[64])(lds + 8 * 1024 + 2 * KGW_MAX_RELS * 32);
    const int t = threadIdx.x, lane = t & 63, li = lane & 31, lk = lane >> 5, wave = t >> 6;
    const int m = q >> 4, tile = q & 15, tm = tile >> 2, tn = tile & 3;
    for (int idx = t; idx < T.n * 32; idx += 256) {
        const int i = idx >> 5, c = idx & 31, r = T.rel_id[i], kk = 32 * tn + c;
        dus[i][c] = P.duv_pieces ? kgw_duv_sum8(P.dUp + r * 8 * FC + kk) : P.dUp[r * FC + kk];
        dvs[i][c] = P.duv_pieces ? kgw_duv_sum8(P.dVp + r * 8 * FC + kk) : P.dVp[r * FC + kk];
    }
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        const int vw = wave + 4 * pass;                   // the 512-thread block's wavefront this pass plays
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        float qb = 0.f;
        int seen = 0;
        for (int i = 0; i < T.n; ++i) {
            if (T.src_m[i] != m) continue;
            if ((seen++ & 7) != vw) continue;
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
            if (tn == 0) {
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
        for (int e = 0; e < 16; ++e) red[vw][((e & 3) + 8 * (e >> 2) + 4 * lk) * 32 + li] = acc0[e] + acc1[e];
        redb[vw][lane] = qb;
    }
    __syncthreads();
    const int col = t & 31, k = 32 * tn + col;
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {                      // (the 512-thread block's threads (t >> 5) and (t >> 5) + 8, rows +0 and +16 each)
        const int hr = (t >> 5) + 8 * it;                 // row inside the tile, 0..15
        float vs[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int row = hr + 16 * h2;
            float vsum = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) vsum += red[w8][row * 32 + col];
            vs[h2] = vsum;
        }
        const int h0 = 32 * tm + hr;
#pragma unroll 8
        for (int i = 0; i < T.n; ++i) {
            const int r = T.rel_id[i];
            const float fs = T.src_m[i] == m ? 1.f : 0.f, fd = T.dst_m[i] == m ? 1.f : 0.f;
            const float du = dus[i][col] * fs, dv = dvs[i][col] * fd;
            vs[0] = fmaf(P.U[r * FC + h0], du, vs[0]);      vs[0] = fmaf(P.V[r * FC + h0], dv, vs[0]);
            vs[1] = fmaf(P.U[r * FC + h0 + 16], du, vs[1]); vs[1] = fmaf(P.V[r * FC + h0 + 16], dv, vs[1]);
        }
        P.dfcw[m][h0 * FC + k] = vs[0];
        P.dfcw[m][(h0 + 16) * FC + k] = vs[1];
    }
    if (tn == 0 && t < 32) {
        float s = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) s += redb[w8][t] + redb[w8][32 + t];
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
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_param_tail(TnJobs JT, FoldTab FT, FoldPtrs FP, RvBwdJobs JR, TailIdx X) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (the longest blocks first: B -- up to two MFMA rounds of 64 steps per wavefront -- then F, R, and the products' row blocks)
    int b = (int)blockIdx.x;
    if (b < X.n_B) { tail_fold_B_block(FT, FP, b, lds); return; }
    b -= X.n_B;
    if (b < X.n_F) { tail_fold_rel_block(FT, FP, JR.j[X.fold_job], b, lds); return; }
    b -= X.n_F;
    if (b < X.n_R) {
        int jq = 0;
        while (jq + 1 < JR.n && b >= JR.j[jq + 1].blk0) ++jq;
        tail_relvec_block(JR.j[jq], b - JR.j[jq].blk0, lds);
        return;
    }
    b -= X.n_R;
    int jq = 0;
    while (jq + 1 < JT.n && b >= X.tn_flat0[jq + 1]) ++jq;
    const TnJob& T = JT.j[jq];
    const int l = b - X.tn_flat0[jq];
    const int bx = l % T.nblk, rest = l / T.nblk;
    tn_gemm_block<2, 2>(T, bx, rest % T.gy, rest / T.gy, lds);
}

}  // namespace

extern "C" int kgw_param_tail(int32_t n_tn, const KgwTnJob* tn_jobs, KgwGradSrc* src, const KgwFoldArgs* fold, int32_t n_relvec,
                              const KgwRelvecJob* relvec, int32_t fold_job, kgw_stream_t stream_) {
    if (n_tn < 0 || n_tn > TN_MAX_JOBS || n_relvec < 0 || n_relvec > KGW_MAX_LAYERS) return KGW_E_RANGE;
    if ((n_tn && (!tn_jobs || !src)) || (n_relvec && !relvec)) return KGW_E_NULL;
    if (fold && (fold_job < 0 || fold_job >= n_relvec)) return KGW_E_RANGE;
    hipStream_t st = (hipStream_t)stream_;
    auto aligned8 = [](const void* p) { return ((uintptr_t)p & 7) == 0; };
    TnPlan PL{};
    if (n_tn) {          // kgw_tn_gemm_multi_partial's checks and plan
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
        const int rc = launch_tn_jobs<2, 2>(d, n_tn, st, src, &PL);
        if (rc != KGW_OK) return rc;
    }
    TailIdx X{};
    X.tn_flat0[0] = 0;
    for (int q = 0; q < n_tn; ++q) X.tn_flat0[q + 1] = X.tn_flat0[q] + PL.J.j[q].nblk * PL.J.j[q].gy * PL.J.j[q].gz;
    X.n_tn = X.tn_flat0[n_tn];
    FoldTab FT{}; FoldPtrs FP{};
    if (fold) {
        const int rc = build(fold, &FT, &FP);
        if (rc) return rc;
        if (!FP.dUp || !FP.dVp || !FP.dkappa || !FP.dWp || !FP.dgamma) return KGW_E_NULL;
        for (int m = 0; m < FT.n_mlp; ++m)
            if (!FP.dfcw[m] || !FP.dfcb[m]) return KGW_E_NULL;
        if (relvec[fold_job].n_live != fold->n) return KGW_E_RANGE;
        X.n_B = 16 * FT.n_mlp;
        X.n_F = 4 * FT.n;
    }
    X.fold_job = fold ? fold_job : -1;
    RvBwdJobs JR{};
    int blk = 0;
    for (int q = 0; q < n_relvec; ++q) {                 // (slot q of JR = job q, so that X.fold_job indexes it; the fold's job has no R blocks)
        const KgwRelvecJob& D = relvec[q];
        if (D.n_live <= 0 && !(fold && q == fold_job)) return KGW_E_UNSUPPORTED;
        if (!D.rel_ids || !D.bip_pos || !D.w_src_t || !D.att_src || !D.att_dst || !D.dw_src_t || !D.datt_src || !D.datt_dst) return KGW_E_NULL;
        RvBwdJob& T = JR.j[q];
        T.blk0 = blk; T.v_by_rel = 1; T.pieces = D.duv_pieces;
        if (!(fold && q == fold_job)) blk += 4 * D.n_live;
        T.rel_ids = D.rel_ids; T.bip_pos = D.bip_pos; T.wsT = D.w_src_t; T.wdT = D.w_dst_t; T.att_src = D.att_src;
        T.att_dst = D.att_dst; T.dU_full = D.dU_full; T.dV = D.dV; T.dwsT = D.dw_src_t; T.dwdT = D.dw_dst_t;
        T.datt_src = D.datt_src; T.datt_dst = D.datt_dst; T.dws_acc = D.dw_src_acc;
    }
    JR.n = n_relvec; JR.blk_end = blk;
    X.n_R = blk;
    constexpr int FRAG = 2 * 2 * 16 * 64;
    constexpr size_t lds_tn = (size_t)(2 * FRAG + 4 * 32 * 2) * sizeof(float);
    constexpr size_t lds_bytes = lds_tn > TAIL_LDS_FLOATS * sizeof(float) ? lds_tn : TAIL_LDS_FLOATS * sizeof(float);
    static_assert(TAIL_LDS_FLOATS >= 2 * FC + FC * 33 + 2 * 32 * 33, "kind F / R fit in kind B's LDS");
    const int total = X.n_B + X.n_F + X.n_R + X.n_tn;
    if (total == 0) return KGW_OK;
    k_param_tail<<<total, 256, lds_bytes, st>>>(PL.J, FT, FP, JR, X);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_relvec_fwd_multi(int32_t n_jobs, const KgwRelvecJob* jobs, kgw_stream_t stream_) {
    if (n_jobs <= 0) return KGW_OK;
    if (!jobs) return KGW_E_NULL;
    if (n_jobs > KGW_MAX_LAYERS) return KGW_E_RANGE;
    return relvec_fwd_launch(n_jobs, jobs, 1, (hipStream_t)stream_);
}

extern "C" int kgw_relvec_bwd_multi(int32_t n_jobs, const KgwRelvecJob* jobs, kgw_stream_t stream_) {
    if (n_jobs <= 0) return KGW_OK;
    if (!jobs) return KGW_E_NULL;
    if (n_jobs > KGW_MAX_LAYERS) return KGW_E_RANGE;
    return relvec_bwd_launch(n_jobs, jobs, 1, (hipStream_t)stream_);
}

extern "C" int kgw_relvec_fwd(int32_t n_rels_total, const int32_t* live_of_rel, const int32_t* bip_pos, const float* w_src_t,
                              const float* w_dst_t, const float* att_src, const float* att_dst, float* U_full, float* V,
                              int32_t v_by_rel, int32_t n_live, const float* bias, const int32_t* blk_of_live,
                              int32_t n_blk, float* bias_sum, float* zero_buf, int64_t zero_floats, kgw_stream_t stream_) {
    if (n_rels_total <= 0) return KGW_OK;
    KgwRelvecJob j{};
    j.n_rels_total = n_rels_total; j.n_live = n_live; j.n_blk = n_blk; j.live_of_rel = live_of_rel; j.bip_pos = bip_pos;
    j.w_src_t = w_src_t; j.w_dst_t = w_dst_t; j.att_src = att_src; j.att_dst = att_dst; j.U_full = U_full; j.V = V; j.bias = bias;
    j.blk_of_live = blk_of_live; j.bias_sum = bias_sum; j.zero_buf = zero_buf; j.zero_floats = zero_floats;
    return relvec_fwd_launch(1, &j, v_by_rel, (hipStream_t)stream_);
}

extern "C" int kgw_relvec_bwd_acc(int32_t n_live, const int32_t* rel_ids, const int32_t* bip_pos, const float* w_src_t,
                                  const float* w_dst_t, const float* att_src, const float* att_dst, const float* dU_full,
                                  const float* dV, const float* dw_src_acc, float* dw_src_t, float* dw_dst_t, float* datt_src,
                                  float* datt_dst, int32_t v_by_rel, kgw_stream_t stream_) {
    if (n_live <= 0) return KGW_OK;
    KgwRelvecJob j{};
    j.n_live = n_live; j.rel_ids = rel_ids; j.bip_pos = bip_pos; j.w_src_t = w_src_t; j.w_dst_t = w_dst_t; j.att_src = att_src;
    j.att_dst = att_dst; j.dU_full = dU_full; j.dV = dV; j.dw_src_acc = dw_src_acc; j.dw_src_t = dw_src_t; j.dw_dst_t = dw_dst_t;
    j.datt_src = datt_src; j.datt_dst = datt_dst;
    return relvec_bwd_launch(1, &j, v_by_rel, (hipStream_t)stream_);
}

extern "C" int kgw_relvec_bwd(int32_t n_live, const int32_t* rel_ids, const int32_t* bip_pos, const float* w_src_t,
                              const float* w_dst_t, const float* att_src, const float* att_dst, const float* dU_full,
                              const float* dV, float* dw_src_t, float* dw_dst_t, float* datt_src, float* datt_dst,
                              int32_t v_by_rel, kgw_stream_t stream_) {
    if (n_live > 0 && (!dU_full || !dV)) return KGW_E_NULL;
    return kgw_relvec_bwd_acc(n_live, rel_ids, bip_pos, w_src_t, w_dst_t, att_src, att_dst, dU_full, dV, nullptr, dw_src_t,
                              dw_dst_t, datt_src, datt_dst, v_by_rel, stream_);
}
