// kgw_fold_common.h -- tables, pointers and the 32 x 32 MFMA tile product of the FC_output fold (kgw_fold.hip), shared with the
// parameter-only riders of the gene layer's kgw_gemm3 launch (kgw_riders.h).
#pragma once
#include "kgw_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int FC = 128;            // hidden width
constexpr int FOLD_MAX_MLP = 4;

struct FoldTab {                   // by value: scalar loads from the kernarg segment
    int32_t n, n_rels, n_mlp, pad_;
    int8_t rel_id[KGW_MAX_RELS];   // relation id of packed slot i
    int8_t live_of[KGW_MAX_RELS];  // packed slot of relation id r, -1 = not in the pack
    int8_t src_m[KGW_MAX_RELS];    // MLP of the source / destination node type of packed slot i
    int8_t dst_m[KGW_MAX_RELS];
};

struct FoldPtrs {
    const float* w_src_t;                       // [n][128][128]  W_i^T ([in, out])
    const float* fcw[FOLD_MAX_MLP];             // FC_output.weight [out c][in k]   (T[k][c] = fcw[c][k])
    const float* fcb[FOLD_MAX_MLP];             // FC_output.bias [c]
    const float* U; const float* V;             // [n_rels][128] by relation id (kgw_relvec_fwd)
    float* Up; float* Vp; float* kappa;         // [n_rels][128], [n_rels][128], [n_rels]
    float* Wp; float* gamma;                    // [n][128][128], [n][128]
    const float* dUp; const float* dVp; const float* dkappa; const float* dWp; const float* dgamma;
    int duv_pieces;                             // dUp / dVp are [n_rels][8][128] pieces (KGW_F_DUV_PIECES)
    float* dU; float* dV;                       // [n_rels][128]
    float* dws;                                 // [n][128][128]
    float* dfcw[FOLD_MAX_MLP]; float* dfcb[FOLD_MAX_MLP];
};

// 32 x 32 tile of C = A B over K = 128 on one wavefront.  A(m, k) = pa[m * sam + k * sak], B(k, n) = pb[k * sbk + n * sbn]
// for the tile's rows m = li and columns n = li; MFMA step j multiplies k = 64 lk + j (a permutation of the sum).
// AK / BK: the operand is contiguous along k (sak / sbk == 1): the lane's 64 values come as 16 float4 loads.
template <bool AK, bool BK>
__device__ __forceinline__ void tile_mma(const float* __restrict__ pa, int sam, int sak, const float* __restrict__ pb, int sbk,
                                         int sbn, int li, int lk, f32x16& acc0, f32x16& acc1) {
    float a[64], b[64];
    const float* qa = pa + li * sam + 64 * lk * sak;
    const float* qb = pb + li * sbn + 64 * lk * sbk;
    if (AK) {
#pragma unroll
        for (int q = 0; q < 16; ++q) { const float4 v = ((const float4*)qa)[q]; a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w; }
    } else {
#pragma unroll
        for (int j = 0; j < 64; ++j) a[j] = qa[j * sak];
    }
    if (BK) {
#pragma unroll
        for (int q = 0; q < 16; ++q) { const float4 v = ((const float4*)qb)[q]; b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w; }
    } else {
#pragma unroll
        for (int j = 0; j < 64; ++j) b[j] = qb[j * sbk];
    }
    __builtin_amdgcn_sched_barrier(0);            // all loads in flight before the first MFMA waits
#pragma unroll
    for (int j = 0; j < 64; j += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j + 1], b[j + 1], acc1, 0, 0, 0);
    }
}

inline int build(const KgwFoldArgs* a, FoldTab* T, FoldPtrs* P) {
    if (!a) return KGW_E_NULL;
    if (a->n < 1 || a->n > KGW_MAX_RELS || a->n_rels < a->n || a->n_rels > KGW_MAX_RELS || a->n_mlp < 1 || a->n_mlp > FOLD_MAX_MLP)
        return KGW_E_RANGE;
    if (!a->rel_ids_host || !a->src_mlp_host || !a->dst_mlp_host || !a->w_src_t || !a->U || !a->V) return KGW_E_NULL;
    T->n = a->n; T->n_rels = a->n_rels; T->n_mlp = a->n_mlp; T->pad_ = 0;
    for (int r = 0; r < KGW_MAX_RELS; ++r) { T->live_of[r] = -1; T->rel_id[r] = 0; T->src_m[r] = 0; T->dst_m[r] = 0; }
    for (int i = 0; i < a->n; ++i) {
        const int r = a->rel_ids_host[i], sm = a->src_mlp_host[i], dm = a->dst_mlp_host[i];
        if (r < 0 || r >= a->n_rels || sm < 0 || sm >= a->n_mlp || dm < 0 || dm >= a->n_mlp) return KGW_E_RANGE;
        T->rel_id[i] = (int8_t)r; T->live_of[r] = (int8_t)i; T->src_m[i] = (int8_t)sm; T->dst_m[i] = (int8_t)dm;
    }
    P->w_src_t = a->w_src_t; P->U = a->U; P->V = a->V;
    for (int m = 0; m < FOLD_MAX_MLP; ++m) {
        P->fcw[m] = m < a->n_mlp ? a->fc_weight[m] : nullptr; P->fcb[m] = m < a->n_mlp ? a->fc_bias[m] : nullptr;
        P->dfcw[m] = m < a->n_mlp ? a->d_fc_weight[m] : nullptr; P->dfcb[m] = m < a->n_mlp ? a->d_fc_bias[m] : nullptr;
        if (m < a->n_mlp && (!P->fcw[m] || !P->fcb[m])) return KGW_E_NULL;
    }
    P->Up = a->Up; P->Vp = a->Vp; P->kappa = a->kappa; P->Wp = a->Wp; P->gamma = a->gamma;
    P->dUp = a->dUp; P->dVp = a->dVp; P->dkappa = a->dkappa; P->dWp = a->dWp; P->dgamma = a->dgamma;
    P->duv_pieces = a->duv_pieces;
    P->dU = a->dU; P->dV = a->dV; P->dws = a->dws;
    return KGW_OK;
}

}  // namespace
