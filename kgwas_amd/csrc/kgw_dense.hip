// kgw_dense.hip -- MFMA (fp32-in / fp32-accumulate, exact) kernels for the dense side of the KGWAS path.
//
// kgw_tn_gemm:  C[M,N] = A[rows,M]^T * B[rows,N]   (+ optional column sums of A)
// for TALL inputs (rows ~ 1e5: every sampled SNP / gene) and small M, N -- the weight gradients of the
// feature MLPs (kgwas/model.py:13-21), of the per-relation lin_src maps (kgwas/conv.py:138,142) and the
// d u_r / d v_r attention-vector gradients.  A library GEMM runs these shapes on a few dozen workgroups
// (K = rows is its reduction dimension: 290-340 us per product on MI355X); here the reduction is split over
// every SIMD of the chip.
//
// Mapping (gfx950): v_mfma_f32_32x32x2_f32.  For a TN product the MFMA operand layout IS the memory layout:
// lane l supplies A[row k = l>>5][column i = l&31] -- a wave-instruction reads two contiguous 128-float rows.
// Columns are interleaved (tile t owns columns MT*i + t) so one 16-byte load per lane feeds all MT tiles.
// One wavefront owns the whole (32 MT) x (32 NT) accumulator (up to 256 accumulator VGPRs, one wave per
// SIMD) and streams its slice of rows: 2 KiB of loads per 16 MFMAs (1024 cycles) -- HBM and the matrix pipe
// are balanced at ~5 TB/s.  Waves of a block are summed through LDS, blocks through a partial buffer and a
// second kernel in a fixed order: no atomics, deterministic.
#include "kgw_common.h"
#include "kgw_fold_common.h"
#include <stdlib.h>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V> struct VecLoad;
template <> struct VecLoad<1> {
    static __device__ __forceinline__ void ld(const float* p, bool ok, float (&o)[1]) { float v = *p; o[0] = ok ? v : 0.f; }
};
template <> struct VecLoad<2> {
    static __device__ __forceinline__ void ld(const float* p, bool ok, float (&o)[2]) {
        float2 v = *(const float2*)p; o[0] = ok ? v.x : 0.f; o[1] = ok ? v.y : 0.f; }
};
template <> struct VecLoad<4> {
    static __device__ __forceinline__ void ld(const float* p, bool ok, float (&o)[4]) {
        float4 v = *(const float4*)p;
        o[0] = ok ? v.x : 0.f; o[1] = ok ? v.y : 0.f; o[2] = ok ? v.z : 0.f; o[3] = ok ? v.w : 0.f; }
};

constexpr int TN_U = 4;   // row pairs per pipeline stage

template <int MT, int NT>
struct Stage { float a[TN_U][MT]; float b[TN_U][NT]; };

// Unmasked stage load: TN_U row pairs starting at the lane's row pointer (pa/pb already include row k and
// the lane's column).  Out-of-range COLUMNS are clamped to column 0 by the caller: they feed accumulator
// rows / columns that are never stored, so they need no masking.
template <int MT, int NT>
__device__ __forceinline__ void tn_load(Stage<MT, NT>& s, const float* pa, int64_t lda2, const float* pb, int64_t ldb2) {
#pragma unroll
    for (int u = 0; u < TN_U; ++u) {
        VecLoad<MT>::ld(pa + u * lda2, true, s.a[u]);
        VecLoad<NT>::ld(pb + u * ldb2, true, s.b[u]);
    }
}

template <int MT, int NT>
__device__ __forceinline__ void tn_mma(const Stage<MT, NT>& s, f32x16 (&acc)[MT][NT], float (&sa)[MT]) {
#pragma unroll
    for (int u = 0; u < TN_U; ++u) {
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            sa[a] += s.a[u][a];
#pragma unroll
            for (int b = 0; b < NT; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(s.a[u][a], s.b[u][b], acc[a][b], 0, 0, 0);
        }
    }
}

// which matrix pipe the 64 x 64-per-wavefront tiling uses: 1 = bf16 with three exact pieces per operand (default), 0 = fp32
// (KGW_TN_SPLIT=0, or kgw_tn_split(0): the A/B of tests/test_gpu_dense.py and the fallback)
static int g_tn_split = -1;
static bool tn_split_on() {
    if (g_tn_split < 0) g_tn_split = !(getenv("KGW_TN_SPLIT") && getenv("KGW_TN_SPLIT")[0] == '0');
    return g_tn_split != 0;
}

static int64_t g_tn_direct = -1;
static int64_t tn_direct_rows() {
    if (g_tn_direct < 0) g_tn_direct = getenv("KGW_TN_DIRECT_ROWS") ? atoll(getenv("KGW_TN_DIRECT_ROWS")) : 0;
    return g_tn_direct;
}

// Up to four products of one tiling per launch (the weight gradients of one MLP: same rows, different operands): the
// x dimension of the grid is the concatenation of the jobs' row blocks.
constexpr int TN_MAX_JOBS = 4;
struct TnJob {
    const float* A; const float* B; float* C; float* colsum; float* ws; float* ws_cs; const int32_t* rows_dev;
    int64_t lda, ldb, rows, rpw, c_rs, c_cs, cs_ld;
    int M, N, nblk, blk0, gy, gz, cs_rep, pad_;
};
struct TnJobs { TnJob j[TN_MAX_JOBS]; int n; };

// ws layout per block: [MT][NT][16][64] floats (fragment order) ; colsum ws per block: [32*MT]
// Rows [r0, r1) of a wavefront's 64 x 64 accumulator on the BF16 matrix pipe with fp32 error (round 5): every operand value is split
// EXACTLY into three bf16 pieces (kgw_split3x8, as in kgw_gemm3.hip / k_mlp2_bwd_first3) and the six piece products of weight
// >= 2^-16 are accumulated in fp32 -- the three dropped ones are below the rounding of one fp32 multiply-add.  An MFMA step
// takes 16 rows (lane group kg the rows 8 kg .. 8 kg + 7, eight float2 loads per operand: the lane's two columns of a row):
// 24 MFMAs of 32 cycles per 16 rows against 32 of 64 on the fp32 pipe.  The bf16 MFMA's internal add truncates (a small negative
// mean error): wavefronts with ``neg`` multiply their A values NEGATED (exact) and the caller negates their accumulator back, so
// the means of the four wavefronts of a block cancel.  sa: the plain column sums of A (fp32 VALU, as before).
struct TnStage3 { float2 a[8], b[8]; };
__device__ __forceinline__ void tn_mma3(const TnStage3& s, const unsigned sgn, f32x16 (&acc)[2][2], float (&sa)[2]) {
    float xa0[8], xa1[8], xb0[8], xb1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sa[0] += s.a[j].x; sa[1] += s.a[j].y;
        xa0[j] = kgw_fxor(s.a[j].x, sgn); xa1[j] = kgw_fxor(s.a[j].y, sgn);
        xb0[j] = s.b[j].x; xb1[j] = s.b[j].y;
    }
    uint4 pa0[3], pa1[3], pb0[3], pb1[3];
    kgw_split3x8(xa0, pa0[0], pa0[1], pa0[2]);
    kgw_split3x8(xa1, pa1[0], pa1[1], pa1[2]);
    kgw_split3x8(xb0, pb0[0], pb0[1], pb0[2]);
    kgw_split3x8(xb1, pb1[0], pb1[1], pb1[2]);
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};       // (piece of A, piece of B), smallest products first
#pragma unroll
    for (int t6 = 0; t6 < 6; ++t6) {
        const kgw_bf8 a0 = __builtin_bit_cast(kgw_bf8, pa0[TA[t6]]), a1 = __builtin_bit_cast(kgw_bf8, pa1[TA[t6]]);
        const kgw_bf8 b0 = __builtin_bit_cast(kgw_bf8, pb0[TB[t6]]), b1 = __builtin_bit_cast(kgw_bf8, pb1[TB[t6]]);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// (sign periods, as in kgw_gemm3: every TN3_FLIP steps the accumulator and the sign of the A operand flip together -- exact -- so
//  that the truncation pulls the running sum down in one period and up in the next, also WITHIN a long row range; sgn: in = the
//  wavefront's starting sign, out = the sign the accumulator is left with)
constexpr int TN3_FLIP = 4;
__device__ __forceinline__ void tn3_flip(unsigned& sgn, f32x16 (&acc)[2][2]) {
    sgn ^= 0x80000000u;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = -acc[a][b][e];
}
__device__ __forceinline__ void tn_rows_split3(const float* __restrict__ A, const int64_t lda, const int cas, const float* __restrict__ B,
                                               const int64_t ldb, const int cbs, const int64_t r0, const int64_t r1, const int kg,
                                               unsigned& sgn, f32x16 (&acc)[2][2], float (&sa)[2]) {
    const int64_t nfull = (r1 - r0) / 16;                  // steps made of valid rows only
    const float* pa = A + (r0 + 8 * kg) * lda + cas;
    const float* pb = B + (r0 + 8 * kg) * ldb + cbs;
    if (nfull > 0) {
        TnStage3 cur, nxt;
#pragma unroll
        for (int j = 0; j < 8; ++j) { cur.a[j] = *(const float2*)(pa + j * lda); cur.b[j] = *(const float2*)(pb + j * ldb); }
        for (int64_t it = 1; it < nfull; ++it) {
            pa += 16 * lda; pb += 16 * ldb;
#pragma unroll
            for (int j = 0; j < 8; ++j) { nxt.a[j] = *(const float2*)(pa + j * lda); nxt.b[j] = *(const float2*)(pb + j * ldb); }
            tn_mma3(cur, sgn, acc, sa);                     // (the next stage's loads in flight under the 24 MFMAs)
            if ((it & (TN3_FLIP - 1)) == 0) tn3_flip(sgn, acc);
            cur = nxt;
        }
        tn_mma3(cur, sgn, acc, sa);
    }
    const int64_t rt = r0 + nfull * 16;
    if (rt < r1) {                                         // tail: < 16 rows, masked per row (loads clamped to the last valid row)
        TnStage3 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t row = rt + 8 * kg + j;
            const bool ok = row < r1;
            const int64_t rc = ok ? row : (r1 - 1);
            const float2 va = *(const float2*)(A + rc * lda + cas), vb = *(const float2*)(B + rc * ldb + cbs);
            t.a[j] = ok ? va : make_float2(0.f, 0.f);
            t.b[j] = ok ? vb : make_float2(0.f, 0.f);
        }
        tn_mma3(t, sgn, acc, sa);
    }
}

// (the body of k_tn_gemm: block (bxg = row block over all jobs, by, bz); also inlined into k_transform_bwd)
template <int MT, int NT>
__device__ __forceinline__ void tn_gemm_block(const TnJob& T, const int bx, const int by, const int bz, float* lds) {
    const float* __restrict__ A = T.A;
    const float* __restrict__ B = T.B;
    const int64_t lda = T.lda, ldb = T.ldb;
    const int M = T.M, N = T.N;
    int64_t rows = T.rows, rows_per_wave = T.rpw;
    float* __restrict__ ws = T.ws;
    float* __restrict__ ws_colsum = T.ws_cs;
    const int nbx = T.nblk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = lane >> 5, i = lane & 31;
    const int m0 = by * 32 * MT, n0 = bz * 32 * NT;
    const int ca = m0 + MT * i, cb = n0 + NT * i;
    const int cas = ca < M ? ca : 0, cbs = cb < N ? cb : 0;
    const int64_t wg = (int64_t)bx * 4 + wave;
    if (T.rows_dev) {                   // actual row count of the batch (<= the static capacity `rows`): re-split evenly
        const int64_t re = min(rows, (int64_t)max(*T.rows_dev, 0));
        rows = re;
        rows_per_wave = ((re + (int64_t)nbx * 4 - 1) / ((int64_t)nbx * 4) + 1) & ~(int64_t)1;
    }
    const int64_t r0 = min(rows, wg * rows_per_wave);
    const int64_t r1 = min(rows, r0 + rows_per_wave);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    float sa[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) sa[a] = 0.f;

    unsigned flipm = 0u;                                    // sign of this wavefront's accumulator (the bf16 path's odd wavefronts)
    bool split3 = false;
    if constexpr (MT == 2 && NT == 2) split3 = T.pad_ != 0;
    if (split3) {
        if constexpr (MT == 2 && NT == 2) {
            flipm = (wg & 1) ? 0x80000000u : 0u;
            if (r0 < r1) tn_rows_split3(A, lda, cas, B, ldb, cbs, r0, r1, k, flipm, acc, sa);
        }
    } else if (r0 < r1) {
        constexpr int STEP = 2 * TN_U;                      // rows per stage
        const int64_t nfull = (r1 - r0) / STEP;             // stages made of valid rows only
        const float* pa = A + (r0 + k) * lda + cas;
        const float* pb = B + (r0 + k) * ldb + cbs;
        const int64_t lda2 = 2 * lda, ldb2 = 2 * ldb;
        if (nfull > 0) {
            Stage<MT, NT> cur, nxt;
            tn_load<MT, NT>(cur, pa, lda2, pb, ldb2);
            for (int64_t it = 1; it < nfull; ++it) {
                pa += STEP * lda; pb += STEP * ldb;
                tn_load<MT, NT>(nxt, pa, lda2, pb, ldb2);   // in flight while the 16*TN_U MFMAs below run
                tn_mma<MT, NT>(cur, acc, sa);
                cur = nxt;
            }
            tn_mma<MT, NT>(cur, acc, sa);
            pa += STEP * lda; pb += STEP * ldb;
        }
        // tail: < STEP rows, masked per row (loads clamped to the last valid row)
        const int64_t rt = r0 + nfull * STEP;
        if (rt < r1) {
            Stage<MT, NT> t;
#pragma unroll
            for (int u = 0; u < TN_U; ++u) {
                const int64_t row = rt + 2 * u + k;
                const bool ok = row < r1;
                const int64_t rc = ok ? row : (r1 - 1);
                VecLoad<MT>::ld(A + rc * lda + cas, ok, t.a[u]);
                VecLoad<NT>::ld(B + rc * ldb + cbs, ok, t.b[u]);
            }
            tn_mma<MT, NT>(t, acc, sa);
        }
    }

    // ---- reduce the 4 waves of the block through LDS, fixed order (w0+w2) + (w1+w3) --------------------
    // Accumulators live in AGPRs: they are only ever READ here (16 at a time), never written back -- a
    // read-modify-write of all 256 would need 256 arch VGPRs at once and spill to scratch.
    constexpr int FRAG = MT * NT * 16 * 64;              // floats per wave
    float* reg0 = lds;
    float* reg1 = lds + FRAG;
    float* cs = lds + 2 * FRAG;                           // [4][32*MT] column sums
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        const float t = sa[a] + kgw_xhalf(sa[a]);        // rows k = 0 and k = 1 of the pairs
        if (k == 0) cs[wave * 32 * MT + MT * i + a] = t;
    }
    if (wave >= 2) {
        float* dst = (wave == 2) ? reg0 : reg1;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) {
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[((a * NT + b) * 16 + e) * 64 + lane] = kgw_fxor(acc[a][b][e], flipm);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    __syncthreads();
    if (wave < 2) {
        float* dst = (wave == 0) ? reg0 : reg1;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int idx = ((a * NT + b) * 16 + e) * 64 + lane;
                    dst[idx] = kgw_fxor(acc[a][b][e], flipm) + dst[idx];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    __syncthreads();
    if (nbx == 1) {
        // few rows (the layer transforms' weight gradients: ~1 k destination rows): ONE row block per tile, so the block's
        // sum is the result -- written straight to C (and the column sums), no partial buffer, no second launch
        float* __restrict__ Cq = T.C;
        for (int f = threadIdx.x * 4; f < FRAG; f += 256 * 4) {
            const float4 x = *(const float4*)(reg0 + f), y = *(const float4*)(reg1 + f);
            const float v[4] = {x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w};
            const int e = (f >> 6) & 15, tb = (f >> 10) % NT, ta = (f >> 10) / NT;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ln = (f & 63) + q;
                const int ti = (e & 3) + 8 * (e >> 2) + 4 * (ln >> 5), tj = ln & 31;
                const int m = m0 + MT * ti + ta, n = n0 + NT * tj + tb;
                if (m < M && n < N) Cq[(int64_t)m * T.c_rs + (int64_t)n * T.c_cs] = v[q];
            }
        }
        if (T.colsum && bz == 0 && threadIdx.x < 32 * MT) {
            const int c = threadIdx.x;
            const float t = (cs[c] + cs[2 * 32 * MT + c]) + (cs[32 * MT + c] + cs[3 * 32 * MT + c]);
            if (m0 + c < M)
                for (int q = 0; q < T.cs_rep; ++q) T.colsum[(int64_t)q * T.cs_ld + m0 + c] = t;
        }
        return;
    }
    const int64_t blk = ((int64_t)bz * T.gy + by) * nbx + bx;
    float* out = ws + blk * FRAG;
    for (int f = threadIdx.x * 4; f < FRAG; f += 256 * 4) {
        const float4 x = *(const float4*)(reg0 + f), y = *(const float4*)(reg1 + f);
        *(float4*)(out + f) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
    if (ws_colsum && bz == 0 && threadIdx.x < 32 * MT) {
        const int c = threadIdx.x;
        const float t = (cs[c] + cs[2 * 32 * MT + c]) + (cs[32 * MT + c] + cs[3 * 32 * MT + c]);
        ws_colsum[((int64_t)by * nbx + bx) * 32 * MT + c] = t;
    }
}

template <int MT, int NT>
__global__ void __launch_bounds__(256, 1) k_tn_gemm(TnJobs J) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int jq = 0;
    while (jq + 1 < J.n && (int)blockIdx.x >= J.j[jq + 1].blk0) ++jq;
    const TnJob& T = J.j[jq];
    if ((int)blockIdx.y >= T.gy || (int)blockIdx.z >= T.gz) return;       // (before any barrier: whole blocks)
    tn_gemm_block<MT, NT>(T, (int)blockIdx.x - T.blk0, (int)blockIdx.y, (int)blockIdx.z, lds);
}

// C[m][n] = sum over row-blocks of the partial fragments (fixed order); also the column sums.
// Block = 64 fragment elements x 4 groups of row-blocks; every thread keeps 8 loads in flight.
// (the body of k_tn_reduce: block (bx = 64 fragment elements, by, bzz = job * gz_max + bz); sm: 256 floats of LDS.  Also the
//  reduce blocks that ride in a later launch: k_transform_bwd, k_tn_gemm_ride)
template <int MT, int NT>
__device__ __forceinline__ void tn_reduce_block(const TnJobs& J, const int gz_max, const int bx_, const int by, const int bzz, float* sm) {
    constexpr int FRAG = MT * NT * 16 * 64;
    const TnJob& T = J.j[bzz / gz_max];
    const int bz = bzz % gz_max;
    if (by >= T.gy || bz >= T.gz) return;
    if (T.nblk == 1) return;                 // single row block: k_tn_gemm wrote C and the column sums itself
    const float* __restrict__ ws = T.ws;
    const float* __restrict__ ws_colsum = T.ws_cs;
    const int nblk = T.nblk, gy = T.gy, M = T.M, N = T.N, cs_rep = T.cs_rep;
    float* __restrict__ C = T.C;
    float* __restrict__ colsum = T.colsum;
    const int64_t c_rs = T.c_rs, c_cs = T.c_cs, cs_ld = T.cs_ld;
    const int m0 = by * 32 * MT, n0 = bz * 32 * NT;
    const int fl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int f = bx_ * 64 + fl;
    {
        const float* p = ws + ((int64_t)bz * gy + by) * nblk * FRAG + f;
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = g;
        for (; b + 28 < nblk; b += 32) {
#pragma unroll
            for (int q = 0; q < 8; ++q) s8[q] += p[(int64_t)(b + 4 * q) * FRAG];
        }
        for (int q = 0; b < nblk; b += 4, ++q) s8[q & 7] += p[(int64_t)b * FRAG];
        sm[threadIdx.x] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    }
    __syncthreads();
    if (g == 0) {
        const float s = (sm[fl] + sm[64 + fl]) + (sm[128 + fl] + sm[192 + fl]);
        const int lane = f & 63, e = (f >> 6) & 15, tb = (f >> 10) % NT, ta = (f >> 10) / NT;
        const int ti = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);     // row of the 32x32 tile
        const int tj = lane & 31;                                     // column of the tile
        const int m = m0 + MT * ti + ta, n = n0 + NT * tj + tb;
        if (m < M && n < N) C[(int64_t)m * c_rs + (int64_t)n * c_cs] = s;
    }
    if (colsum && bz == 0 && bx_ == 0) {
        // 32*MT columns x (256 / (32*MT)) groups of row-blocks, 4 loads in flight per thread, fixed order
        constexpr int NC = 32 * MT, NG = 256 / NC;
        __syncthreads();
        const int c = threadIdx.x % NC, gq = threadIdx.x / NC;
        const float* p = ws_colsum + (int64_t)by * nblk * NC + c;
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        int b = gq;
        for (; b + 3 * NG < nblk; b += 4 * NG) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s4[q] += p[(int64_t)(b + q * NG) * NC];
        }
        for (int q = 0; b < nblk; b += NG, ++q) s4[q & 3] += p[(int64_t)b * NC];
        sm[threadIdx.x] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        __syncthreads();
        if (gq == 0) {
            float t = 0.f;
            for (int q = 0; q < NG; ++q) t += sm[q * NC + c];
            if (m0 + c < M)
                for (int q = 0; q < cs_rep; ++q) colsum[(int64_t)q * cs_ld + m0 + c] = t;
        }
    }
}

template <int MT, int NT>
__global__ void __launch_bounds__(256) k_tn_reduce(TnJobs J, int gz_max) {
    __shared__ float sm[256];
    tn_reduce_block<MT, NT>(J, gz_max, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, sm);
}

// A product group's second launch (k_tn_reduce<2,2>) that has not been issued: the C ABI's KgwTnReducePlan.  The gradients it
// finishes feed nothing before the end of the backward pass, so its blocks ride in a later launch of this file instead
// (kgw_transform_bwd_ex, kgw_tn_gemm_partial_ride) or are launched by kgw_tn_reduce_launch.
struct TnReducePlan { int32_t valid, blocks, gy_max, gz_max; int32_t n, pad_[3]; TnJobs J; };     // (valid, blocks: KgwTnReducePlan's public fields)
static_assert(sizeof(TnReducePlan) <= sizeof(KgwTnReducePlan), "KgwTnReducePlan holds a TnReducePlan");
constexpr int TN22_FRAG = 2 * 2 * 16 * 64;
inline int tn_reduce_plan_blocks(const TnReducePlan& R) { return R.valid ? (TN22_FRAG / 64) * R.gy_max * R.gz_max * R.n : 0; }
// flat block index b of a plan's grid (TN22_FRAG / 64, gy_max, gz_max * n)
__device__ __forceinline__ void tn_reduce_plan_block(const TnJobs& J, int gy_max, int gz_max, int b, float* sm) {
    constexpr int NX = TN22_FRAG / 64;
    tn_reduce_block<2, 2>(J, gz_max, b % NX, (b / NX) % gy_max, b / (NX * gy_max), sm);
}

// k_tn_gemm<2,2> with the reduce blocks of an earlier product group in front (flat grid; the product's blocks in the 3-D grid's order)
struct TnRideIdx { int n_rd, rd_gy, rd_gz, blk, gy_max, gz_max; };
__global__ void __launch_bounds__(256, 1) k_tn_gemm_ride(TnJobs J, TnJobs JR, TnRideIdx X) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = (int)blockIdx.x;
    if (b < X.n_rd) { tn_reduce_plan_block(JR, X.rd_gy, X.rd_gz, b, lds); return; }
    const int l = b - X.n_rd;
    const int bx = l % X.blk, by = (l / X.blk) % X.gy_max, bz = l / (X.blk * X.gy_max);
    int jq = 0;
    while (jq + 1 < J.n && bx >= J.j[jq + 1].blk0) ++jq;
    const TnJob& T = J.j[jq];
    if (by >= T.gy || bz >= T.gz) return;
    tn_gemm_block<2, 2>(T, bx - T.blk0, by, bz, lds);
}

struct TnDesc {      // one product as the C ABI describes it
    const float* A; int64_t lda; int M; const float* B; int64_t ldb; int N; int64_t rows; float* C; int64_t ldc; bool c_t;
    float* colsum; int cs_rep; int64_t cs_ld; float* ws; int64_t ws_floats; const int32_t* rows_dev;
};

// ``defer`` (nullable, 2 n records: [product, column sums] of every job): the second launch is left to kgw_adam_fused, which
// adds the row blocks' partials in k_tn_reduce's order while it updates the parameter the gradient belongs to.
// ``plan`` (nullable): fill it with the job table and return without launching anything (k_transform_bwd runs the blocks)
struct TnPlan { TnJobs J; int blk, gy_max, gz_max; bool all_direct; };
template <int MT, int NT>
int launch_tn_jobs(const TnDesc* d, int n, hipStream_t st, KgwGradSrc* defer = nullptr, TnPlan* plan = nullptr,
                   const TnReducePlan* ride = nullptr) {
    constexpr int FRAG = MT * NT * 16 * 64;
    TnJobs J{};
    J.n = n;
    int blk = 0, gy_max = 0, gz_max = 0;
    bool all_direct = true;
    for (int q = 0; q < n; ++q) {
        const TnDesc& D = d[q];
        TnJob& T = J.j[q];
        const int gy = (D.M + 32 * MT - 1) / (32 * MT), gz = (D.N + 32 * NT - 1) / (32 * NT);
        // one block per CU at most; at least 64 rows per wavefront
        int64_t nblk = (D.rows + 4 * 64 - 1) / (4 * 64);
        static const int64_t cap_small = getenv("KGW_TN_CAP") ? atoll(getenv("KGW_TN_CAP")) : 512;
        int64_t cap = ((MT * NT <= 4) ? cap_small : 256) / ((int64_t)gy * gz);      // (small accumulators: two blocks per CU)
        if (cap < 1) cap = 1;
        if (nblk > cap) nblk = cap;
        if (nblk < 1) nblk = 1;
        // (ONE row block for a product of up to n rows and >= 16 tiles -- its blocks write the result directly, no partial slabs, no
        //  second launch -- is NOT the default.  On the fp32 pipe it lost: 1.491 -> 1.501 ms at 640 rows, 1.513 at 2048, the serial
        //  row loop cost more than the launch.  On the bf16 pipe it wins a little -- layer 1's transform products, 1 171 rows x 68
        //  tiles, without their 9 088-block k_tn_reduce: 1.0594 / 1.0559 against 1.0600 / 1.0619 ms, 26 -> 25 launches -- but a
        //  wavefront then adds ~430 rows into one accumulator instead of ~60 and the longer chain shows: max error / sum|a||b|
        //  4.2e-7 (the fp32 pipe in the same structure: 6.1e-7) against 2.1e-7 / 2.0e-7 with row blocks at 1 700 x 128 x 1 408.
        //  Twice the error for 3 us: off.  kgw_tn_direct_rows(n) / KGW_TN_DIRECT_ROWS=n turn it on.)
        const int64_t direct_max = tn_direct_rows();
        if (D.rows <= direct_max && (int64_t)gy * gz >= 16) nblk = 1;
        all_direct = all_direct && nblk == 1;
        int64_t rpw = (D.rows + nblk * 4 - 1) / (nblk * 4);
        rpw = (rpw + 1) & ~(int64_t)1;
        const int64_t need = nblk * gy * gz * FRAG + nblk * gy * 32 * MT;
        if (need > D.ws_floats) return KGW_E_RANGE;
        T.A = D.A; T.B = D.B; T.C = D.C; T.colsum = D.colsum; T.ws = D.ws;
        T.ws_cs = D.colsum ? D.ws + nblk * gy * gz * FRAG : nullptr;
        T.rows_dev = D.rows_dev;
        T.lda = D.lda; T.ldb = D.ldb; T.rows = D.rows; T.rpw = rpw;
        T.c_rs = D.c_t ? 1 : D.ldc; T.c_cs = D.c_t ? D.ldc : 1; T.cs_ld = D.cs_ld;
        T.M = D.M; T.N = D.N; T.nblk = (int)nblk; T.blk0 = blk; T.gy = gy; T.gz = gz; T.cs_rep = D.cs_rep;
        // (round 5: the 64 x 64-per-wavefront tiling runs on the bf16 pipe with three exact pieces per operand, tn_rows_split3;
        //  KGW_TN_SPLIT=0: the fp32 pipe as before)
        T.pad_ = (MT == 2 && NT == 2 && tn_split_on() && (D.lda & 1) == 0 && (D.ldb & 1) == 0) ? 1 : 0;
        if (defer) {
            // (the fused consumer walks the gradient tensor in its own linear order: it must be dense)
            if (D.ldc != (D.c_t ? D.M : D.N) || (D.colsum && D.cs_rep != 1)) return KGW_E_UNSUPPORTED;
            KgwGradSrc& W = defer[2 * q];
            KgwGradSrc& Bc = defer[2 * q + 1];
            W = KgwGradSrc{};
            Bc = KgwGradSrc{};
            if (nblk > 1) {
                W.kind = KGW_GRAD_TN; W.nblk = (int)nblk; W.ws = D.ws; W.M = D.M; W.N = D.N; W.MT = MT; W.NT = NT; W.gy = gy; W.gz = gz;
                W.c_transposed = D.c_t ? 1 : 0;
                if (D.colsum) {
                    Bc = W;
                    Bc.kind = KGW_GRAD_TN_COLSUM; Bc.ws = T.ws_cs;
                }
            }
        }
        blk += (int)nblk;
        gy_max = gy > gy_max ? gy : gy_max;
        gz_max = gz > gz_max ? gz : gz_max;
    }
    if (plan) { plan->J = J; plan->blk = blk; plan->gy_max = gy_max; plan->gz_max = gz_max; plan->all_direct = all_direct; return KGW_OK; }
    const size_t lds_bytes = (size_t)(2 * FRAG + 4 * 32 * MT) * sizeof(float);
    auto kern = k_tn_gemm<MT, NT>;
    static KgwPerDevice attr_once;
    if (lds_bytes > 64 * 1024 && attr_once.need()) {
        KGW_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    if (ride && !ride->valid) ride = nullptr;
    bool rode = false;
    if constexpr (MT == 2 && NT == 2) {
        if (ride) {         // the pending second launch of an earlier product group: its blocks in front of this product's
            const TnRideIdx X{tn_reduce_plan_blocks(*ride), ride->gy_max, ride->gz_max, blk, gy_max, gz_max};
            static KgwPerDevice attr_ride;
            if (lds_bytes > 64 * 1024 && attr_ride.need()) {
                KGW_HIP(hipFuncSetAttribute((const void*)k_tn_gemm_ride, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            }
            k_tn_gemm_ride<<<X.n_rd + blk * gy_max * gz_max, 256, lds_bytes, st>>>(J, ride->J, X);
            KGW_LAUNCH_CHECK();
            rode = true;
        }
    }
    if (!rode) {
        if (ride) {
            k_tn_reduce<2, 2><<<dim3(TN22_FRAG / 64, ride->gy_max, ride->gz_max * ride->n), 256, 0, st>>>(ride->J, ride->gz_max);
            KGW_LAUNCH_CHECK();
        }
        kern<<<dim3((unsigned)blk, gy_max, gz_max), 256, lds_bytes, st>>>(J);
        KGW_LAUNCH_CHECK();
    }
    if (all_direct || defer) return KGW_OK;            // every product wrote its result itself / the sums are taken later
    k_tn_reduce<MT, NT><<<dim3(FRAG / 64, gy_max, gz_max * n), 256, 0, st>>>(J, gz_max);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

template <int MT, int NT>
int launch_tn(const float* A, int64_t lda, int M, const float* B, int64_t ldb, int N, int64_t rows, float* C,
              int64_t ldc, bool c_t, float* colsum, int cs_rep, int64_t cs_ld, float* ws, int64_t ws_floats,
              const int32_t* rows_dev, hipStream_t st, KgwGradSrc* defer = nullptr, const TnReducePlan* ride = nullptr) {
    const TnDesc d{A, lda, M, B, ldb, N, rows, C, ldc, c_t, colsum, cs_rep, cs_ld, ws, ws_floats, rows_dev};
    return launch_tn_jobs<MT, NT>(&d, 1, st, defer, nullptr, ride);
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int64_t kgw_tn_direct_rows(int64_t rows) {
    const int64_t was = tn_direct_rows();
    if (rows >= 0) g_tn_direct = rows;
    return was;
}

extern "C" int kgw_tn_split(int on) {
    const int was = tn_split_on() ? 1 : 0;
    if (on >= 0) g_tn_split = on ? 1 : 0;
    return was;
}

extern "C" int64_t kgw_tn_gemm_workspace_floats(int64_t rows, int M, int N) {
    // upper bound over every tiling the dispatcher may choose
    const int64_t gy1 = 4 * ((M + 127) / 128), gz1 = 4 * ((N + 127) / 128);   // tiles, rounded to the widest tiling
    int64_t nblk = (rows + 255) / 256;
    if (nblk > 1024) nblk = 1024;
    if (nblk < 1) nblk = 1;
    return nblk * gy1 * gz1 * 1024 + nblk * gy1 * 32 + 4096;   // 1024 floats per 32x32 tile per row-block
}

static int tn_gemm_ex(const float* A, int64_t lda, int32_t M, const float* B, int64_t ldb, int32_t N,
                      int64_t rows, float* C, int64_t ldc, int32_t c_transposed, float* colsum_a,
                      int32_t colsum_repeat, int64_t colsum_ld, float* workspace, int64_t workspace_floats,
                      const int32_t* rows_dev, kgw_stream_t stream_, KgwGradSrc* defer, const TnReducePlan* ride = nullptr) {
    if (!A || !B || !C || !workspace) return KGW_E_NULL;
    if (M <= 0 || N <= 0 || rows <= 0 || lda < M || ldb < N || ldc < (c_transposed ? M : N)) return KGW_E_RANGE;
    if (colsum_a && (colsum_repeat < 1 || (colsum_repeat > 1 && colsum_ld < M))) return KGW_E_RANGE;
    hipStream_t st = (hipStream_t)stream_;
    const bool ct = c_transposed != 0;
    const int rep = colsum_a ? colsum_repeat : 0;
    auto aligned8 = [](const void* p) { return ((uintptr_t)p & 7) == 0; };
    const bool a2 = (M % 2 == 0) && (lda % 2 == 0) && aligned8(A) && M >= 64;          // float2 per lane feeds two column tiles
    const bool b2 = (N % 2 == 0) && (ldb % 2 == 0) && aligned8(B) && N >= 64;
    const bool b4 = (N % 4 == 0) && (ldb % 4 == 0) && aligned16(B) && N >= 128;
    // 64x64 accumulators per wavefront (MT = NT = 2) and up to two blocks per CU rather than one 128x128 accumulator:
    // a quarter of the per-block LDS reduction / partial-slab traffic and twice the row blocks in flight -- 51 vs 72 us
    // at 123 k x 128 x 128, 21 vs 26 us at 20 k rows (each A / B element is read by two blocks, the second time from L2)
#define KGW_TN_ARGS A, lda, M, B, ldb, N, rows, C, ldc, ct, colsum_a, rep, colsum_ld, workspace, workspace_floats, rows_dev, st, defer, ride
    if (a2 && b2) return launch_tn<2, 2>(KGW_TN_ARGS);
    if (a2)       return launch_tn<2, 1>(KGW_TN_ARGS);       // narrow B (the 20-wide SNP feature layer)
    if (b4)       return launch_tn<1, 4>(KGW_TN_ARGS);       // narrow A (d a_src of a few relations)
    if (b2)       return launch_tn<1, 2>(KGW_TN_ARGS);
    return launch_tn<1, 1>(KGW_TN_ARGS);
#undef KGW_TN_ARGS
}

extern "C" int kgw_tn_gemm_ex(const float* A, int64_t lda, int32_t M, const float* B, int64_t ldb, int32_t N,
                              int64_t rows, float* C, int64_t ldc, int32_t c_transposed, float* colsum_a,
                              int32_t colsum_repeat, int64_t colsum_ld, float* workspace, int64_t workspace_floats,
                              const int32_t* rows_dev, kgw_stream_t stream_) {
    return tn_gemm_ex(A, lda, M, B, ldb, N, rows, C, ldc, c_transposed, colsum_a, colsum_repeat, colsum_ld, workspace,
                      workspace_floats, rows_dev, stream_, nullptr);
}

// The product's first launch only: the row blocks' partial sums stay in the workspace and src[0] (the product) / src[1] (the
// column sums) say how kgw_adam_fused finds them.  A product with a single row block is complete (kind KGW_GRAD_DIRECT).
extern "C" int kgw_tn_gemm_partial(const float* A, int64_t lda, int32_t M, const float* B, int64_t ldb, int32_t N,
                                   int64_t rows, float* C, int64_t ldc, int32_t c_transposed, float* colsum_a,
                                   float* workspace, int64_t workspace_floats, const int32_t* rows_dev, KgwGradSrc* src,
                                   kgw_stream_t stream_) {
    if (!src) return KGW_E_NULL;
    return tn_gemm_ex(A, lda, M, B, ldb, N, rows, C, ldc, c_transposed, colsum_a, 1, M, workspace, workspace_floats, rows_dev,
                      stream_, src);
}

// kgw_tn_gemm_partial with the pending second launch of an earlier product group (ride_in, nullable) as blocks of its own launch
// -- or, where the product does not run on the 64 x 64-per-wavefront tiling, as a launch of its own just ahead of it: the plan is
// consumed either way.
extern "C" int kgw_tn_gemm_partial_ride(const float* A, int64_t lda, int32_t M, const float* B, int64_t ldb, int32_t N,
                                        int64_t rows, float* C, int64_t ldc, int32_t c_transposed, float* colsum_a,
                                        float* workspace, int64_t workspace_floats, const int32_t* rows_dev, KgwGradSrc* src,
                                        const KgwTnReducePlan* ride_in, kgw_stream_t stream_) {
    if (!src) return KGW_E_NULL;
    return tn_gemm_ex(A, lda, M, B, ldb, N, rows, C, ldc, c_transposed, colsum_a, 1, M, workspace, workspace_floats, rows_dev,
                      stream_, src, (const TnReducePlan*)ride_in);
}

static int tn_gemm_multi(int32_t n_jobs, const KgwTnJob* jobs, kgw_stream_t stream_, KgwGradSrc* defer) {
    if (n_jobs == 0) return KGW_OK;
    if (!jobs) return KGW_E_NULL;
    if (n_jobs < 0 || n_jobs > TN_MAX_JOBS) return KGW_E_RANGE;
    TnDesc d[TN_MAX_JOBS];
    auto aligned8 = [](const void* p) { return ((uintptr_t)p & 7) == 0; };
    for (int q = 0; q < n_jobs; ++q) {
        const KgwTnJob& j = jobs[q];
        if (!j.A || !j.B || !j.C || !j.workspace) return KGW_E_NULL;
        if (j.M <= 0 || j.N <= 0 || j.rows <= 0 || j.lda < j.M || j.ldb < j.N || j.ldc < (j.c_transposed ? j.M : j.N)) return KGW_E_RANGE;
        if (j.colsum_a && (j.colsum_repeat < 1 || (j.colsum_repeat > 1 && j.colsum_ld < j.M))) return KGW_E_RANGE;
        // every job runs on the 64 x 64-per-wavefront tiling: float2 operand loads
        if ((j.M & 1) || (j.lda & 1) || !aligned8(j.A) || (j.N & 1) || (j.ldb & 1) || !aligned8(j.B)) return KGW_E_UNSUPPORTED;
        d[q] = TnDesc{j.A, j.lda, j.M, j.B, j.ldb, j.N, j.rows, j.C, j.ldc, j.c_transposed != 0, j.colsum_a,
                      j.colsum_a ? j.colsum_repeat : 0, j.colsum_ld, j.workspace, j.workspace_floats, j.rows_dev};
    }
    return launch_tn_jobs<2, 2>(d, n_jobs, (hipStream_t)stream_, defer);
}

extern "C" int kgw_tn_gemm_multi(int32_t n_jobs, const KgwTnJob* jobs, kgw_stream_t stream_) {
    return tn_gemm_multi(n_jobs, jobs, stream_, nullptr);
}

// ... and of kgw_tn_gemm_multi: src holds 2 n_jobs records, [product, column sums] of every job
extern "C" int kgw_tn_gemm_multi_partial(int32_t n_jobs, const KgwTnJob* jobs, KgwGradSrc* src, kgw_stream_t stream_) {
    if (n_jobs > 0 && !src) return KGW_E_NULL;
    return tn_gemm_multi(n_jobs, jobs, stream_, src);
}

extern "C" int kgw_tn_gemm(const float* A, int64_t lda, int32_t M, const float* B, int64_t ldb, int32_t N,
                           int64_t rows, float* C, int64_t ldc, float* colsum_a, float* workspace,
                           int64_t workspace_floats, kgw_stream_t stream_) {
    return kgw_tn_gemm_ex(A, lda, M, B, ldb, N, rows, C, ldc, 0, colsum_a, 1, M, workspace, workspace_floats, nullptr, stream_);
}

// ======================================================================================================
// kgw_linear: Y[rows,N] = act( X[rows,K] * Wop + bias ) (* relu-mask), fp32 MFMA, LDS-tiled.
//   Wop = W^T with W [N,K] row-major (nn.Linear forward, kgwas/model.py:13-21; conv.py:138,142), or
//   Wop = W   with W [K,N] row-major (the dX = dY * W product of the same layers' backward).
// Block = 128 rows x 128 cols, BK = 32, 4 wavefronts (32 rows x 128 cols each = four 32x32x2 MFMA tiles);
// the next K-tile is fetched into registers while the current one is consumed from LDS (row stride 33
// floats: the 32 lanes of an MFMA operand read hit 32 different banks).
// ======================================================================================================
namespace {

constexpr int LBM = 128, LBN = 128, LBK = 32, LPAD = 33;

struct LinArgs {
    const float* X; int64_t ldx;
    const float* W; int64_t ldw;
    const float* bias;      // [N] or null
    const float* mask;      // [rows, ldm]: output multiplied by (mask > 0), or null
    int64_t ldm;
    float* Y; int64_t ldy;
    int64_t rows; int K, N;
    int relu, w_kn;
    const int32_t* rows_dev;   // device: actual row count (<= rows); rows beyond it are written as zeros
};

// rows the batch really has; the rest of the static capacity is padding: not computed, written as zeros
__device__ __forceinline__ int64_t lin_rows_eff(const LinArgs& a) {
    if (!a.rows_dev) return a.rows;
    const int64_t r = *a.rows_dev;
    return r < 0 ? 0 : (r < a.rows ? r : a.rows);
}

__device__ __forceinline__ void lin_zero_padding(const LinArgs& a, int64_t rows_eff, int64_t tid, int64_t nthreads) {
    const int64_t npad = a.rows - rows_eff;
    if (npad <= 0) return;
    if ((a.N & 3) == 0 && (a.ldy & 3) == 0 && ((uintptr_t)a.Y & 15) == 0) {
        const int n4 = a.N >> 2;
        for (int64_t q = tid; q < npad * n4; q += nthreads)
            *(float4*)(a.Y + (rows_eff + q / n4) * a.ldy + (q % n4) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int64_t q = tid; q < npad * a.N; q += nthreads) a.Y[(rows_eff + q / a.N) * a.ldy + q % a.N] = 0.f;
    }
}

__global__ void __launch_bounds__(256, 2) k_linear(LinArgs a_) {
    LinArgs a = a_;
    {
        const int64_t re = lin_rows_eff(a_);
        lin_zero_padding(a_, re, ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x,
                         (int64_t)gridDim.x * gridDim.y * 256);
        a.rows = re;
        if ((int64_t)blockIdx.x * LBM >= re) return;
    }
    __shared__ float Xs[LBM * LPAD];
    __shared__ float Ws[LBN * LPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * LBM;
    const int n0 = blockIdx.y * LBN;
    const int li = lane & 31, lk = lane >> 5;

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    float4 xr[4], wr[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + 256 * j;
            {   // X tile: 128 rows x 32 k, float4 along k
                const int row = idx >> 3, kq = (idx & 7) * 4;
                const int64_t r = r0 + row;
                const bool ok = (r < a.rows) && (k0 + kq < a.K);
                const float* p = a.X + (ok ? r : 0) * a.ldx + (ok ? k0 + kq : 0);
                const float4 v = *(const float4*)p;
                xr[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (!a.w_kn) {   // W [N,K]: 128 n x 32 k, float4 along k
                const int n = idx >> 3, kq = (idx & 7) * 4;
                const bool ok = (n0 + n < a.N) && (k0 + kq < a.K);
                const float* p = a.W + (int64_t)(ok ? n0 + n : 0) * a.ldw + (ok ? k0 + kq : 0);
                const float4 v = *(const float4*)p;
                wr[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {         // W [K,N]: 32 k x 128 n, float4 along n
                const int k = idx >> 5, nq = (idx & 31) * 4;
                const bool ok = (k0 + k < a.K) && (n0 + nq < a.N);
                const float* p = a.W + (int64_t)(ok ? k0 + k : 0) * a.ldw + (ok ? n0 + nq : 0);
                const float4 v = *(const float4*)p;
                wr[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = tid + 256 * j;
            {
                const int row = idx >> 3, kq = (idx & 7) * 4;
                float* d = Xs + row * LPAD + kq;
                d[0] = xr[j].x; d[1] = xr[j].y; d[2] = xr[j].z; d[3] = xr[j].w;
            }
            if (!a.w_kn) {
                const int n = idx >> 3, kq = (idx & 7) * 4;
                float* d = Ws + n * LPAD + kq;
                d[0] = wr[j].x; d[1] = wr[j].y; d[2] = wr[j].z; d[3] = wr[j].w;
            } else {
                const int k = idx >> 5, nq = (idx & 31) * 4;
                Ws[(nq + 0) * LPAD + k] = wr[j].x; Ws[(nq + 1) * LPAD + k] = wr[j].y;
                Ws[(nq + 2) * LPAD + k] = wr[j].z; Ws[(nq + 3) * LPAD + k] = wr[j].w;
            }
        }
    };

    fetch(0);
    for (int k0 = 0; k0 < a.K; k0 += LBK) {
        __syncthreads();                 // previous tile fully consumed
        stage();
        __syncthreads();
        if (k0 + LBK < a.K) fetch(k0 + LBK);          // in flight during the MFMAs below
        const float* xa = Xs + (wave * 32 + li) * LPAD + lk;
        const float* wb = Ws + li * LPAD + lk;
#pragma unroll
        for (int kp = 0; kp < LBK / 2; ++kp) {
            const float av = xa[2 * kp];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[t * 32 * LPAD + 2 * kp], acc[t], 0, 0, 0);
        }
    }
    // epilogue: bias, ReLU, mask; lanes 0-31 of a register write 32 consecutive floats of one row
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = n0 + t * 32 + li;
        if (col >= a.N) continue;
        const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t r = r0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
            if (r >= a.rows) continue;
            float v = acc[t][e] + bv;
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.mask) v = (a.mask[r * a.ldm + col] > 0.f) ? v : 0.f;
            a.Y[r * a.ldy + col] = v;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// K <= 128, N <= 128 (every Linear of the feature MLPs, forward and dX): persistent 8-wavefront blocks keep
// the weight matrix RESIDENT in LDS (staged once per block) and stream 256-row X tiles through a second LDS
// buffer in K-chunks of KC, the next chunk prefetched into registers while the MFMAs of the current one run.
// MFMA step kp of a chunk multiplies k = lk*(KC/2) + kp (a permutation of the K order -- the sum is the same
// set of products), so each lane's operands are CONTIGUOUS in LDS: one ds_read_b128 feeds four MFMA steps
// (5 LDS reads per 16 MFMAs instead of 20).  Row strides of 4 mod 64 floats keep those reads conflict free.
// Epilogue: accumulators -> the wavefront's own 32 rows of the X buffer -> full-row float4 stores.
// ------------------------------------------------------------------------------------------------------
namespace {

constexpr int WST = 132;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// RT = rows per tile: 256 (each wavefront 32 rows x all 128 columns) for tall inputs; 64 (2 row groups x 4 column
// groups of 32) for mid-size inputs, so that 8k-32k rows still spread over every CU.
template <int KC, bool WKN, int RT>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) k_linear_wres(LinArgs a_) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    LinArgs a = a_;
    {
        const int64_t re = lin_rows_eff(a_);
        lin_zero_padding(a_, re, (int64_t)blockIdx.x * 512 + threadIdx.x, (int64_t)gridDim.x * 512);
        a.rows = re;
        if ((int64_t)blockIdx.x * RT >= re) return;      // (before any barrier: the whole block leaves)
    }
    constexpr int XST = KC + 4;
    float* Wl = lds;                       // [128 n][WST]
    float* Xs = lds + 128 * WST;           // [RT rows][XST]
    constexpr int WRG = RT / 32, WCG = 8 / WRG, CT = 4 / WCG;   // wave grid (rows x column groups), col tiles per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int nch = (a.K + KC - 1) / KC;   // 1 or 2 chunks (K <= 128)
    // the first X chunk is requested before the weights are staged: both latencies overlap
    const int64_t ntiles = (a.rows + RT - 1) / RT;
    constexpr int F4 = RT * KC / 4 / 512;              // float4 per thread per chunk
    f32x4 xr[F4];
#define KGW_FETCH(TILE, CH)                                                                            \
    _Pragma("unroll") for (int j = 0; j < F4; ++j) {                                                   \
        const int idx = tid + 512 * j;                                                                 \
        const int row = idx / (KC / 4), kq = (idx % (KC / 4)) * 4;                                     \
        int64_t r = (TILE) * RT + row;                                                                 \
        if (r >= a.rows) r = a.rows - 1;           /* clamped rows: outputs never stored */           \
        int k = (CH) * KC + kq;                                                                        \
        if (k > a.K - 4) k = a.K - 4;              /* beyond K the staged weights are zero */         \
        xr[j] = *(const f32x4*)(a.X + r * a.ldx + k);                                                   \
    }
    int64_t tile = blockIdx.x;
    if (tile < ntiles) { KGW_FETCH(tile, 0) }
    // stage W once (zero outside [N, K]); a thread's 8 loads are all in flight before its first LDS write
    {
        f32x4 wv[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + 512 * it;
            wv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!WKN) {
                const int n = idx >> 5, k4 = (idx & 31) * 4;
                if (n < a.N && k4 < a.K) wv[it] = *(const f32x4*)(a.W + (int64_t)n * a.ldw + k4);
            } else {
                const int k = idx >> 5, n4 = (idx & 31) * 4;
                if (k < a.K && n4 < a.N) wv[it] = *(const f32x4*)(a.W + (int64_t)k * a.ldw + n4);
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + 512 * it;
            if (!WKN) {
                *(f32x4*)(Wl + (idx >> 5) * WST + (idx & 31) * 4) = wv[it];
            } else {
                const int k = idx >> 5, n4 = (idx & 31) * 4;
                Wl[(n4 + 0) * WST + k] = wv[it].x; Wl[(n4 + 1) * WST + k] = wv[it].y;
                Wl[(n4 + 2) * WST + k] = wv[it].z; Wl[(n4 + 3) * WST + k] = wv[it].w;
            }
        }
    }
    const int rg = wave % WRG, cg = wave / WRG;
    const float* wb = Wl + (cg * CT * 32 + li) * WST + lk * (KC / 2);
    float* slice = Xs + rg * 32 * XST;                 // this wavefront's 32 rows (private when WCG == 1)
    const float* xa = slice + li * XST + lk * (KC / 2);
    constexpr int Q = KC / 8;                          // groups of four MFMA steps per chunk
    const float relu_lo = a.relu ? 0.f : -__builtin_inff();
    for (; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        for (int ch = 0; ch < nch; ++ch) {
            __syncthreads();                            // previous chunk / epilogue done (and W staged, first time)
#pragma unroll
            for (int j = 0; j < F4; ++j) {
                const int idx = tid + 512 * j;
                *(f32x4*)(Xs + (idx / (KC / 4)) * XST + (idx % (KC / 4)) * 4) = xr[j];
            }
            __syncthreads();
            // prefetch the next chunk (same tile or the block's next tile) while computing
            {
                const bool same = ch + 1 < nch;
                const int64_t nt = same ? tile : tile + gridDim.x;
                const int nc = same ? ch + 1 : 0;
                if (nt < ntiles) { KGW_FETCH(nt, nc) }
            }
            const float* wk = wb + ch * KC;
            f32x4 af[2], bf[2][CT];
            af[0] = *(const f32x4*)xa;
#pragma unroll
            for (int t = 0; t < CT; ++t) bf[0][t] = *(const f32x4*)(wk + t * 32 * WST);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int cur = q & 1;
                if (q + 1 < Q) {
                    af[cur ^ 1] = *(const f32x4*)(xa + 4 * (q + 1));
#pragma unroll
                    for (int t = 0; t < CT; ++t) bf[cur ^ 1][t] = *(const f32x4*)(wk + t * 32 * WST + 4 * (q + 1));
                }
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur].x, bf[cur][t].x, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur].y, bf[cur][t].y, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur].z, bf[cur][t].z, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur].w, bf[cur][t].w, acc[t], 0, 0, 0);
            }
        }
        if constexpr (WCG == 1) {
        // epilogue through the wavefront's own rows of Xs (nobody else touches them before the next barrier)
        constexpr int CP = (KC >= 64) ? 64 : 32;        // columns per pass
        constexpr int LR = CP / 4;                      // lanes per output row
        constexpr int RP = 64 / LR;                     // rows per store instruction
        const int64_t rbase = tile * RT + rg * 32;
#pragma unroll
        for (int pass = 0; pass < 128 / CP; ++pass) {
            if (pass * CP >= a.N) continue;
#pragma unroll
            for (int tt = 0; tt < CP / 32; ++tt) {
                const int t = pass * (CP / 32) + tt;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    slice[((e & 3) + 8 * (e >> 2) + 4 * lk) * XST + tt * 32 + li] = acc[t][e];
            }
            const int c = (lane % LR) * 4, col = pass * CP + c;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias && col < a.N) bv = *(const float4*)(a.bias + col);
            // all mask rows of the pass are requested before the first store: a load between two stores would make
            // every store wait for the previous one (vmcnt counts both)
            float4 mk[32 / RP];
            if (a.mask) {
#pragma unroll
                for (int it = 0; it < 32 / RP; ++it) {
                    int64_t rr = rbase + it * RP + lane / LR;
                    if (rr >= a.rows) rr = a.rows - 1;
                    mk[it] = *(const float4*)(a.mask + rr * a.ldm + (col < a.N ? col : 0));
                }
            } else {
#pragma unroll
                for (int it = 0; it < 32 / RP; ++it) mk[it] = make_float4(1.f, 1.f, 1.f, 1.f);
            }
#pragma unroll
            for (int it = 0; it < 32 / RP; ++it) {
                const int row = it * RP + lane / LR;
                float4 v = *(const float4*)(slice + row * XST + c);
                const int64_t rr = rbase + row;
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                v.x = fmaxf(v.x, relu_lo); v.y = fmaxf(v.y, relu_lo); v.z = fmaxf(v.z, relu_lo); v.w = fmaxf(v.w, relu_lo);   // (no branch between stores)
                v.x = mk[it].x > 0.f ? v.x : 0.f; v.y = mk[it].y > 0.f ? v.y : 0.f;
                v.z = mk[it].z > 0.f ? v.z : 0.f; v.w = mk[it].w > 0.f ? v.w : 0.f;
                if (rr < a.rows && col < a.N) *(float4*)(a.Y + rr * a.ldy + col) = v;
            }
        }
        } else {
        // the A rows are shared by WCG wavefronts: store straight from the accumulators (32 lanes = one 128-B
        // row segment); all mask loads are issued before the first store
        static_assert(CT == 1 || WCG == 1, "direct epilogue handles one column tile per wavefront");
        const int col = cg * 32 + li;
        const int64_t rbase = tile * RT + rg * 32 + 4 * lk;
        if (col < a.N) {
            const float bv = a.bias ? a.bias[col] : 0.f;
            float mv[16];
            if (a.mask) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int64_t rr = rbase + (e & 3) + 8 * (e >> 2);
                    if (rr >= a.rows) rr = a.rows - 1;
                    mv[e] = a.mask[rr * a.ldm + col];
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t rr = rbase + (e & 3) + 8 * (e >> 2);
                float v = fmaxf(acc[0][e] + bv, relu_lo);
                if (a.mask) v = mv[e] > 0.f ? v : 0.f;
                if (rr < a.rows) a.Y[rr * a.ldy + col] = v;
            }
        }
        }
    }
}

#undef KGW_FETCH

// ------------------------------------------------------------------------------------------------------
// K == 128, N == 128 (the hidden layers of the feature MLPs, forward and dX): the weight matrix lives in REGISTERS.
// One wavefront per SIMD (512 registers): 192 of them hold columns 0-95 of W as MFMA operands (staged once per block
// through LDS; the last 32 columns are read from LDS a step ahead), each wavefront streams 32-row tiles of X
// straight from global memory into the other operand
// -- lane (i, h) owns the contiguous half row X[r0 + i][64 h .. 64 h + 63], the K order being permuted so that MFMA
// step s multiplies k = 64 h + s -- and refills the tile in place with the wavefront's NEXT tile, half a row (eight
// float4 = one 128-B line per lane) at a time: the first half right after its last use, the second after the tile's
// stores (the loads have ~8 k cycles to land either way).  No LDS traffic, no barrier and
// no waitcnt on a fresh load inside the MFMA stream: the matrix pipe sees 256 back-to-back MFMAs per tile over four
// independent accumulators.  ReLU-mask rows (dX) are fetched 16 at a time under the MFMAs and kept as bits.
// ------------------------------------------------------------------------------------------------------
template <bool WKN, bool MASK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_linear_wreg(LinArgs a_) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    LinArgs a = a_;
    {
        const int64_t re = lin_rows_eff(a_);
        lin_zero_padding(a_, re, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
        a.rows = re;
        if ((int64_t)blockIdx.x * 4 * 32 >= re) return;      // (before any barrier: the whole block leaves)
    }
    float* Wl = lds;                                         // [128 n][WST], k contiguous
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int ntiles = (int)((a.rows + 31) / 32);
    const int nw = (int)gridDim.x * 4;
    int tile = (int)blockIdx.x * 4 + wave;
    f32x4 xa[16];
    {
        int64_t r = (int64_t)(tile < ntiles ? tile : ntiles - 1) * 32 + li;
        if (r >= a.rows) r = a.rows - 1;
        const float* xp = a.X + r * a.ldx + lk * 64;
#pragma unroll
        for (int q = 0; q < 16; ++q) xa[q] = *(const f32x4*)(xp + 4 * q);
    }
    {   // stage W: all 16 loads of a thread in flight before the first LDS write (one round trip, not sixteen)
        f32x4 wv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * it;
            if (!WKN) wv[it] = *(const f32x4*)(a.W + (int64_t)(idx >> 5) * a.ldw + (idx & 31) * 4);
            else wv[it] = *(const f32x4*)(a.W + (int64_t)(idx & 127) * a.ldw + (idx >> 7) * 4);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * it;
            if (!WKN) {
                *(f32x4*)(Wl + (idx >> 5) * WST + (idx & 31) * 4) = wv[it];
            } else {
                const int k = idx & 127, n4 = (idx >> 7) * 4;     // lanes along k: conflict-free transposing writes
                Wl[(n4 + 0) * WST + k] = wv[it].x; Wl[(n4 + 1) * WST + k] = wv[it].y;
                Wl[(n4 + 2) * WST + k] = wv[it].z; Wl[(n4 + 3) * WST + k] = wv[it].w;
            }
        }
    }
    if (tid < 128) Wl[128 * WST + tid] = a.bias ? a.bias[tid] : 0.f;
    __syncthreads();
    // columns 0-95 of W as registers; the last 32 columns stay in LDS (one ds_read_b128 per four MFMA steps, fetched a
    // step ahead): all 256 would leave the compiler a handful of registers short of 512
    f32x4 bw[3][16];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) bw[t][q] = *(const f32x4*)(Wl + (t * 32 + li) * WST + lk * 64 + 4 * q);
    const float* w3 = Wl + (96 + li) * WST + lk * 64;
    // W is the MFMA's A operand (32 output columns x 2 k) and the X tile its B operand (2 k x 32 rows): the accumulator
    // registers of lane (j, h) are then FOUR CONSECUTIVE output columns 32 t + 8 g + 4 h .. + 3 of row j, so the epilogue
    // is 16 float4 stores (and 16 float4 mask loads) per tile instead of 64 scalar ones
    const float* bl = Wl + 128 * WST + 4 * lk;             // bias staged behind W
    const float lo = a.relu ? 0.f : -__builtin_inff();
    for (; tile < ntiles; tile += nw) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        const float* xn;
        {
            int nt = tile + nw;
            if (nt >= ntiles) nt = ntiles - 1;               // last round: a harmless re-read
            int64_t r = (int64_t)nt * 32 + li;
            if (r >= a.rows) r = a.rows - 1;
            xn = a.X + r * a.ldx + lk * 64;
        }
        const int64_t row = (int64_t)tile * 32 + li;         // this lane's output row
        const bool live = row < a.rows;
        const float* mp = nullptr;                            // mask row (the last row for lanes past the end: never stored)
        if (MASK) mp = a.mask + (live ? row : a.rows - 1) * a.ldm + 4 * lk;
        unsigned mb[2] = {0xffffffffu, 0xffffffffu};
        f32x4 mv[2][4];
        f32x4 b3n = *(const f32x4*)w3;
#define KGW_MASK_FETCH(T)                                                                                 \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) mv[(T) & 1][g] = *(const f32x4*)(mp + (T) * 32 + 8 * g);
#define KGW_MASK_BITS(T)                                                                                  \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                  \
            const int b0 = ((T) & 1) * 16 + 4 * g;                                                        \
            if (!(mv[(T) & 1][g].x > 0.f)) mb[(T) >> 1] &= ~(1u << (b0 + 0));                             \
            if (!(mv[(T) & 1][g].y > 0.f)) mb[(T) >> 1] &= ~(1u << (b0 + 1));                             \
            if (!(mv[(T) & 1][g].z > 0.f)) mb[(T) >> 1] &= ~(1u << (b0 + 2));                             \
            if (!(mv[(T) & 1][g].w > 0.f)) mb[(T) >> 1] &= ~(1u << (b0 + 3));                             \
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (MASK) {                                       // column block t: fetched at step t (0, 1) / t + 4 (6, 7), folded five steps on
                if (q == 0) { KGW_MASK_FETCH(0) }
                if (q == 1) { KGW_MASK_FETCH(1) }
                if (q == 5) { KGW_MASK_BITS(0) }
                if (q == 6) { KGW_MASK_BITS(1) KGW_MASK_FETCH(2) }
                if (q == 7) { KGW_MASK_FETCH(3) }
                if (q == 11) { KGW_MASK_BITS(2) }
                if (q == 12) { KGW_MASK_BITS(3) }
            }
            const f32x4 b3 = b3n;
            if (q + 1 < 16) b3n = *(const f32x4*)(w3 + 4 * (q + 1));
#define KGW_WREG_STEP(C)                                                                                  \
            _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[t][q].C, xa[q].C, acc[t], 0, 0, 0);      \
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b3.C, xa[q].C, acc[3], 0, 0, 0);
            KGW_WREG_STEP(x) KGW_WREG_STEP(y) KGW_WREG_STEP(z) KGW_WREG_STEP(w)
#undef KGW_WREG_STEP
            // the next tile's float4s, in place, half a row (one 128-B line per lane) at a time: the eight loads of a line
            // are issued back to back so that the line is fetched from L2 once
            if (q == 7) {
                __builtin_amdgcn_sched_barrier(0);           // (keeps the refill below its registers' last use)
#pragma unroll
                for (int qq = 0; qq < 8; ++qq) xa[qq] = *(const f32x4*)(xn + 4 * qq);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef KGW_MASK_FETCH
#undef KGW_MASK_BITS
        if (live) {
            float* yp = a.Y + row * a.ldy + 4 * lk;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *(const f32x4*)(bl + t * 32 + 8 * g);
                    f32x4 v;
                    v.x = acc[t][4 * g + 0] + b4.x; v.y = acc[t][4 * g + 1] + b4.y;
                    v.z = acc[t][4 * g + 2] + b4.z; v.w = acc[t][4 * g + 3] + b4.w;
                    v.x = fmaxf(v.x, lo); v.y = fmaxf(v.y, lo); v.z = fmaxf(v.z, lo); v.w = fmaxf(v.w, lo);   // (ReLU without a branch per store)
                    if (MASK) {
                        const unsigned m4 = mb[t >> 1] >> (((t & 1) << 4) + 4 * g);
                        v.x = (m4 & 1u) ? v.x : 0.f; v.y = (m4 & 2u) ? v.y : 0.f;
                        v.z = (m4 & 4u) ? v.z : 0.f; v.w = (m4 & 8u) ? v.w : 0.f;
                    }
                    *(f32x4*)(yp + t * 32 + 8 * g) = v;
                }
            }
        }
        // the second half row of the next tile is requested AFTER this tile's stores: the wait for it (step 8 of the next
        // tile, in-order vmcnt) then has only loads behind it, not the stores
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qq = 8; qq < 16; ++qq) xa[qq] = *(const f32x4*)(xn + 4 * qq);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------------------
// kgw_mlp2_fwd: H2 = relu(relu(X W1^T + b1) W2^T + b2) for a NARROW first layer (K1 <= 20: the 20-wide SNP features,
// kgwas/model.py:17-20 on ~120 k sampled rows) in ONE launch -- k_linear_wreg with its X tile COMPUTED instead of loaded:
//   product 1 runs in the same orientation as product 2 (weights = MFMA A operand, the X' tile = B operand), so lane
//   (row j, half h) ends up holding h1[j][32 t + 8 g + 4 h + c] -- exactly "one row, 64 of its 128 columns" as the second
//   product's B operand wants it; only the K order differs from k_linear_wreg's, so W2 is loaded into its operand
//   registers in THAT order.  The hidden state never goes through LDS or memory on its way to the second product.
//   X' = [X | 1 | 0..] (24 wide), W1' = [W1 | b1 | 0..]: the bias of the first layer rides in the product.
//   H1 is written too when the caller wants it (the backward's ReLU mask and weight gradient read it).
// ------------------------------------------------------------------------------------------------------
struct Mlp2Args {
    const float* X; int64_t ldx;
    const float* W1; int64_t ldw1; const float* b1;
    const float* W2; int64_t ldw2; const float* b2;
    float* H1; int64_t ldh1;          // nullable
    float* H2; int64_t ldh2;
    int64_t rows; int K1;
    const int32_t* rows_dev;
    const int32_t* ids;               // nullable: row r of the input is X[ids[r]] (the loader's x[n_id] slicing folded in)
    float* Xg; int64_t ldxg;          // nullable: the gathered rows, written for the backward's weight gradient
};

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_mlp2_fwd(Mlp2Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Wl = lds;                                         // [128 n][WST], k contiguous; bias b2 behind it
    float* W1l = lds + 128 * WST + 128;                      // [128 n][24]: W1 | b1 | 0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    int64_t rows = a.rows;
    if (a.rows_dev) { const int64_t r = *a.rows_dev; rows = r < 0 ? 0 : (r < a.rows ? r : a.rows); }
    {   // padding rows of a static layout: zeros
        const int64_t npad = a.rows - rows;
        for (int64_t q = (int64_t)blockIdx.x * 256 + tid; q < npad * 32; q += (int64_t)gridDim.x * 256) {
            const int64_t r = rows + q / 32; const int c4 = (int)(q % 32) * 4;
            *(float4*)(a.H2 + r * a.ldh2 + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.H1) *(float4*)(a.H1 + r * a.ldh1 + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if ((int64_t)blockIdx.x * 4 * 32 >= rows) return;    // (before any barrier: the whole block leaves)
    }
    const int ntiles = (int)((rows + 31) / 32);
    const int nw = (int)gridDim.x * 4;
    int tile = (int)blockIdx.x * 4 + wave;
    const int K1 = a.K1;
    // this lane's part of an X' row: k = 12 lk + 0..11 as three float4 (a chunk is data, the bias slot (1,0,0,0), or zero)
    auto fetch_x = [&](int t, f32x4 (&x)[3]) {
        int64_t r = (int64_t)(t < ntiles ? t : ntiles - 1) * 32 + li;
        if (r >= rows) r = rows - 1;
        const float* xp = a.X + (a.ids ? (int64_t)a.ids[r] : r) * a.ldx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int k4 = 12 * lk + 4 * c;
            if (k4 + 4 <= K1) x[c] = *(const f32x4*)(xp + k4);
            else x[c] = f32x4{k4 == K1 ? 1.f : 0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 xn[3];
    fetch_x(tile, xn);
    {   // stage W2 (as k_linear_wreg) and W1' through LDS
        f32x4 wv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * it;
            wv[it] = *(const f32x4*)(a.W2 + (int64_t)(idx >> 5) * a.ldw2 + (idx & 31) * 4);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * it;
            *(f32x4*)(Wl + (idx >> 5) * WST + (idx & 31) * 4) = wv[it];
        }
        for (int idx = tid; idx < 128 * 24; idx += 256) {
            const int n = idx / 24, k = idx % 24;
            W1l[idx] = k < K1 ? a.W1[(int64_t)n * a.ldw1 + k] : (k == K1 ? (a.b1 ? a.b1[n] : 0.f) : 0.f);
        }
    }
    if (tid < 128) Wl[128 * WST + tid] = a.b2 ? a.b2[tid] : 0.f;
    __syncthreads();
    // operand registers.  Second product: MFMA step (q = 4 t + g, c) multiplies k = 32 t + 8 g + 4 lk + c -- the column the
    // first product leaves in accumulator element 4 g + c of tile t of this lane.
    f32x4 bw[3][16];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) bw[t][q] = *(const f32x4*)(Wl + (t * 32 + li) * WST + 32 * (q >> 2) + 8 * (q & 3) + 4 * lk);
    const float* w3 = Wl + (96 + li) * WST + 4 * lk;
    f32x4 w1[4][3];                                          // W1'[32 t + li][12 lk + 0..11]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) w1[t][c] = *(const f32x4*)(W1l + (t * 32 + li) * 24 + 12 * lk + 4 * c);
    const float* bl = Wl + 128 * WST + 4 * lk;
    for (; tile < ntiles; tile += nw) {
        f32x4 x[3] = {xn[0], xn[1], xn[2]};
        fetch_x(tile + nw, xn);                              // next tile's rows: in flight under this tile's MFMAs
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        // product 1: 12 steps x 4 column tiles
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#define KGW_MLP_STEP(C)                                                                                   \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[t][c].C, x[c].C, acc[t], 0, 0, 0);
            KGW_MLP_STEP(x) KGW_MLP_STEP(y) KGW_MLP_STEP(z) KGW_MLP_STEP(w)
#undef KGW_MLP_STEP
        }
        const int64_t row = (int64_t)tile * 32 + li;         // this lane's row
        const bool live = row < rows;
        if (a.Xg && live) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (12 * lk + 4 * c + 4 <= K1) *(f32x4*)(a.Xg + row * a.ldxg + 12 * lk + 4 * c) = x[c];
        }
        f32x4 xa[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                v.x = fmaxf(acc[t][4 * g + 0], 0.f); v.y = fmaxf(acc[t][4 * g + 1], 0.f);
                v.z = fmaxf(acc[t][4 * g + 2], 0.f); v.w = fmaxf(acc[t][4 * g + 3], 0.f);
                xa[4 * t + g] = v;
            }
        if (a.H1 && live) {
            float* hp = a.H1 + row * a.ldh1 + 4 * lk;
#pragma unroll
            for (int q = 0; q < 16; ++q) *(f32x4*)(hp + 32 * (q >> 2) + 8 * (q & 3)) = xa[q];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        // product 2: k_linear_wreg's MFMA stream
        f32x4 b3n = *(const f32x4*)w3;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x4 b3 = b3n;
            if (q + 1 < 16) b3n = *(const f32x4*)(w3 + 32 * ((q + 1) >> 2) + 8 * ((q + 1) & 3));
#define KGW_MLP_STEP(C)                                                                                   \
            _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[t][q].C, xa[q].C, acc[t], 0, 0, 0);      \
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b3.C, xa[q].C, acc[3], 0, 0, 0);
            KGW_MLP_STEP(x) KGW_MLP_STEP(y) KGW_MLP_STEP(z) KGW_MLP_STEP(w)
#undef KGW_MLP_STEP
        }
        if (live) {
            float* yp = a.H2 + row * a.ldh2 + 4 * lk;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *(const f32x4*)(bl + t * 32 + 8 * g);
                    f32x4 v;
                    v.x = fmaxf(acc[t][4 * g + 0] + b4.x, 0.f); v.y = fmaxf(acc[t][4 * g + 1] + b4.y, 0.f);
                    v.z = fmaxf(acc[t][4 * g + 2] + b4.z, 0.f); v.w = fmaxf(acc[t][4 * g + 3] + b4.w, 0.f);
                    *(f32x4*)(yp + t * 32 + 8 * g) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// k_mlp2_fwd3: the same launch with the SECOND product (128 x 128, 84 % of the multiply-adds) on the bf16 matrix pipe at fp32
// error -- the exact three-way bf16 split of kgw_gemm3.hip: six v_mfma_f32_32x32x16_bf16 per 16 k instead of eight
// v_mfma_f32_32x32x2_f32 of twice the issue time, 6 144 instead of 16 384 MFMA cycles per 32-row tile.  W2 is split once per
// block into an LDS image of MFMA operands (96 KB: [8 steps][3 pieces][4 output tiles][64 lanes] x 16 B, conflict-free
// ds_read_b128); the hidden state stays in the registers the first product leaves it in (lane = row, 64 columns) and is split
// there, 8 values per step -- MFMA step s multiplies, in lane group lk, k = 32 (s >> 1) + 16 (s & 1) + 8 e + 4 lk + c (i = 4 e + c),
// and the W2 image is packed with the same map.  Weights no longer sit in registers (W1' comes from LDS too), so a block is
// 8 wavefronts = two per SIMD instead of one.
// ------------------------------------------------------------------------------------------------------
static constexpr int M3_W2_U4 = 8 * 3 * 4 * 64;              // uint4 in the W2 operand image

__global__ void __launch_bounds__(512, 1) k_mlp2_fwd3(Mlp2Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4* W2p = (uint4*)lds;
    float* W1l = lds + M3_W2_U4 * 4;                         // [128 n][24]: W1 | b1 | 0
    float* bl = W1l + 128 * 24;                              // b2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    int64_t rows = a.rows;
    if (a.rows_dev) { const int64_t r = *a.rows_dev; rows = r < 0 ? 0 : (r < a.rows ? r : a.rows); }
    {   // padding rows of a static layout: zeros
        const int64_t npad = a.rows - rows;
        for (int64_t q = (int64_t)blockIdx.x * 512 + tid; q < npad * 32; q += (int64_t)gridDim.x * 512) {
            const int64_t r = rows + q / 32; const int c4 = (int)(q % 32) * 4;
            *(float4*)(a.H2 + r * a.ldh2 + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.H1) *(float4*)(a.H1 + r * a.ldh1 + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if ((int64_t)blockIdx.x * 8 * 32 >= rows) return;    // (before any barrier: the whole block leaves)
    }
    const int ntiles = (int)((rows + 31) / 32);
    const int nw = (int)gridDim.x * 8;
    int tile = (int)blockIdx.x * 8 + wave;
    const int K1 = a.K1;
    auto fetch_x = [&](int t, f32x4 (&x)[3]) {
        int64_t r = (int64_t)(t < ntiles ? t : ntiles - 1) * 32 + li;
        if (r >= rows) r = rows - 1;
        const float* xp = a.X + (a.ids ? (int64_t)a.ids[r] : r) * a.ldx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int k4 = 12 * lk + 4 * c;
            if (k4 + 4 <= K1) x[c] = *(const f32x4*)(xp + k4);
            else x[c] = f32x4{k4 == K1 ? 1.f : 0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 xn[3];
    fetch_x(tile, xn);
    // the W2 operand image: entry (s, p, ot, lane) = piece p of W2[32 ot + li][k(s, lk, i)], i = 0..7
    for (int idx = tid; idx < 8 * 4 * 64; idx += 512) {
        const int ln = idx & 63, ot = (idx >> 6) & 3, s_ = idx >> 8;
        const float* wp = a.W2 + (int64_t)(32 * ot + (ln & 31)) * a.ldw2 + 32 * (s_ >> 1) + 16 * (s_ & 1) + 4 * (ln >> 5);
        const f32x4 u = *(const f32x4*)wp, v = *(const f32x4*)(wp + 8);
        const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
        uint4 p1, p2, p3;
        kgw_split3x8(x, p1, p2, p3);
        uint4* o = W2p + ((s_ * 3) * 4 + ot) * 64 + ln;
        o[0] = p1; o[4 * 64] = p2; o[8 * 64] = p3;
    }
    for (int idx = tid; idx < 128 * 24; idx += 512) {
        const int n = idx / 24, k = idx % 24;
        W1l[idx] = k < K1 ? a.W1[(int64_t)n * a.ldw1 + k] : (k == K1 ? (a.b1 ? a.b1[n] : 0.f) : 0.f);
    }
    if (tid < 128) bl[tid] = a.b2 ? a.b2[tid] : 0.f;
    __syncthreads();
    const float* w1p = W1l + li * 24 + 12 * lk;              // W1'[32 t + li][12 lk + 0..11] at + t * 32 * 24
    const uint4* w2p = W2p + lane;
    const float* blp = bl + 4 * lk;
    for (; tile < ntiles; tile += nw) {
        f32x4 x[3] = {xn[0], xn[1], xn[2]};
        fetch_x(tile + nw, xn);                              // next tile's rows: in flight under this tile's MFMAs
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        // product 1 (fp32 pipe, K = 21 -> 24): 12 steps x 4 column tiles, operands from LDS
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x4 w[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) w[t] = *(const f32x4*)(w1p + t * 32 * 24 + 4 * c);
#define KGW_MLP_STEP(C)                                                                                   \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].C, x[c].C, acc[t], 0, 0, 0);
            KGW_MLP_STEP(x) KGW_MLP_STEP(y) KGW_MLP_STEP(z) KGW_MLP_STEP(w)
#undef KGW_MLP_STEP
        }
        const int64_t row = (int64_t)tile * 32 + li;         // this lane's row
        const bool live = row < rows;
        if (a.Xg && live) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (12 * lk + 4 * c + 4 <= K1) *(f32x4*)(a.Xg + row * a.ldxg + 12 * lk + 4 * c) = x[c];
        }
        f32x4 xa[16];                                        // h1: element c of xa[4 t + g] = column 32 t + 8 g + 4 lk + c
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                v.x = fmaxf(acc[t][4 * g + 0], 0.f); v.y = fmaxf(acc[t][4 * g + 1], 0.f);
                v.z = fmaxf(acc[t][4 * g + 2], 0.f); v.w = fmaxf(acc[t][4 * g + 3], 0.f);
                xa[4 * t + g] = v;
            }
        if (a.H1 && live) {
            float* hp = a.H1 + row * a.ldh1 + 4 * lk;
#pragma unroll
            for (int q = 0; q < 16; ++q) *(f32x4*)(hp + 32 * (q >> 2) + 8 * (q & 3)) = xa[q];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        // product 2 (bf16 pipe, three exact pieces per operand): 8 steps x 6 piece products x 4 output tiles
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            const f32x4 u = xa[2 * s_], v = xa[2 * s_ + 1];
            const float h[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
            uint4 p1, p2, p3;
            kgw_split3x8(h, p1, p2, p3);
            const kgw_bf8 hb[3] = {__builtin_bit_cast(kgw_bf8, p1), __builtin_bit_cast(kgw_bf8, p2), __builtin_bit_cast(kgw_bf8, p3)};
            kgw_bf8 wa[3][4];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) wa[p][ot] = __builtin_bit_cast(kgw_bf8, w2p[((s_ * 3 + p) * 4 + ot) * 64]);
            constexpr int TW[6] = {0, 2, 1, 0, 1, 0}, TH[6] = {2, 0, 1, 1, 0, 0};       // (piece of W2, piece of h1), smallest first
#pragma unroll
            for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                for (int ot = 0; ot < 4; ++ot)
                    acc[ot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[TW[t6]][ot], hb[TH[t6]], acc[ot], 0, 0, 0);
        }
        if (live) {
            float* yp = a.H2 + row * a.ldh2 + 4 * lk;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *(const f32x4*)(blp + t * 32 + 8 * g);
                    f32x4 v;
                    v.x = fmaxf(acc[t][4 * g + 0] + b4.x, 0.f); v.y = fmaxf(acc[t][4 * g + 1] + b4.y, 0.f);
                    v.z = fmaxf(acc[t][4 * g + 2] + b4.z, 0.f); v.w = fmaxf(acc[t][4 * g + 3] + b4.w, 0.f);
                    *(f32x4*)(yp + t * 32 + 8 * g) = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// kgw_mlp2w_fwd: the same two hidden layers for a 128-wide input on FEW rows (the three GO node types of a batch share
// go_feat_mlp, kgwas/model.py:58-60: ~7 k rows), rows gathered from up to four resident feature matrices -- one launch
// instead of gather + Linear + Linear.  A wavefront takes (32-row tile, half of the OUTPUT columns): it computes all of
// h1 for its rows (256 MFMAs, first-layer operands from LDS) and its half of h2 (128 MFMAs) -- the duplicated first
// product buys twice the wavefronts for a launch that has ~220 tiles for 1024 SIMDs.  Hidden state handed over in
// registers as in k_mlp2_fwd; the column-half-0 wavefront also writes the gathered rows and h1 for the backward.
// ------------------------------------------------------------------------------------------------------
struct Mlp2wArgs {
    const float* src[4]; const int32_t* ids[4]; int64_t row0[5];     // job j covers rows [row0[j], row0[j+1])
    int n_jobs; int64_t ldx;
    const float* W1; int64_t ldw1; const float* b1;
    const float* W2; int64_t ldw2; const float* b2;
    float* Xg; float* H1; float* H2; int64_t ldo;                    // [rows, 128] each (row stride ldo)
    int64_t rows;
};

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_mlp2w_fwd(Mlp2wArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* W1l = lds;                                        // [128 n][WST]
    float* W2l = lds + 128 * WST;                            // [128 n][WST]
    float* bl = lds + 2 * 128 * WST;                         // b1 | b2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 2 + (wave >> 1);
    const int half = wave & 1, t0 = half * 2;
    const int64_t row = tile * 32 + li;
    const bool live = row < a.rows;
    const int64_t rc = live ? row : a.rows - 1;
    f32x4 xa[16];
    {
        int j = 0;
        while (j + 1 < a.n_jobs && rc >= a.row0[j + 1]) ++j;
        const float* xp = a.src[j] + (int64_t)a.ids[j][rc - a.row0[j]] * a.ldx + lk * 64;
#pragma unroll
        for (int q = 0; q < 16; ++q) xa[q] = *(const f32x4*)(xp + 4 * q);
    }
    {   // stage both weight matrices: 32 float4 per thread in flight before the LDS writes
        f32x4 wv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * it;
            wv[it] = *(const f32x4*)(a.W1 + (int64_t)(idx >> 5) * a.ldw1 + (idx & 31) * 4);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * it;
            *(f32x4*)(W1l + (idx >> 5) * WST + (idx & 31) * 4) = wv[it];
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * it;
            wv[it] = *(const f32x4*)(a.W2 + (int64_t)(idx >> 5) * a.ldw2 + (idx & 31) * 4);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * it;
            *(f32x4*)(W2l + (idx >> 5) * WST + (idx & 31) * 4) = wv[it];
        }
    }
    if (tid < 128) bl[tid] = a.b1 ? a.b1[tid] : 0.f; else bl[tid] = a.b2 ? a.b2[tid - 128] : 0.f;
    __syncthreads();
    if (half == 0 && live) {                                 // the gathered rows, for the first layer's weight gradient
        float* gp = a.Xg + row * a.ldo + lk * 64;
#pragma unroll
        for (int q = 0; q < 16; ++q) *(f32x4*)(gp + 4 * q) = xa[q];
    }
    // product 1: every column tile; operands W1[32 t + li][64 lk + 4 q + c] from LDS, a step ahead
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const float* w1p = W1l + li * WST + lk * 64;
    f32x4 wn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) wn[t] = *(const f32x4*)(w1p + t * 32 * WST);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        f32x4 w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = wn[t];
        if (q + 1 < 16) {
#pragma unroll
            for (int t = 0; t < 4; ++t) wn[t] = *(const f32x4*)(w1p + t * 32 * WST + 4 * (q + 1));
        }
#define KGW_MLPW_STEP(C)                                                                                  \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                    \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].C, xa[q].C, acc[t], 0, 0, 0);
        KGW_MLPW_STEP(x) KGW_MLPW_STEP(y) KGW_MLPW_STEP(z) KGW_MLPW_STEP(w)
#undef KGW_MLPW_STEP
    }
    // h1 = relu(. + b1): accumulator element 4 g + c of tile t = column 32 t + 8 g + 4 lk + c of this lane's row
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *(const f32x4*)(bl + t * 32 + 8 * g + 4 * lk);
            f32x4 v;
            v.x = fmaxf(acc[t][4 * g + 0] + b4.x, 0.f); v.y = fmaxf(acc[t][4 * g + 1] + b4.y, 0.f);
            v.z = fmaxf(acc[t][4 * g + 2] + b4.z, 0.f); v.w = fmaxf(acc[t][4 * g + 3] + b4.w, 0.f);
            xa[4 * t + g] = v;
        }
    if (half == 0 && live) {
        float* hp = a.H1 + row * a.ldo + 4 * lk;
#pragma unroll
        for (int q = 0; q < 16; ++q) *(f32x4*)(hp + 32 * (q >> 2) + 8 * (q & 3)) = xa[q];
    }
    // product 2: this wavefront's two column tiles; MFMA step (q, c) multiplies k = 32 (q >> 2) + 8 (q & 3) + 4 lk + c
    f32x16 ac2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) ac2[t][e] = 0.f;
    const float* w2p = W2l + (t0 * 32 + li) * WST + 4 * lk;
    f32x4 vn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) vn[t] = *(const f32x4*)(w2p + t * 32 * WST);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        f32x4 w[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) w[t] = vn[t];
        if (q + 1 < 16) {
#pragma unroll
            for (int t = 0; t < 2; ++t) vn[t] = *(const f32x4*)(w2p + t * 32 * WST + 32 * ((q + 1) >> 2) + 8 * ((q + 1) & 3));
        }
#define KGW_MLPW_STEP(C)                                                                                  \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                    \
            ac2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].C, xa[q].C, ac2[t], 0, 0, 0);
        KGW_MLPW_STEP(x) KGW_MLPW_STEP(y) KGW_MLPW_STEP(z) KGW_MLPW_STEP(w)
#undef KGW_MLPW_STEP
    }
    if (live) {
        float* yp = a.H2 + row * a.ldo + 4 * lk;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *(const f32x4*)(bl + 128 + (t0 + t) * 32 + 8 * g + 4 * lk);
                f32x4 v;
                v.x = fmaxf(ac2[t][4 * g + 0] + b4.x, 0.f); v.y = fmaxf(ac2[t][4 * g + 1] + b4.y, 0.f);
                v.z = fmaxf(ac2[t][4 * g + 2] + b4.z, 0.f); v.w = fmaxf(ac2[t][4 * g + 3] + b4.w, 0.f);
                *(f32x4*)(yp + (t0 + t) * 32 + 8 * g) = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------------
// kgw_mlp2_bwd_first: the backward of a NARROW first layer behind kgw_mlp2_fwd (the 20-wide SNP features need no input
// gradient): d W1 = dh1^T x, d b1 = colsum(dh1) with dh1 = (dh2 W2) * (h1 > 0) -- WITHOUT materialising dh1.  It is
// k_linear_wreg<w_kn, mask> (dh2 tiles streamed into the MFMA B operand, W2 stationary) whose epilogue, instead of
// storing the 32 x 128 tile of dh1, hands it through wavefront-private LDS to a second product: the tile, read back
// "column per lane", is the B operand of  C[k][col] += x'[row][k] dh1[row][col]  (x' = [x | 1]: row K1 of C is d b1), 64
// more MFMAs per tile into four persistent accumulators.  The blocks' partial C's are added by k_mlp2_bwd_fold.
// Replaces a 61 MB store, its re-read and the [rows, 128]^T [rows, 20] product (k_tn_gemm<2,1> + reduce: 35 us).
// ------------------------------------------------------------------------------------------------------
struct Mlp2BwdArgs {
    const float* dH2; int64_t ldd;      // [rows, 128] upstream gradient (already multiplied by h2 > 0)
    const float* W2; int64_t ldw;       // [128 out, 128 in] (nn.Linear layout): dh1 = dh2 @ W2
    const float* H1; int64_t ldm;       // [rows, 128] ReLU mask
    const float* X; int64_t ldx;        // [rows, K1] the first layer's input rows
    float* part;                        // [gridDim.x][4096] block partials, fragment order
    int64_t rows; int K1;
    const int32_t* rows_dev;
    const int32_t* in_ids;              // nullable: row r of the product reads dH2[in_ids[r]]; in_ids[r] < 0 => dh1 row r is zero
    float* dZ; int64_t ldz;             // nullable: the masked dh1 rows are ALSO written here (a wide first layer's own
                                        // weight gradient is a library product over them); K1 = 0 then leaves just d b1
    uint4* packed; int flip;            // nullable (k_mlp2_bwd_first3 only): the masked dh1 rows ALSO as kgw_gemm3's B operand image
                                        // ([rows rounded up to 32][128] in three bf16 pieces, kgw_gemm3_pack's s_is_kn form, sign
                                        // periods of `flip` chunks) -- the wavefront packs the tile it has in LDS anyway
};

constexpr int TST = 132;                // LDS row stride of the transposing tile

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_mlp2_bwd_first(Mlp2BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Wl = lds;                                         // [128 k][WST]: W2^T as the MFMA A operand wants it
    float* Tl = lds + 128 * WST;                             // [4 wavefronts][32 rows][TST]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    int64_t rows = a.rows;
    if (a.rows_dev) { const int64_t r = *a.rows_dev; rows = r < 0 ? 0 : (r < a.rows ? r : a.rows); }
    const int ntiles = (int)((rows + 31) / 32);
    const int nw = (int)gridDim.x * 4;
    float* Tw = Tl + wave * 32 * TST;
    {   // stage W2 transposed (dX form, w_kn): Wl[k][n] = W2[n][k]; lanes along n: conflict-free transposing writes
        f32x4 wv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * it;
            wv[it] = *(const f32x4*)(a.W2 + (int64_t)(idx & 127) * a.ldw + (idx >> 7) * 4);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * it;
            const int k = idx & 127, n4 = (idx >> 7) * 4;
            Wl[(n4 + 0) * WST + k] = wv[it].x; Wl[(n4 + 1) * WST + k] = wv[it].y;
            Wl[(n4 + 2) * WST + k] = wv[it].z; Wl[(n4 + 3) * WST + k] = wv[it].w;
        }
    }
    __syncthreads();
    // output columns (= input features of W2) 0-63 of the operand in registers, 64-127 from LDS a step ahead
    f32x4 bw[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) bw[t][q] = *(const f32x4*)(Wl + (t * 32 + li) * WST + lk * 64 + 4 * q);
    const float* w2 = Wl + (64 + li) * WST + lk * 64;
    const float* w3 = Wl + (96 + li) * WST + lk * 64;
    f32x16 accw[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[t][e] = 0.f;
    const int K1 = a.K1;
    int tile = (int)blockIdx.x * 4 + wave;
    f32x4 xa[16];
    int src_cur;                                              // input row of this lane's row (-1: none -> zero row)
    {
        int64_t r = (int64_t)(tile < ntiles ? tile : ntiles - 1) * 32 + li;
        if (r >= rows) r = rows - 1;
        src_cur = a.in_ids ? a.in_ids[r] : 0;
        const float* xp = a.dH2 + (a.in_ids ? (int64_t)(src_cur < 0 ? 0 : src_cur) : r) * a.ldd + lk * 64;
#pragma unroll
        for (int q = 0; q < 16; ++q) xa[q] = *(const f32x4*)(xp + 4 * q);
    }
    for (; tile < ntiles; tile += nw) {
        const int64_t r0 = (int64_t)tile * 32;
        const int64_t row = r0 + li;
        const bool live = row < rows && src_cur >= 0;
        const int64_t rc = row < rows ? row : rows - 1;
        const float* xn;                                      // this lane's half row of the wavefront's NEXT tile
        {
            int nt = tile + nw;
            if (nt >= ntiles) nt = ntiles - 1;               // last round: a harmless re-read
            int64_t r = (int64_t)nt * 32 + li;
            if (r >= rows) r = rows - 1;
            src_cur = a.in_ids ? a.in_ids[r] : 0;            // (of the NEXT tile from here on: `live` above is this tile's)
            xn = a.dH2 + (a.in_ids ? (int64_t)(src_cur < 0 ? 0 : src_cur) : r) * a.ldd + lk * 64;
        }
        // x' in "k per lane" form for the second product: lane (k = li, row parity lk), step s = row pair (needed after
        // the first product: the loads ride under its MFMAs)
        float xs[16];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const int64_t rr = r0 + 2 * s2 + lk;
            float v = 0.f;
            if (rr < rows) v = li < K1 ? a.X[rr * a.ldx + li] : (li == K1 ? 1.f : 0.f);
            xs[s2] = v;
        }
        // ReLU mask of this lane's row (columns 32 t + 8 g + 4 lk + c), fetched under the MFMAs, kept as bits
        const float* mp = a.H1 + rc * a.ldm + 4 * lk;
        unsigned mb[2] = {0u, 0u};
        f32x4 mv[4];
#define KGW_MLPB_MFETCH(T)                                                                                \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) mv[g] = *(const f32x4*)(mp + (T) * 32 + 8 * g);
#define KGW_MLPB_MBITS(T)                                                                                 \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                  \
            const int b0 = ((T) & 1) * 16 + 4 * g;                                                        \
            mb[(T) >> 1] |= (mv[g].x > 0.f ? 1u : 0u) << (b0 + 0);                                        \
            mb[(T) >> 1] |= (mv[g].y > 0.f ? 1u : 0u) << (b0 + 1);                                        \
            mb[(T) >> 1] |= (mv[g].z > 0.f ? 1u : 0u) << (b0 + 2);                                        \
            mb[(T) >> 1] |= (mv[g].w > 0.f ? 1u : 0u) << (b0 + 3);                                        \
        }
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        f32x4 b2n = *(const f32x4*)w2, b3n = *(const f32x4*)w3;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q == 0) { KGW_MLPB_MFETCH(0) }
            if (q == 3) { KGW_MLPB_MBITS(0) KGW_MLPB_MFETCH(1) }
            if (q == 6) { KGW_MLPB_MBITS(1) KGW_MLPB_MFETCH(2) }
            if (q == 10) { KGW_MLPB_MBITS(2) KGW_MLPB_MFETCH(3) }
            if (q == 14) { KGW_MLPB_MBITS(3) }
            const f32x4 b2 = b2n, b3 = b3n;
            if (q + 1 < 16) { b2n = *(const f32x4*)(w2 + 4 * (q + 1)); b3n = *(const f32x4*)(w3 + 4 * (q + 1)); }
#define KGW_MLPB_STEP(C)                                                                                  \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[0][q].C, xa[q].C, acc[0], 0, 0, 0);          \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[1][q].C, xa[q].C, acc[1], 0, 0, 0);          \
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2.C, xa[q].C, acc[2], 0, 0, 0);                \
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b3.C, xa[q].C, acc[3], 0, 0, 0);
            KGW_MLPB_STEP(x) KGW_MLPB_STEP(y) KGW_MLPB_STEP(z) KGW_MLPB_STEP(w)
#undef KGW_MLPB_STEP
            if (q == 7) {                                      // first half row of the next tile, in place
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int qq = 0; qq < 8; ++qq) xa[qq] = *(const f32x4*)(xn + 4 * qq);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef KGW_MLPB_MFETCH
#undef KGW_MLPB_MBITS
        if (!live) { mb[0] = 0u; mb[1] = 0u; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qq = 8; qq < 16; ++qq) xa[qq] = *(const f32x4*)(xn + 4 * qq);     // second half: under the second product
        __builtin_amdgcn_sched_barrier(0);
        // masked dh1 tile -> the wavefront's LDS tile, row per lane (nobody else reads it: no barrier)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned m4 = mb[t >> 1] >> (((t & 1) << 4) + 4 * g);
                f32x4 v;
                v.x = (m4 & 1u) ? acc[t][4 * g + 0] : 0.f; v.y = (m4 & 2u) ? acc[t][4 * g + 1] : 0.f;
                v.z = (m4 & 4u) ? acc[t][4 * g + 2] : 0.f; v.w = (m4 & 8u) ? acc[t][4 * g + 3] : 0.f;
                *(f32x4*)(Tw + li * TST + t * 32 + 8 * g + 4 * lk) = v;
                if (a.dZ && row < rows) *(f32x4*)(a.dZ + row * a.ldz + t * 32 + 8 * g + 4 * lk) = v;
            }
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // second product: C[k][col] += x'[row][k] dh1[row][col], two rows per MFMA step
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const float* tp = Tw + (2 * s2 + lk) * TST + li;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                accw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[s2], tp[t * 32], accw[t], 0, 0, 0);
        }
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the tile is rewritten by the next iteration)
    }
    // the block's four partial C's through LDS (fragment order), added in wavefront order
    __syncthreads();
    float* R = Tl;                                            // 4 x 4096 floats
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) R[wave * 4096 + (t * 16 + e) * 64 + lane] = accw[t][e];
    __syncthreads();
    for (int f = tid; f < 4096; f += 256)
        a.part[(int64_t)blockIdx.x * 4096 + f] = (R[f] + R[4096 + f]) + (R[2 * 4096 + f] + R[3 * 4096 + f]);
}

// k_mlp2_bwd_first3: the same kernel with its FIRST product (dh1 = dH2 W2, 128 x 128, 80 % of the MFMA cycles) on the bf16
// matrix pipe, three exact bf16 pieces per operand as in kgw_gemm3.hip / k_mlp2_fwd3: W2^T is split once per block into an LDS
// image of MFMA operands (96 KB), the dH2 half row a lane holds is split in its registers, eight values per step; 6 144 instead
// of 16 384 MFMA cycles per 32-row tile.  The masked tile goes through LDS 64 columns at a time (35 KB for the four wavefronts).
constexpr int TS2 = 68;                 // LDS row stride of the half-width transposing tile

// (PACK: also write the tile as kgw_gemm3's operand image -- a template so that the variant without it keeps its schedule)
template <bool PACK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_mlp2_bwd_first3(Mlp2BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4* W2p = (uint4*)lds;                                // W2^T operand image: [8 steps][3 pieces][4 column tiles][64 lanes]
    float* Tl = lds + M3_W2_U4 * 4;                          // [4 wavefronts][32 rows][TS2]: half of the columns at a time
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const unsigned sgn = (li & 1) ? 0x80000000u : 0u;
    int64_t rows = a.rows;
    if (a.rows_dev) { const int64_t r = *a.rows_dev; rows = r < 0 ? 0 : (r < a.rows ? r : a.rows); }
    const int ntiles = (int)((rows + 31) / 32);
    const int nw = (int)gridDim.x * 4;
    float* Tw = Tl + wave * 32 * TS2;
    // the operand image: entry (s, p, jt, lane) = piece p of W2[o = 64 lk + 8 s + i][32 jt + li], i = 0..7 -- the eight values of
    // dH2 lane group lk multiplies in step s (its half row, in order)
    for (int idx = tid; idx < 8 * 4 * 64; idx += 256) {
        const int ln = idx & 63, jt = (idx >> 6) & 3, s_ = idx >> 8;
        const float* wp = a.W2 + (int64_t)(64 * (ln >> 5) + 8 * s_) * a.ldw + 32 * jt + (ln & 31);
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = wp[(int64_t)i * a.ldw];
        uint4 p1, p2, p3;
        kgw_split3x8(x, p1, p2, p3);
        uint4* o = W2p + ((s_ * 3) * 4 + jt) * 64 + ln;
        o[0] = p1; o[4 * 64] = p2; o[8 * 64] = p3;
    }
    __syncthreads();
    const uint4* w2p = W2p + lane;
    f32x16 accw[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[t][e] = 0.f;
    const int K1 = a.K1;
    int tile = (int)blockIdx.x * 4 + wave;
    f32x4 xa[16];
    int src_cur;                                              // input row of this lane's row (-1: none -> zero row)
    {
        int64_t r = (int64_t)(tile < ntiles ? tile : ntiles - 1) * 32 + li;
        if (r >= rows) r = rows - 1;
        src_cur = a.in_ids ? a.in_ids[r] : 0;
        const float* xp = a.dH2 + (a.in_ids ? (int64_t)(src_cur < 0 ? 0 : src_cur) : r) * a.ldd + lk * 64;
#pragma unroll
        for (int q = 0; q < 16; ++q) xa[q] = *(const f32x4*)(xp + 4 * q);
    }
    // ReLU mask of this lane's row (columns 32 t + 8 g + 4 lk + c) as bits.  Round 5: the NEXT tile's mask rows are requested at the
    // start of a tile's first product and turned into bits after its second (with one wavefront per SIMD nothing else hides the
    // latency: fetched in four groups inside the product that consumes them, 50 % of the kernel's cycles were s_waitcnt).
#define KGW_MLPB_BITS(MV, MB)                                                                            \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) _Pragma("unroll") for (int g = 0; g < 4; ++g) {    \
        const int b0 = (t_ & 1) * 16 + 4 * g;                                                            \
        MB[t_ >> 1] |= (MV[t_][g].x > 0.f ? 1u : 0u) << (b0 + 0);                                        \
        MB[t_ >> 1] |= (MV[t_][g].y > 0.f ? 1u : 0u) << (b0 + 1);                                        \
        MB[t_ >> 1] |= (MV[t_][g].z > 0.f ? 1u : 0u) << (b0 + 2);                                        \
        MB[t_ >> 1] |= (MV[t_][g].w > 0.f ? 1u : 0u) << (b0 + 3);                                        \
    }
    unsigned mb[2] = {0u, 0u};
    {
        int64_t r = (int64_t)(tile < ntiles ? tile : ntiles - 1) * 32 + li;
        if (r >= rows) r = rows - 1;
        const float* mp = a.H1 + r * a.ldm + 4 * lk;
        f32x4 mv[4][4];
#pragma unroll
        for (int t_ = 0; t_ < 4; ++t_)
#pragma unroll
            for (int g = 0; g < 4; ++g) mv[t_][g] = *(const f32x4*)(mp + t_ * 32 + 8 * g);
        KGW_MLPB_BITS(mv, mb)
    }
    for (; tile < ntiles; tile += nw) {
        const int64_t r0 = (int64_t)tile * 32;
        const int64_t row = r0 + li;
        const bool live = row < rows && src_cur >= 0;
        const int64_t rc = row < rows ? row : rows - 1;
        const float* xn;                                      // this lane's half row of the wavefront's NEXT tile
        {
            int nt = tile + nw;
            if (nt >= ntiles) nt = ntiles - 1;               // last round: a harmless re-read
            int64_t r = (int64_t)nt * 32 + li;
            if (r >= rows) r = rows - 1;
            src_cur = a.in_ids ? a.in_ids[r] : 0;            // (of the NEXT tile from here on: `live` above is this tile's)
            xn = a.dH2 + (a.in_ids ? (int64_t)(src_cur < 0 ? 0 : src_cur) : r) * a.ldd + lk * 64;
        }
        // x' in "k per lane" form for the second product: lane (k = li, row parity lk), step s = row pair (needed after
        // the first product: the loads ride under its MFMAs)
        float xs[16];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const int64_t rr = r0 + 2 * s2 + lk;
            float v = 0.f;
            if (rr < rows) v = li < K1 ? a.X[rr * a.ldx + li] : (li == K1 ? 1.f : 0.f);
            xs[s2] = v;
        }
        // the NEXT tile's mask rows: requested now, read after this tile's second product
        f32x4 mvn[4][4];
        {
            int nt = tile + nw;
            if (nt >= ntiles) nt = ntiles - 1;
            int64_t r = (int64_t)nt * 32 + li;
            if (r >= rows) r = rows - 1;
            const float* mpn = a.H1 + r * a.ldm + 4 * lk;
#pragma unroll
            for (int t_ = 0; t_ < 4; ++t_)
#pragma unroll
                for (int g = 0; g < 4; ++g) mvn[t_][g] = *(const f32x4*)(mpn + t_ * 32 + 8 * g);
        }
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        // first product on the bf16 pipe (three exact pieces per operand): 8 steps x 6 piece products x 4 column tiles
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            const f32x4 u = xa[2 * s_], v = xa[2 * s_ + 1];
            // odd rows are multiplied NEGATED (exact) and their result negated back: the bf16 MFMA's internal addition truncates
            // (a small negative mean error), and dW1 / db1 sum dh1 over the rows -- with alternating signs the means cancel
            const float h[8] = {kgw_fxor(u.x, sgn), kgw_fxor(u.y, sgn), kgw_fxor(u.z, sgn), kgw_fxor(u.w, sgn),
                                kgw_fxor(v.x, sgn), kgw_fxor(v.y, sgn), kgw_fxor(v.z, sgn), kgw_fxor(v.w, sgn)};
            uint4 p1, p2, p3;
            kgw_split3x8(h, p1, p2, p3);
            const kgw_bf8 hb[3] = {__builtin_bit_cast(kgw_bf8, p1), __builtin_bit_cast(kgw_bf8, p2), __builtin_bit_cast(kgw_bf8, p3)};
            kgw_bf8 wa[3][4];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) wa[p][jt] = __builtin_bit_cast(kgw_bf8, w2p[((s_ * 3 + p) * 4 + jt) * 64]);
            constexpr int TW[6] = {0, 2, 1, 0, 1, 0}, TH[6] = {2, 0, 1, 1, 0, 0};       // (piece of W2, piece of dH2), smallest first
#pragma unroll
            for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
                    acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[TW[t6]][jt], hb[TH[t6]], acc[jt], 0, 0, 0);
            if (s_ == 3) {                                     // first half row of the next tile, in place
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int qq = 0; qq < 8; ++qq) xa[qq] = *(const f32x4*)(xn + 4 * qq);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!live) { mb[0] = 0u; mb[1] = 0u; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qq = 8; qq < 16; ++qq) xa[qq] = *(const f32x4*)(xn + 4 * qq);     // second half: under the second product
        __builtin_amdgcn_sched_barrier(0);
        // masked dh1 tile -> the wavefront's LDS tile, row per lane (nobody else reads it: no barrier), 64 columns at a time;
        // second product: C[k][col] += x'[row][k] dh1[row][col], two rows per MFMA step
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int t = 2 * hh; t < 2 * hh + 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned m4 = mb[t >> 1] >> (((t & 1) << 4) + 4 * g);
                    f32x4 v;
                    v.x = (m4 & 1u) ? kgw_fxor(acc[t][4 * g + 0], sgn) : 0.f; v.y = (m4 & 2u) ? kgw_fxor(acc[t][4 * g + 1], sgn) : 0.f;
                    v.z = (m4 & 4u) ? kgw_fxor(acc[t][4 * g + 2], sgn) : 0.f; v.w = (m4 & 8u) ? kgw_fxor(acc[t][4 * g + 3], sgn) : 0.f;
                    *(f32x4*)(Tw + li * TS2 + (t - 2 * hh) * 32 + 8 * g + 4 * lk) = v;
                    if (a.dZ && row < rows) *(f32x4*)(a.dZ + row * a.ldz + t * 32 + 8 * g + 4 * lk) = v;
                }
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (PACK) {
                // k_g3_pack<true>'s work for this half tile: chunk c = tile, item (j, nt, lane) = the eight rows
                // k = 16 j + 8 (lane >> 5) + i of column 32 nt + (lane & 31); same values, same three pieces, same image index
                const bool neg = a.flip && ((tile / a.flip) & 1);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int ntl = it & 1, j = it >> 1;
                    const float* tp = Tw + (16 * j + 8 * lk) * TS2 + 32 * ntl + li;
                    float x[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = neg ? -tp[i * TS2] : tp[i * TS2];
                    uint4 p1, p2, p3;
                    kgw_split3x8(x, p1, p2, p3);
                    uint4* o = a.packed + (((int64_t)tile * 2 + j) * 3 * 4 + (2 * hh + ntl)) * 64 + lane;
                    o[0] = p1; o[4 * 64] = p2; o[8 * 64] = p3;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const float* tp = Tw + (2 * s2 + lk) * TS2 + li;
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    accw[2 * hh + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[s2], tp[t * 32], accw[2 * hh + t], 0, 0, 0);
            }
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the tile is rewritten next)
        }
        mb[0] = 0u; mb[1] = 0u;
        KGW_MLPB_BITS(mvn, mb)
    }
#undef KGW_MLPB_BITS
    // the block's four partial C's through LDS (fragment order), added in wavefront order
    __syncthreads();
    float* R = lds;                                           // 4 x 4096 floats over the operand image (done with)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) R[wave * 4096 + (t * 16 + e) * 64 + lane] = accw[t][e];
    __syncthreads();
    for (int f = tid; f < 4096; f += 256)
        a.part[(int64_t)blockIdx.x * 4096 + f] = (R[f] + R[4096 + f]) + (R[2 * 4096 + f] + R[3 * 4096 + f]);
}

// d W1 [128, K1] and d b1 [128] from the block partials: fragment f = (t * 16 + e) * 64 + lane holds
// C[k = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)][col = 32 t + (lane & 31)]; fixed summation order
__global__ void __launch_bounds__(1024) k_mlp2_bwd_fold(const float* __restrict__ part, int nblk, int K1, float* __restrict__ dW1,
                                                        int64_t ldw, float* __restrict__ db1) {
    __shared__ float sm[1024];
    const int fl = threadIdx.x & 63, g = threadIdx.x >> 6;        // 64 fragment elements x 16 groups of blocks
    const int f = blockIdx.x * 64 + fl;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    int q = 0;
    for (int b = g; b < nblk; b += 16, ++q) s4[q & 3] += part[(int64_t)b * 4096 + f];
    sm[threadIdx.x] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    __syncthreads();
    if (g == 0) {
        float sv = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k += 4) sv += (sm[k * 64 + fl] + sm[(k + 1) * 64 + fl]) + (sm[(k + 2) * 64 + fl] + sm[(k + 3) * 64 + fl]);
        const int lane = f & 63, e = (f >> 6) & 15, t = f >> 10;
        const int k = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col = 32 * t + (lane & 31);
        if (k < K1) dW1[(int64_t)col * ldw + k] = sv;
        else if (k == K1) db1[col] = sv;
    }
}

// Same product for FEW row tiles (up to 512: the GO / gene matrices of a batch): with one 32-row tile per wavefront
// only ntiles of the chip's 1024 SIMDs get work.  Here a wavefront takes one tile x ONE HALF of the output columns
// (128 MFMAs, 128 registers of W), two wavefronts per SIMD, every task resident at once: no tile loop, no refill.
template <bool WKN, bool MASK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_linear_wreg_half(LinArgs a_) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    LinArgs a = a_;
    {
        const int64_t re = lin_rows_eff(a_);
        lin_zero_padding(a_, re, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
        a.rows = re;
        if ((int64_t)blockIdx.x * 2 * 32 >= re) return;      // (before any barrier: the whole block leaves)
    }
    float* Wl = lds;                                         // [128 n][WST], k contiguous; bias behind it
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lk = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 2 + (wave >> 1);
    const int t0 = (wave & 1) * 2;                            // this wavefront's two 32-column blocks
    const int64_t row = tile * 32 + li;
    const bool live = row < a.rows;
    const int64_t rc = live ? row : a.rows - 1;
    f32x4 xa[16];
    {
        const float* xp = a.X + rc * a.ldx + lk * 64;
#pragma unroll
        for (int q = 0; q < 16; ++q) xa[q] = *(const f32x4*)(xp + 4 * q);
    }
    f32x4 mv[2][4];
    if (MASK) {
        const float* mp = a.mask + rc * a.ldm + 4 * lk;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) mv[t][g] = *(const f32x4*)(mp + (t0 + t) * 32 + 8 * g);
    }
    {   // stage W: all 16 loads of a thread in flight before the first LDS write
        f32x4 wv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * it;
            if (!WKN) wv[it] = *(const f32x4*)(a.W + (int64_t)(idx >> 5) * a.ldw + (idx & 31) * 4);
            else wv[it] = *(const f32x4*)(a.W + (int64_t)(idx & 127) * a.ldw + (idx >> 7) * 4);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = tid + 256 * it;
            if (!WKN) {
                *(f32x4*)(Wl + (idx >> 5) * WST + (idx & 31) * 4) = wv[it];
            } else {
                const int k = idx & 127, n4 = (idx >> 7) * 4;
                Wl[(n4 + 0) * WST + k] = wv[it].x; Wl[(n4 + 1) * WST + k] = wv[it].y;
                Wl[(n4 + 2) * WST + k] = wv[it].z; Wl[(n4 + 3) * WST + k] = wv[it].w;
            }
        }
    }
    if (tid < 128) Wl[128 * WST + tid] = a.bias ? a.bias[tid] : 0.f;
    __syncthreads();
    f32x4 bw[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) bw[t][q] = *(const f32x4*)(Wl + ((t0 + t) * 32 + li) * WST + lk * 64 + 4 * q);
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[t][q].x, xa[q].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[t][q].y, xa[q].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[t][q].z, xa[q].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[t][q].w, xa[q].w, acc[t], 0, 0, 0);
    }
    if (live) {
        const float* bl = Wl + 128 * WST + 4 * lk;
        float* yp = a.Y + row * a.ldy + 4 * lk;
        const float lo = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *(const f32x4*)(bl + (t0 + t) * 32 + 8 * g);
                f32x4 v;
                v.x = fmaxf(acc[t][4 * g + 0] + b4.x, lo); v.y = fmaxf(acc[t][4 * g + 1] + b4.y, lo);
                v.z = fmaxf(acc[t][4 * g + 2] + b4.z, lo); v.w = fmaxf(acc[t][4 * g + 3] + b4.w, lo);
                if (MASK) {
                    v.x = mv[t][g].x > 0.f ? v.x : 0.f; v.y = mv[t][g].y > 0.f ? v.y : 0.f;
                    v.z = mv[t][g].z > 0.f ? v.z : 0.f; v.w = mv[t][g].w > 0.f ? v.w : 0.f;
                }
                *(f32x4*)(yp + (t0 + t) * 32 + 8 * g) = v;
            }
        }
    }
}

template <bool WKN>
int launch_wreg(const LinArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(128 * WST + 128) * sizeof(float);
    static KgwPerDevice attr_once;
    if (attr_once.need()) {
        KGW_HIP(hipFuncSetAttribute((const void*)k_linear_wreg<WKN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        KGW_HIP(hipFuncSetAttribute((const void*)k_linear_wreg<WKN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int64_t ntiles = (a.rows + 31) / 32;
    static const int64_t half_max = getenv("KGW_WREG_HALF_MAX_TILES") ? atoll(getenv("KGW_WREG_HALF_MAX_TILES")) : 512;
    // (measured in the step: 1.648 ms with the half-tile kernel up to 512 tiles, 1.653 up to 1024, 1.671 without it)
    if (ntiles <= half_max) {                                 // few tiles: one (tile, column half) per wavefront, all resident
        static KgwPerDevice attr_half;
        if (attr_half.need()) {
            KGW_HIP(hipFuncSetAttribute((const void*)k_linear_wreg_half<WKN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            KGW_HIP(hipFuncSetAttribute((const void*)k_linear_wreg_half<WKN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        const int grid = (int)((ntiles + 1) / 2);
        if (a.mask) k_linear_wreg_half<WKN, true><<<grid, 256, lds, st>>>(a);
        else k_linear_wreg_half<WKN, false><<<grid, 256, lds, st>>>(a);
        KGW_LAUNCH_CHECK();
        return KGW_OK;
    }
    const int64_t nblk = (ntiles + 3) / 4;
    const int grid = (int)(nblk < 256 ? nblk : 256);
    if (a.mask) k_linear_wreg<WKN, true><<<grid, 256, lds, st>>>(a);
    else k_linear_wreg<WKN, false><<<grid, 256, lds, st>>>(a);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

template <int KC, bool WKN, int RT>
int launch_wres(const LinArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(128 * WST + RT * (KC + 4)) * sizeof(float);
    auto kern = k_linear_wres<KC, WKN, RT>;
    static KgwPerDevice attr_once;
    if (attr_once.need()) {
        KGW_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    int64_t ntiles = (a.rows + RT - 1) / RT;
    int grid = (int)(ntiles < 256 ? ntiles : 256);
    kern<<<grid, 512, lds, st>>>(a);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

}  // namespace

extern "C" int kgw_linear(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias,
                          const float* mask, int64_t ldm, float* Y, int64_t ldy, int64_t rows, int32_t K,
                          int32_t N, int32_t relu, int32_t w_is_kn, const int32_t* rows_dev, kgw_stream_t stream_) {
    if (rows == 0) return KGW_OK;
    if (!X || !W || !Y) return KGW_E_NULL;
    if (rows < 0 || K <= 0 || N <= 0) return KGW_E_RANGE;
    // float4 tiles: leading dimensions and K (N for the [K,N] form) must be multiples of 4, bases 16-B aligned
    if ((K & 3) || (ldx & 3) || (ldw & 3) || !aligned16(X) || !aligned16(W)) return KGW_E_UNSUPPORTED;
    if (w_is_kn && (N & 3)) return KGW_E_UNSUPPORTED;
    LinArgs a{X, ldx, W, ldw, bias, mask, ldm, Y, ldy, rows, K, N, relu, w_is_kn, rows_dev};
    static const int64_t wres_min = getenv("KGW_WRES_MIN_ROWS") ? atoll(getenv("KGW_WRES_MIN_ROWS")) : 4096;
    static const int64_t wres_tall = getenv("KGW_WRES_TALL_ROWS") ? atoll(getenv("KGW_WRES_TALL_ROWS")) : 32768;
    if (K <= 128 && N <= 128 && rows >= wres_min && (N & 3) == 0 && (ldy & 3) == 0 && aligned16(Y) && aligned16(bias) &&
        (!mask || ((ldm & 3) == 0 && aligned16(mask)))) {           // weight-resident persistent kernel (tall inputs)
        hipStream_t st = (hipStream_t)stream_;
        // measured (MI355X) against the LDS-staged kernel below: 15-20 % faster under 32 k rows, 5-10 % faster with a ReLU
        // mask, equal otherwise
        static const int64_t wreg_min = getenv("KGW_WREG_MIN_ROWS") ? atoll(getenv("KGW_WREG_MIN_ROWS")) : 4096;
        if (K == 128 && N == 128 && rows >= wreg_min)
            return w_is_kn ? launch_wreg<true>(a, st) : launch_wreg<false>(a, st);
        if (rows >= wres_tall) {
            if (w_is_kn) return K <= 32 ? launch_wres<32, true, 256>(a, st) : launch_wres<64, true, 256>(a, st);
            return K <= 32 ? launch_wres<32, false, 256>(a, st) : launch_wres<64, false, 256>(a, st);
        }
        if (w_is_kn) return K <= 32 ? launch_wres<32, true, 64>(a, st) : launch_wres<64, true, 64>(a, st);
        return K <= 32 ? launch_wres<32, false, 64>(a, st) : launch_wres<64, false, 64>(a, st);
    }
    dim3 grid((unsigned)((rows + LBM - 1) / LBM), (unsigned)((N + LBN - 1) / LBN));
    k_linear<<<grid, 256, 0, (hipStream_t)stream_>>>(a);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_mlp2_fwd(const float* X, int64_t ldx, int32_t K1, const float* W1, int64_t ldw1, const float* b1,
                            const float* W2, int64_t ldw2, const float* b2, float* H1, int64_t ldh1, float* H2, int64_t ldh2,
                            int64_t rows, const int32_t* rows_dev, const int32_t* ids, float* Xg, int64_t ldxg,
                            kgw_stream_t stream_) {
    if (rows == 0) return KGW_OK;
    if (!X || !W1 || !W2 || !H2) return KGW_E_NULL;
    if (Xg && ((ldxg & 3) || !aligned16(Xg))) return KGW_E_UNSUPPORTED;
    if (rows < 0 || K1 <= 0) return KGW_E_RANGE;
    if (K1 > 20 || (K1 & 3) || (ldx & 3) || (ldw2 & 3) || (ldh2 & 3) || (H1 && (ldh1 & 3)) || !aligned16(X) || !aligned16(W2) ||
        !aligned16(H2) || (H1 && !aligned16(H1)))
        return KGW_E_UNSUPPORTED;
    Mlp2Args a{X, ldx, W1, ldw1, b1, W2, ldw2, b2, H1, ldh1, H2, ldh2, rows, K1, rows_dev, ids, Xg, ldxg};
    const size_t lds = (size_t)(128 * WST + 128 + 128 * 24) * sizeof(float);
    static KgwPerDevice attr_once;
    if (attr_once.need()) {
        KGW_HIP(hipFuncSetAttribute((const void*)k_mlp2_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    static const bool split3 = !(getenv("KGW_MLP2_SPLIT") && getenv("KGW_MLP2_SPLIT")[0] == '0');
    if (split3) {           // second product on the bf16 pipe (three exact pieces per operand)
        const size_t lds3 = (size_t)M3_W2_U4 * 16 + (size_t)(128 * 24 + 128) * sizeof(float);
        static KgwPerDevice attr3_set;
        if (attr3_set.need()) {
            KGW_HIP(hipFuncSetAttribute((const void*)k_mlp2_fwd3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        }
        const int64_t nblk3 = ((rows + 31) / 32 + 7) / 8;
        k_mlp2_fwd3<<<(int)(nblk3 < 256 ? nblk3 : 256), 512, lds3, (hipStream_t)stream_>>>(a);
        KGW_LAUNCH_CHECK();
        return KGW_OK;
    }
    const int64_t nblk = ((rows + 31) / 32 + 3) / 4;
    k_mlp2_fwd<<<(int)(nblk < 256 ? nblk : 256), 256, lds, (hipStream_t)stream_>>>(a);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_mlp2w_fwd(int32_t n_jobs, const float* const* src, const int32_t* const* ids, const int64_t* n_rows,
                             int64_t ldx, const float* W1, int64_t ldw1, const float* b1, const float* W2, int64_t ldw2,
                             const float* b2, float* Xg, float* H1, float* H2, int64_t ldo, kgw_stream_t stream_) {
    if (n_jobs <= 0) return KGW_OK;
    if (n_jobs > 4) return KGW_E_RANGE;
    if (!src || !ids || !n_rows || !W1 || !W2 || !Xg || !H1 || !H2) return KGW_E_NULL;
    if ((ldx & 3) || (ldw1 & 3) || (ldw2 & 3) || (ldo & 3) || !aligned16(W1) || !aligned16(W2) || !aligned16(Xg) || !aligned16(H1) ||
        !aligned16(H2))
        return KGW_E_UNSUPPORTED;
    Mlp2wArgs a{};
    a.n_jobs = n_jobs; a.ldx = ldx;
    a.row0[0] = 0;
    for (int j = 0; j < n_jobs; ++j) {
        if (!src[j] || !ids[j] || n_rows[j] < 0) return KGW_E_NULL;
        if (!aligned16(src[j])) return KGW_E_UNSUPPORTED;
        a.src[j] = src[j]; a.ids[j] = ids[j]; a.row0[j + 1] = a.row0[j] + n_rows[j];
    }
    a.rows = a.row0[n_jobs];
    if (a.rows == 0) return KGW_OK;
    a.W1 = W1; a.ldw1 = ldw1; a.b1 = b1; a.W2 = W2; a.ldw2 = ldw2; a.b2 = b2;
    a.Xg = Xg; a.H1 = H1; a.H2 = H2; a.ldo = ldo;
    const size_t lds = (size_t)(2 * 128 * WST + 256) * sizeof(float);
    static KgwPerDevice attr_once;
    if (attr_once.need()) {
        KGW_HIP(hipFuncSetAttribute((const void*)k_mlp2w_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int64_t ntiles = (a.rows + 31) / 32;
    k_mlp2w_fwd<<<(unsigned)((ntiles + 1) / 2), 256, lds, (hipStream_t)stream_>>>(a);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int64_t kgw_mlp2_bwd_first_workspace_floats(int64_t rows) {
    int64_t nblk = ((rows + 31) / 32 + 3) / 4;
    if (nblk > 256) nblk = 256;
    if (nblk < 1) nblk = 1;
    return nblk * 4096;
}

static int mlp2_bwd_first(const float* dH2, int64_t ldd, const float* W2, int64_t ldw2, const float* H1, int64_t ldh1,
                          const float* X, int64_t ldx, int32_t K1, int64_t rows, const int32_t* rows_dev, float* dW1,
                          int64_t ldw1, float* db1, float* workspace, int64_t workspace_floats, const int32_t* in_ids,
                          float* dZ, int64_t ldz, kgw_stream_t stream_, KgwGradSrc* defer, void* packed = nullptr, int flip = 0) {
    if (!dH2 || !W2 || !H1 || !db1 || !workspace) return KGW_E_NULL;
    if (defer && K1 > 0 && ldw1 != K1) return KGW_E_UNSUPPORTED;
    if (rows <= 0 || K1 < 0) return KGW_E_RANGE;
    if (K1 > 0 && (!X || !dW1)) return KGW_E_NULL;
    if (dZ && ((ldz & 3) || !aligned16(dZ))) return KGW_E_UNSUPPORTED;
    if (K1 > 31 || (ldd & 3) || (ldw2 & 3) || (ldh1 & 3) || !aligned16(dH2) || !aligned16(W2) || !aligned16(H1)) return KGW_E_UNSUPPORTED;
    int64_t nblk = ((rows + 31) / 32 + 3) / 4;
    if (nblk > 256) nblk = 256;
    if (workspace_floats < nblk * 4096) return KGW_E_RANGE;
    Mlp2BwdArgs a{dH2, ldd, W2, ldw2, H1, ldh1, X, ldx, workspace, rows, K1, rows_dev, in_ids, dZ, ldz, (uint4*)packed, flip};
    const size_t lds = (size_t)(128 * WST + 4 * 32 * TST) * sizeof(float);
    static KgwPerDevice attr_once;
    if (attr_once.need()) {
        KGW_HIP(hipFuncSetAttribute((const void*)k_mlp2_bwd_first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipStream_t st = (hipStream_t)stream_;
    static const bool split3 = !(getenv("KGW_MLP2_SPLIT") && getenv("KGW_MLP2_SPLIT")[0] == '0');
    if (packed && (!split3 || rows_dev || ((uintptr_t)packed & 15))) return KGW_E_UNSUPPORTED;     // (the image is written by k_mlp2_bwd_first3 only; every chunk)
    if (split3) {
        const size_t lds3 = (size_t)M3_W2_U4 * 16 + (size_t)(4 * 32 * TS2) * sizeof(float);
        static KgwPerDevice attr3_set;
        if (attr3_set.need()) {
            KGW_HIP(hipFuncSetAttribute((const void*)k_mlp2_bwd_first3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
            KGW_HIP(hipFuncSetAttribute((const void*)k_mlp2_bwd_first3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        }
        if (packed) k_mlp2_bwd_first3<true><<<(int)nblk, 256, lds3, st>>>(a);
        else k_mlp2_bwd_first3<false><<<(int)nblk, 256, lds3, st>>>(a);
    } else {
        k_mlp2_bwd_first<<<(int)nblk, 256, lds, st>>>(a);
    }
    KGW_LAUNCH_CHECK();
    if (defer) {                  // the blocks' partials are added by kgw_adam_fused, in k_mlp2_bwd_fold's order
        defer[0] = KgwGradSrc{};
        defer[1] = KgwGradSrc{};
        if (K1 > 0) { defer[0].ws = workspace; defer[0].kind = KGW_GRAD_MLP2_W; defer[0].nblk = (int)nblk; defer[0].K1 = K1; }
        defer[1].ws = workspace; defer[1].kind = KGW_GRAD_MLP2_B; defer[1].nblk = (int)nblk; defer[1].K1 = K1;
        return KGW_OK;
    }
    k_mlp2_bwd_fold<<<4096 / 64, 1024, 0, st>>>(workspace, (int)nblk, K1, dW1, ldw1, db1);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_mlp2_bwd_first(const float* dH2, int64_t ldd, const float* W2, int64_t ldw2, const float* H1, int64_t ldh1,
                                  const float* X, int64_t ldx, int32_t K1, int64_t rows, const int32_t* rows_dev, float* dW1,
                                  int64_t ldw1, float* db1, float* workspace, int64_t workspace_floats, const int32_t* in_ids,
                                  float* dZ, int64_t ldz, kgw_stream_t stream_) {
    return mlp2_bwd_first(dH2, ldd, W2, ldw2, H1, ldh1, X, ldx, K1, rows, rows_dev, dW1, ldw1, db1, workspace, workspace_floats,
                          in_ids, dZ, ldz, stream_, nullptr);
}

extern "C" int kgw_mlp2_bwd_first_partial(const float* dH2, int64_t ldd, const float* W2, int64_t ldw2, const float* H1,
                                          int64_t ldh1, const float* X, int64_t ldx, int32_t K1, int64_t rows,
                                          const int32_t* rows_dev, float* dW1, int64_t ldw1, float* db1, float* workspace,
                                          int64_t workspace_floats, const int32_t* in_ids, float* dZ, int64_t ldz, KgwGradSrc* src,
                                          kgw_stream_t stream_) {
    if (!src) return KGW_E_NULL;
    return mlp2_bwd_first(dH2, ldd, W2, ldw2, H1, ldh1, X, ldx, K1, rows, rows_dev, dW1, ldw1, db1, workspace, workspace_floats,
                          in_ids, dZ, ldz, stream_, src);
}

// ... with the masked dh1 rows written as kgw_gemm3's B operand image as well (packed: kgw_gemm3_packed_bytes(rows rounded up to
// 32) bytes; flip: kgw_gemm3_flip()) -- the k_g3_pack launch of the resident first layer's weight gradient and the fp32 rows it
// would read disappear (dZ may then be null).  src nullable: null finishes d b1 (/ d W1) with the fold launch.
extern "C" int kgw_mlp2_bwd_first_packed(const float* dH2, int64_t ldd, const float* W2, int64_t ldw2, const float* H1,
                                         int64_t ldh1, const float* X, int64_t ldx, int32_t K1, int64_t rows, float* dW1,
                                         int64_t ldw1, float* db1, float* workspace, int64_t workspace_floats,
                                         const int32_t* in_ids, float* dZ, int64_t ldz, void* packed, int32_t flip, KgwGradSrc* src,
                                         kgw_stream_t stream_) {
    if (!packed) return KGW_E_NULL;
    if (flip < 0 || (flip & (flip - 1))) return KGW_E_RANGE;
    return mlp2_bwd_first(dH2, ldd, W2, ldw2, H1, ldh1, X, ldx, K1, rows, nullptr, dW1, ldw1, db1, workspace, workspace_floats,
                          in_ids, dZ, ldz, stream_, src, packed, flip);
}

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
        static const int target = getenv("KGW_SPLITK_BLOCKS") ? atoi(getenv("KGW_SPLITK_BLOCKS")) : 512;
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
    static const int fused = getenv("KGW_SPLITK_FUSED") ? atoi(getenv("KGW_SPLITK_FUSED")) : 1;
    if (fused && a.KS > 1 && w_is_kn && N == 128) {        // forward transform: one launch
        SplitKJobs J{};
        J.j[0] = a; J.n = 1; J.blk0[0] = 0; J.blk0[1] = a.RT * 4;
        return splitk_fused_launch(J, (hipStream_t)stream_);
    }
    if (a.KS > 1 && (!workspace || workspace_floats < kgw_linear_splitk_workspace_floats(rows, K, N))) return KGW_E_NULL;
    // row-tile groups per slab: about two blocks per CU in total, at most one tile... at least one tile per block
    static const int target = getenv("KGW_SPLITK_BLOCKS") ? atoi(getenv("KGW_SPLITK_BLOCKS")) : 512;
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
            static const int target = getenv("KGW_SPLITK_BLOCKS") ? atoi(getenv("KGW_SPLITK_BLOCKS")) : 512;
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

// ======================================================================================================
// kgw_wmse: LD-score weighted MSE of the seed predictions, loss = mean(w[n_id] * (pred - y[n_id])^2) in float64
// (kgwas/kgwas.py:139-145: float32 residual and square, float64 weight, float64 mean), and its gradient.
// One block; fixed-order reduction.
// ======================================================================================================
namespace {

__global__ void __launch_bounds__(256) k_wmse_fwd(const float* __restrict__ pred, const int32_t* __restrict__ n_id,
                                                  const float* __restrict__ y, const double* __restrict__ w, int n,
                                                  double* __restrict__ loss) {
    __shared__ double sm[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int g = n_id[i];
        const float d = pred[i] - y[g];
        acc += w[g] * (double)(d * d);
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = sm[0] / (double)n;
}

__global__ void __launch_bounds__(256) k_wmse_bwd(const float* __restrict__ pred, const int32_t* __restrict__ n_id,
                                                  const float* __restrict__ y, const double* __restrict__ w, int n,
                                                  const double* __restrict__ gloss, float* __restrict__ dpred) {
    const double g0 = gloss[0] / (double)n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int g = n_id[i];
        const float d = pred[i] - y[g];
        dpred[i] = (float)(g0 * w[g]) * (2.0f * d);        // the float64 product meets the float32 square here
    }
}

}  // namespace

extern "C" int kgw_wmse_fwd(const float* pred, const int32_t* n_id, const float* y, const double* w, int32_t n,
                            double* loss, kgw_stream_t stream_) {
    if (!pred || !n_id || !y || !w || !loss) return KGW_E_NULL;
    if (n <= 0) return KGW_E_RANGE;
    k_wmse_fwd<<<1, 256, 0, (hipStream_t)stream_>>>(pred, n_id, y, w, n, loss);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_wmse_bwd(const float* pred, const int32_t* n_id, const float* y, const double* w, int32_t n,
                            const double* grad_loss, float* dpred, kgw_stream_t stream_) {
    if (!pred || !n_id || !y || !w || !grad_loss || !dpred) return KGW_E_NULL;
    if (n <= 0) return KGW_E_RANGE;
    k_wmse_bwd<<<(n + 255) / 256, 256, 0, (hipStream_t)stream_>>>(pred, n_id, y, w, n, grad_loss, dpred);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

// ======================================================================================================
// kgw_readout_wmse: read-out Linear(128 -> 1) (+ ReLU) of the seed rows (kgwas/model.py:86) fused with the
// LD-score weighted MSE (kgwas/kgwas.py:139-145).  One block: wavefront w takes seeds w, w+4, ...; partial sums are
// combined in a fixed order.  _bwd also produces the gradients of the read-out weight / bias and dH (zero for the
// rows beyond the seeds).
// ======================================================================================================
namespace {

// One wavefront per seed, four per block; per-seed / per-block partial results go to a scratch buffer and a second,
// single-block launch folds them in index order -- parallel across the chip, yet a fixed summation order.  (A
// "last block folds" hand-off inside one launch was tried: its device-scope fence cost more than the second launch.)
__global__ void __launch_bounds__(256) k_readout_wmse_fwd(const float* __restrict__ H, const float* __restrict__ wl,
                                                          const float* __restrict__ bl, const int32_t* __restrict__ n_id,
                                                          const float* __restrict__ y, const double* __restrict__ w, int n,
                                                          int relu, float* __restrict__ pred, double* __restrict__ terms) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n) return;
    const float2 w2 = ((const float2*)wl)[lane];
    const float2 h2 = ((const float2*)(H + (int64_t)i * KGW_C))[lane];
    float p = kgw_wave_allsum(fmaf(h2.x, w2.x, h2.y * w2.y)) + bl[0];
    if (relu) p = fmaxf(p, 0.f);
    if (lane == 0) {
        const int g = n_id[i];
        const float d = p - y[g];
        pred[i] = p;
        terms[i] = w[g] * (double)(d * d);
    }
}

__global__ void __launch_bounds__(256) k_fold_f64(const double* __restrict__ terms, int n, double* __restrict__ out) {
    __shared__ double sm[256];
    double acc = 0.0;
    for (int q = threadIdx.x; q < n; q += 256) acc += terms[q];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0] / (double)n;
}

__global__ void __launch_bounds__(256) k_readout_wmse_bwd(const float* __restrict__ H, const float* __restrict__ wl,
                                                          const float* __restrict__ pred, const int32_t* __restrict__ n_id,
                                                          const float* __restrict__ y, const double* __restrict__ w, int n,
                                                          int64_t rows, int relu, const double* __restrict__ gloss,
                                                          float* __restrict__ dH, float* __restrict__ part) {
    __shared__ float sw[4][KGW_C + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    float2 dw = make_float2(0.f, 0.f);
    float dp = 0.f;
    if (i < n) {
        const float2 w2 = ((const float2*)wl)[lane];
        const int g = n_id[i];
        const float p = pred[i];
        dp = (float)(gloss[0] / (double)n * w[g]) * (2.0f * (p - y[g]));
        if ((relu & 1) && !(p > 0.f)) dp = 0.f;
        const float2 h2 = ((const float2*)(H + i * KGW_C))[lane];
        // (bit 1 of `relu`: H itself is the output of a ReLU whose backward the caller folds in here: dH *= (H > 0))
        const bool mk = (relu & 2) != 0;
        ((float2*)(dH + i * KGW_C))[lane] = make_float2((!mk || h2.x > 0.f) ? dp * w2.x : 0.f,
                                                         (!mk || h2.y > 0.f) ? dp * w2.y : 0.f);
        dw = make_float2(dp * h2.x, dp * h2.y);
    } else if (i < rows) {
        ((float2*)(dH + i * KGW_C))[lane] = make_float2(0.f, 0.f);
    }
    if ((int64_t)blockIdx.x * 4 >= n) return;            // blocks without seeds hold no partial
    sw[wave][2 * lane] = dw.x; sw[wave][2 * lane + 1] = dw.y;
    if (lane == 0) sw[wave][KGW_C] = dp;
    __syncthreads();
    if (threadIdx.x <= KGW_C) {                    // block partial: 128 weight columns + the bias term
        const int c = threadIdx.x;
        part[(int64_t)blockIdx.x * (KGW_C + 1) + c] = (sw[0][c] + sw[1][c]) + (sw[2][c] + sw[3][c]);
    }
}

// d w_lin [128] and d b_lin from the per-block partials [nb][129]: 129 columns x 7 row groups of one block, fixed order
__global__ void __launch_bounds__(1024) k_readout_fold(const float* __restrict__ part, int nb, float* __restrict__ dwl,
                                                       float* __restrict__ dbl) {
    __shared__ float sm[7][KGW_C + 1];
    const int c = threadIdx.x % (KGW_C + 1), g = threadIdx.x / (KGW_C + 1);
    if (g < 7) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int q = g;
        for (; q + 21 < nb; q += 28) {
            a0 += part[(int64_t)q * (KGW_C + 1) + c];        a1 += part[(int64_t)(q + 7) * (KGW_C + 1) + c];
            a2 += part[(int64_t)(q + 14) * (KGW_C + 1) + c]; a3 += part[(int64_t)(q + 21) * (KGW_C + 1) + c];
        }
        for (; q < nb; q += 7) a0 += part[(int64_t)q * (KGW_C + 1) + c];
        sm[g][c] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (g == 0) {
        const float t = ((sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c])) + ((sm[4][c] + sm[5][c]) + sm[6][c]);
        if (c < KGW_C) dwl[c] = t; else dbl[0] = t;
    }
}

// Training step with a unit loss gradient (loss.backward()): forward and backward of the read-out in ONE launch per
// stage -- the per-seed stage computes prediction, loss term, d prediction, the dH row and the block's weight-gradient
// partial; the fold stage adds up the loss terms (float64, index order) and the partials.  Two launches instead of four.
__global__ void __launch_bounds__(256) k_readout_wmse_train(const float* __restrict__ H, const float* __restrict__ wl,
                                                            const float* __restrict__ bl, const int32_t* __restrict__ n_id,
                                                            const float* __restrict__ y, const double* __restrict__ w, int n,
                                                            int64_t rows, int relu, float* __restrict__ pred,
                                                            double* __restrict__ terms, float* __restrict__ dH,
                                                            float* __restrict__ part) {
    __shared__ float sw[4][KGW_C + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    float2 dw = make_float2(0.f, 0.f);
    float dp = 0.f;
    if (i < n) {
        const float2 w2 = ((const float2*)wl)[lane];
        const float2 h2 = ((const float2*)(H + i * KGW_C))[lane];
        float p = kgw_wave_allsum(fmaf(h2.x, w2.x, h2.y * w2.y)) + bl[0];
        if (relu & 1) p = fmaxf(p, 0.f);
        const int g = n_id[i];
        const float d = p - y[g];
        if (lane == 0) {
            pred[i] = p;
            terms[i] = w[g] * (double)(d * d);
        }
        dp = (float)(1.0 / (double)n * w[g]) * (2.0f * d);
        if ((relu & 1) && !(p > 0.f)) dp = 0.f;
        const bool mk = (relu & 2) != 0;
        ((float2*)(dH + i * KGW_C))[lane] = make_float2((!mk || h2.x > 0.f) ? dp * w2.x : 0.f,
                                                         (!mk || h2.y > 0.f) ? dp * w2.y : 0.f);
        dw = make_float2(dp * h2.x, dp * h2.y);
    } else if (i < rows) {
        ((float2*)(dH + i * KGW_C))[lane] = make_float2(0.f, 0.f);
    }
    if ((int64_t)blockIdx.x * 4 >= n) return;            // blocks without seeds hold no partial
    sw[wave][2 * lane] = dw.x; sw[wave][2 * lane + 1] = dw.y;
    if (lane == 0) sw[wave][KGW_C] = dp;
    __syncthreads();
    if (threadIdx.x <= KGW_C) {
        const int c = threadIdx.x;
        part[(int64_t)blockIdx.x * (KGW_C + 1) + c] = (sw[0][c] + sw[1][c]) + (sw[2][c] + sw[3][c]);
    }
}

__global__ void __launch_bounds__(1024) k_readout_train_fold(const float* __restrict__ part, int nb, const double* __restrict__ terms,
                                                             int n, float* __restrict__ dwl, float* __restrict__ dbl,
                                                             double* __restrict__ loss) {
    __shared__ float sm[7][KGW_C + 1];
    __shared__ double sd[256];
    const int c = threadIdx.x % (KGW_C + 1), g = threadIdx.x / (KGW_C + 1);
    if (g < 7) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int q = g;
        for (; q + 21 < nb; q += 28) {
            a0 += part[(int64_t)q * (KGW_C + 1) + c];        a1 += part[(int64_t)(q + 7) * (KGW_C + 1) + c];
            a2 += part[(int64_t)(q + 14) * (KGW_C + 1) + c]; a3 += part[(int64_t)(q + 21) * (KGW_C + 1) + c];
        }
        for (; q < nb; q += 7) a0 += part[(int64_t)q * (KGW_C + 1) + c];
        sm[g][c] = (a0 + a1) + (a2 + a3);
    }
    if (threadIdx.x < 256) {                                  // the loss: same order as k_fold_f64
        double acc = 0.0;
        for (int q = threadIdx.x; q < n; q += 256) acc += terms[q];
        sd[threadIdx.x] = acc;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sd[threadIdx.x] += sd[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = sd[0] / (double)n;
    if (g == 0) {
        const float t = ((sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c])) + ((sm[4][c] + sm[5][c]) + sm[6][c]);
        if (c < KGW_C) dwl[c] = t; else dbl[0] = t;
    }
}

}  // namespace

extern "C" int kgw_readout_wmse_train_parts(const float* H, const float* w_lin, const float* b_lin, const int32_t* n_id,
                                            const float* y, const double* w, int32_t n, int64_t rows, int32_t relu, float* pred,
                                            double* loss, float* dH, float* dw_lin, float* db_lin, double* terms, float* scratch,
                                            KgwReadoutFold* fold_out, kgw_stream_t stream_) {
    if (!H || !w_lin || !b_lin || !n_id || !y || !w || !pred || !loss || !dH || !dw_lin || !db_lin || !terms || !scratch || !fold_out)
        return KGW_E_NULL;
    if (n <= 0 || rows < n) return KGW_E_RANGE;
    k_readout_wmse_train<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream_>>>(H, w_lin, b_lin, n_id, y, w, n, rows, relu, pred,
                                                                                        terms, dH, scratch);
    KGW_LAUNCH_CHECK();
    *fold_out = KgwReadoutFold{scratch, terms, dw_lin, db_lin, loss, (n + 3) / 4, n};
    return KGW_OK;
}

extern "C" int kgw_readout_train_fold(const KgwReadoutFold* f, kgw_stream_t stream_) {
    if (!f || !f->scratch || !f->terms || !f->dw_lin || !f->db_lin || !f->loss) return KGW_E_NULL;
    if (f->n <= 0 || f->nb <= 0) return KGW_E_RANGE;
    k_readout_train_fold<<<1, 1024, 0, (hipStream_t)stream_>>>(f->scratch, f->nb, f->terms, f->n, f->dw_lin, f->db_lin, f->loss);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_readout_wmse_fwd(const float* H, const float* w_lin, const float* b_lin, const int32_t* n_id,
                                    const float* y, const double* w, int32_t n, int32_t relu, float* pred,
                                    double* loss, double* scratch, kgw_stream_t stream_) {
    if (!H || !w_lin || !b_lin || !n_id || !y || !w || !pred || !loss || !scratch) return KGW_E_NULL;
    if (n <= 0) return KGW_E_RANGE;
    hipStream_t st = (hipStream_t)stream_;
    k_readout_wmse_fwd<<<(n + 3) / 4, 256, 0, st>>>(H, w_lin, b_lin, n_id, y, w, n, relu, pred, scratch);
    KGW_LAUNCH_CHECK();
    k_fold_f64<<<1, 256, 0, st>>>(scratch, n, loss);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_readout_wmse_bwd(const float* H, const float* w_lin, const float* pred, const int32_t* n_id,
                                    const float* y, const double* w, int32_t n, int64_t rows, int32_t relu,
                                    const double* grad_loss, float* dH, float* dw_lin, float* db_lin, float* scratch,
                                    kgw_stream_t stream_) {
    if (!H || !w_lin || !pred || !n_id || !y || !w || !grad_loss || !dH || !dw_lin || !db_lin || !scratch)
        return KGW_E_NULL;
    if (n <= 0 || rows < n) return KGW_E_RANGE;
    hipStream_t st = (hipStream_t)stream_;
    k_readout_wmse_bwd<<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(H, w_lin, pred, n_id, y, w, n, rows, relu, grad_loss, dH,
                                                                    scratch);
    KGW_LAUNCH_CHECK();
    k_readout_fold<<<1, 1024, 0, st>>>(scratch, (n + 3) / 4, dw_lin, db_lin);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_readout_wmse_train(const float* H, const float* w_lin, const float* b_lin, const int32_t* n_id,
                                      const float* y, const double* w, int32_t n, int64_t rows, int32_t relu, float* pred,
                                      double* loss, float* dH, float* dw_lin, float* db_lin, double* terms, float* scratch,
                                      kgw_stream_t stream_) {
    if (!H || !w_lin || !b_lin || !n_id || !y || !w || !pred || !loss || !dH || !dw_lin || !db_lin || !terms || !scratch)
        return KGW_E_NULL;
    if (n <= 0 || rows < n) return KGW_E_RANGE;
    hipStream_t st = (hipStream_t)stream_;
    // (round 4, measured and dropped: the whole node as ONE block of 16 wavefronts walking the 512 rows -- no partial buffer, no
    //  fold launch -- ran the step 40 - 45 us SLOWER: 32 dependent row trips per wavefront instead of one)
    k_readout_wmse_train<<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(H, w_lin, b_lin, n_id, y, w, n, rows, relu, pred, terms, dH,
                                                                      scratch);
    KGW_LAUNCH_CHECK();
    k_readout_train_fold<<<1, 1024, 0, st>>>(scratch, (n + 3) / 4, terms, n, dw_lin, db_lin, loss);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

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
