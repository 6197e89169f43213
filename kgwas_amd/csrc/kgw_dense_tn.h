// kgw_dense_tn.h -- part of kgw_dense.hip (ONE translation unit, split by kernel family in round 6; include order matters:
// later families use device functions of earlier ones): weight-gradient products C = A^T B over tall inputs (kgw_tn_gemm*), their two-stage reduction and riding second launches.
#pragma once

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
        const int64_t cap_small = 512;               // (round 6, whole step: 256 / 512 / 1 024 measured 1.0035 / 0.9993 / 1.0086 ms)
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
