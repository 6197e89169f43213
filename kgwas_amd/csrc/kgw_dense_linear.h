// kgw_dense_linear.h -- part of kgw_dense.hip (ONE translation unit, split by kernel family in round 6; include order matters:
// later families use device functions of earlier ones): Linear forward / dX (kgw_linear: LDS-tiled, weight-resident, weights-in-registers) and the fused two-layer feature MLP kernels (kgw_mlp2_fwd, kgw_mlp2w_fwd, kgw_mlp2_bwd_first).
#pragma once

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
    // product 1: THIS wavefront's two column tiles t0, t0 + 1 (round 6; until then both wavefronts of a row tile computed all four:
    // 256 MFMAs each, half of them twice); the other two come from the partner wavefront through LDS below.  Operands
    // W1[32 t + li][64 lk + 4 q + c] from LDS, a step ahead
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const float* w1p = W1l + (t0 * 32 + li) * WST + lk * 64;
    f32x4 wn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) wn[t] = *(const f32x4*)(w1p + t * 32 * WST);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        f32x4 w[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) w[t] = wn[t];
        if (q + 1 < 16) {
#pragma unroll
            for (int t = 0; t < 2; ++t) wn[t] = *(const f32x4*)(w1p + t * 32 * WST + 4 * (q + 1));
        }
#define KGW_MLPW_STEP(C)                                                                                  \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                    \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].C, xa[q].C, acc[t], 0, 0, 0);
        KGW_MLPW_STEP(x) KGW_MLPW_STEP(y) KGW_MLPW_STEP(z) KGW_MLPW_STEP(w)
#undef KGW_MLPW_STEP
    }
    // h1 = relu(. + b1) of my tiles: accumulator element 4 g + c of tile t = column 32 (t0 + t) + 8 g + 4 lk + c of this lane's row
    f32x4 xh[8];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *(const f32x4*)(bl + (t0 + t) * 32 + 8 * g + 4 * lk);
            f32x4 v;
            v.x = fmaxf(acc[t][4 * g + 0] + b4.x, 0.f); v.y = fmaxf(acc[t][4 * g + 1] + b4.y, 0.f);
            v.z = fmaxf(acc[t][4 * g + 2] + b4.z, 0.f); v.w = fmaxf(acc[t][4 * g + 3] + b4.w, 0.f);
            xh[4 * t + g] = v;
        }
    if (live) {                                              // (each half writes its own 64 columns of h1)
        float* hp = a.H1 + row * a.ldo + 4 * lk;
#pragma unroll
        for (int q = 0; q < 8; ++q) *(f32x4*)(hp + 32 * (t0 + (q >> 2)) + 8 * (q & 3)) = xh[q];
    }
    // exchange with the partner wavefront (same rows, the other two tiles) through the LDS that held W1: lane l of both maps to
    // the same (row, k group), so slot [wavefront][q][lane] is read back by the same lane of the partner -- no bank conflicts
    __syncthreads();                                         // (every wavefront is done reading W1)
    {
        f32x4* ex = (f32x4*)W1l;
#pragma unroll
        for (int q = 0; q < 8; ++q) ex[(wave * 8 + q) * 64 + lane] = xh[q];
    }
    __syncthreads();
    {
        const f32x4* ex = (const f32x4*)W1l + ((wave ^ 1) * 8) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) {                        // (static register indices: half 0 owns tiles 0 / 1, half 1 tiles 2 / 3)
            const f32x4 pq = ex[q * 64];
            xa[q] = half ? pq : xh[q];
            xa[8 + q] = half ? xh[q] : pq;
        }
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
    const int64_t half_max = 1024;
    // (measured in the step, rounds 1-2: 1.648 ms with the half-tile kernel up to 512 tiles, 1.653 up to 1024, 1.671 without it;
    //  round 6, today's step: up to 1024 -- the gene MLP's 626 tiles of 20 032 rows take it too -- 0.9865 -> 0.9844 ms, A/B)
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
    const int64_t wres_min = 4096;
    const int64_t wres_tall = 32768;
    if (K <= 128 && N <= 128 && rows >= wres_min && (N & 3) == 0 && (ldy & 3) == 0 && aligned16(Y) && aligned16(bias) &&
        (!mask || ((ldm & 3) == 0 && aligned16(mask)))) {           // weight-resident persistent kernel (tall inputs)
        hipStream_t st = (hipStream_t)stream_;
        // measured (MI355X) against the LDS-staged kernel below: 15-20 % faster under 32 k rows, 5-10 % faster with a ReLU
        // mask, equal otherwise
        const int64_t wreg_min = 4096;
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
