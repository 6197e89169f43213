// kgw_gemm3.hip -- C[M, 128] = A[M, K] * B[K, 128] for a tall RESIDENT fp32 matrix A: the first gene Linear
// (kgwas/model.py:13,19 -- FC_hidden over the 5 120-wide gene features, kgwas_data.py:236,244) forward, A = X, B = W1^T, and
// its weight gradient, A = X^T (a second resident copy), B = dz.  2 x 26 GFLOP per step: on the fp32 matrix pipe (157 TF)
// that is 0.17 ms each at 100 %.
//
// Here the product runs on the bf16 matrix pipe WITHOUT narrowing the arithmetic.  Every fp32 operand is split EXACTLY into
// three bf16 pieces, a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1), a3 = a - a1 - a2: 3 x 8 significand bits = the 24 of
// fp32, the residuals are exact in fp32), and of the nine piece products the six of weight >= 2^-16,
//      a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1),
// are accumulated in fp32 (each bf16 x bf16 product is exact in fp32).  The three dropped ones are bounded by
// 3 * 2^-25 |a||b| per term of the dot product -- below the rounding of ONE fp32 multiply-add (2^-24 |a||b|), i.e. the
// result differs from an fp32 chain by less than the fp32 chain's own error bound (tests/test_gpu_gemm3.py measures both
// against float64: the split product is 3x CLOSER to float64 than the library's fp32 product on the benchmark shapes).
// 6 x 26 GFLOP on the 2.5 PF pipe = 63 us at the nominal clock and reading A once (410 MB) ~ 75 us.  Measured (MI355X, 20 032 x
// 5 120 x 128): 153 us forward, 172 us for the weight-gradient shape = 1.03 PFLOP/s of bf16 work = 172 TFLOP/s of the fp32
// product it replaces (the fp32 matrix peak is 157; the tuned library product takes 212 / 200 us); 143 us with 256-row blocks.  Where the rest goes
// (ablations, tools/micro/mfma_rate.hip): with random operands v_mfma_f32_32x32x16_bf16 sustains 18-19 ns per instruction
// and SIMD (1.75-2.0 PF, power), the kernel's MFMA stream alone runs 110 us on its busiest CUs (two blocks), its memory side
// alone 90-95 us, and the two overlap to 153.
//
// Shape of the kernel.  Work item = (256-row tile of A, K range); 8 wavefronts (two per SIMD, one block per CU), each owns
// 32 rows x 128 columns (four 32x32x16 accumulators) and all share the B tile (128-row blocks, two per CU, read B from L2
// twice as often: 153 vs 143 us).  K advances in chunks of 32:
//   A: a wavefront loads its 32 x 32 fp32 block coalesced (8 rows x 128 B per instruction, non-temporal), two chunks ahead,
//      writes it to a wavefront-PRIVATE 4 KB LDS tile (XOR-swizzled 16-byte slots: ds_write_b128 and ds_read_b128 conflict
//      free) and reads it back in MFMA operand layout (lane = row, 8 consecutive k) -- no block barrier on this path; the
//      three-way split is 44 VALU instructions per 24 MFMAs, done on the operand registers;
//   B: split once per call by kgw_gemm3_pack_* into the LDS image itself ([chunk][k16 step][piece][column tile][lane] x 16 B,
//      lane-linear: a straight copy, conflict-free ds_read_b128), double buffered, one barrier per chunk.
// Partial products go to a workspace [split][M][128]; kgw_gemm3 ends with a fixed-order reduction (+ bias, ReLU, or a
// transposed store for the weight gradient).  Deterministic: no atomics, fixed K ranges.
#include "kgw_common.h"
#include "kgw_riders.h"
#include <stdlib.h>

typedef kgw_bf8 g3_bf8;
typedef __attribute__((ext_vector_type(16))) float g3_f16;
typedef __attribute__((ext_vector_type(4))) float g3_f4;
typedef __attribute__((ext_vector_type(4))) unsigned g3_u4;

static constexpr int G3_CH_U4 = 1536;          // uint4 per packed chunk of 32 k: 2 steps x 3 pieces x 4 column tiles x 64 lanes
static constexpr int G3_FLIP = 16;             // chunks (of 32 k) per sign period of the accumulation, a power of two -- see k_g3_gemm
// (periods of 4 chunks measured the same error and 2 - 10 us per step of pipe drains, 16 nothing: round 4; the A/B knob is gone)
static int g3_flip() { return G3_FLIP; }
static constexpr int G3_MAX_ITEMS = 512;       // 128-row blocks resident at once (two per CU on 256 CUs; 256-row blocks: half)

// ---- B operand packing ------------------------------------------------------------------------------------------------------
// image index (((c * 2 + j) * 3 + p) * 4 + nt) * 64 + lane: the eight bf16 of piece p for k = 32 c + 16 j + 8 (lane >> 5) + i,
// column 32 nt + (lane & 31).  One thread per (c, j, nt, lane).  k >= kv reads as zero (S holds kv rows / columns only: a gene
// count or a feature width that is not a multiple of 32 -- the A operand then carries zero padding there as well).
template <bool KN>
__global__ void __launch_bounds__(256) k_g3_pack(const float* __restrict__ S, long lds_, int K, long kv, int vec, uint4* __restrict__ out, int flip) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)K * 16) return;
    const int lane = (int)(t & 63), nt = (int)((t >> 6) & 3), j = (int)((t >> 8) & 1);
    const long c = t >> 9;
    const int n = 32 * nt + (lane & 31);
    const long k0 = 32 * c + 16 * j + 8 * (lane >> 5);
    float x[8];
    if (KN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = k0 + i < kv ? S[(k0 + i) * lds_ + n] : 0.f;
    } else if (vec && k0 + 8 <= kv) {
        const float4 u = *(const float4*)(S + n * lds_ + k0), v = *(const float4*)(S + n * lds_ + k0 + 4);
        x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = k0 + i < kv ? S[n * lds_ + k0 + i] : 0.f;
    }
    // sign period of the accumulation (k_g3_gemm): the chunks of every other period are packed NEGATED (exact: the three
    // pieces of -x are the negated pieces of x, bf16 rounding is symmetric)
    if (flip && ((c / flip) & 1)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = -x[i];
    }
    uint4 p1, p2, p3;
    kgw_split3x8(x, p1, p2, p3);
    uint4* o = out + ((c * 2 + j) * 3 * 4 + nt) * 64 + lane;
    o[0] = p1; o[4 * 64] = p2; o[8 * 64] = p3;
}

// ---- the product ---------------------------------------------------------------------------------------------------------------
struct G3Args {
    const float* A; long lda; int M, K;          // lda == 0: A in 32 x 32 tiles, [ceil(M / 32)][K / 32][32][32]
    const uint4* Bp;
    float* ws;
    int nsplit, n_tiles, per_xcd, flip;
};

// Sign periods (round 4).  The bf16 MFMA's internal add TRUNCATES (toward -inf: the error of a long accumulation has a negative
// mean whatever the sign of the sum -- measured: K = 5 120, results ~57, mean error -5.2e-6 against -2.4e-7 for the fp32 pipe).
// The accumulator therefore changes sign every G3_FLIP chunks: it holds s * (partial sum) with s = (-1)^(chunk / G3_FLIP), the
// packed B of those chunks is negated to match (s a b accumulates onto s S), and the flip itself is an exact sign change of the
// accumulator registers -- 64 VALU per 4 x 48 MFMAs, no extra registers.  The truncation then pulls the partial sum down in one
// period and up in the next: the bias of adjacent periods cancels.  Measured (tests/test_gpu_gemm3.py, positive operands, results
// ~57, forward / weight-gradient shape): mean error -2.4e-6 / -2.3e-6 without sign periods, -6e-8 / -3e-8 with periods of 4 or 8
// chunks, +3e-9 / -5e-8 with 16 (the fp32 pipe's own: -3.3e-7 / -1.2e-6); mean |error| 1.07e-5 / 5.8e-6 against the fp32
// product's 4.1e-5 / 3.9e-5.  Step time: periods of 16 chunks are free, periods of 4 cost 2 - 10 us per step (the flip drains the
// MFMA pipe) -- hence 16.
//
// Both LDS tiles are double buffered in SEPARATE arrays (the compiler then knows that the stores of chunk c + 1 do not alias
// the operand reads of chunk c).  MT = 32-row tiles per wavefront (1: 128-row blocks, two per CU; 2 was measured too -- 256-row
// blocks, one per CU, each B operand feeding two MFMAs: 157-180 us against 153 -- and is not instantiated).
struct G3KArgs { G3Args a; G3Riders R; };     // ONE kernel argument: the riders' tables are read where they lie, in the kernarg segment

template <int MT, int NW = 4>
__global__ void __launch_bounds__(64 * NW, (8 / NW) / MT) k_g3_gemm(G3KArgs ka_) {
    const G3Args& a = ka_.a;
    constexpr int NT = 64 * NW, NB = G3_CH_U4 / NT;   // threads, B staging registers (uint4) per thread
    constexpr int NQ = 4 * MT;                 // A load instructions per chunk (8 rows x 128 B each)
    constexpr int AT = 256 * MT;               // 16-byte slots per wavefront-private A tile
    __shared__ g3_u4 smB0[G3_CH_U4], smB1[G3_CH_U4];
    __shared__ g3_f4 smA0[NW * AT], smA1[NW * AT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block -> work item: workgroups go round-robin to the 8 XCDs (blockIdx % 8), and XCD x takes the contiguous range
    // [x per_xcd, (x + 1) per_xcd) of the items in (K range, row tile) order -- one or two K ranges per XCD, so the packed B
    // of a range (1.3 MB of the 15 MB in the weight-gradient product) is fetched into that XCD's L2 once and hit by the rest
    // rider blocks (kgw_riders.h): the launch's blocks beyond the product's own take the step's parameter-only forward work on
    // the compute units the product leaves idle (240 blocks of one CU each on 256 CUs at the benchmark shape)
    if ((int)blockIdx.x >= a.per_xcd * 8) {
        // (the rider functions are real calls -- their registers are their own, the product's loop keeps its allocation -- and
        //  take their tables by ADDRESS in the kernarg segment: a by-value struct whose address escapes would be copied to scratch)
        const char* kseg = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
        const G3Riders* R = (const G3Riders*)(kseg + __builtin_offsetof(G3KArgs, R));
        if (ka_.R.n_blocks) g3_param_riders<NW>(*R, (int)blockIdx.x - a.per_xcd * 8);
        return;
    }
    const int item = (int)(blockIdx.x & 7) * a.per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= a.per_xcd || item >= a.n_tiles * a.nsplit) return;
    const int split = item / a.n_tiles, tile = item - split * a.n_tiles;
    const int nch = a.K >> 5;
    const int c0 = (int)((long)nch * split / a.nsplit), c1 = (int)((long)nch * (split + 1) / a.nsplit);
    const int nc = c1 - c0;
    const int row0 = tile * (32 * NW * MT) + wave * (32 * MT);

    // A addressing: instruction q covers rows 8 q .. 8 q + 7 of the wavefront's rows, 128 contiguous bytes each; 16-byte
    // slot s of row r lives at slot s ^ ((r >> 1) & 7) of its 128-byte LDS row
    const long cstep = a.lda ? 32 : 1024;      // floats between consecutive chunks of a row
    const float* ap[NQ];
    int aw[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int r = 8 * q + (lane >> 3);
        int row = row0 + r;
        row = row < a.M ? row : a.M - 1;
        // row-major: 128 contiguous bytes of each of 8 rows per instruction; tiled: the wavefront's 32 x 32 block of a chunk
        // is 4 KB contiguous (1 KB per instruction), the next chunk follows it
        ap[q] = a.lda ? a.A + (long)row * a.lda + (long)c0 * 32 + (lane & 7) * 4
                      : a.A + ((long)(row >> 5) * nch + c0) * 1024 + (row & 31) * 32 + (lane & 7) * 4;
        aw[q] = wave * AT + r * 8 + ((lane & 7) ^ ((r >> 1) & 7));
    }
    const int m = lane & 31, g = lane >> 5, sw = (m >> 1) & 7;
    const g3_u4* bp = (const g3_u4*)a.Bp + (long)c0 * G3_CH_U4 + tid;

    g3_f4 ra[NQ], rb[NQ];
    g3_u4 bs[NB];
    g3_f16 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

    auto load_a = [&](g3_f4 (&r)[NQ], int c) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) r[q] = __builtin_nontemporal_load((const g3_f4*)(ap[q] + (long)c * cstep));
    };
    auto store_a = [&](const g3_f4 (&r)[NQ], g3_f4* A) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) A[aw[q]] = r[q];
    };
    auto load_b = [&](int c) {
#pragma unroll
        for (int i = 0; i < NB; ++i) bs[i] = bp[(long)c * G3_CH_U4 + i * NT];
    };
    auto store_b = [&](g3_u4* B) {
#pragma unroll
        for (int i = 0; i < NB; ++i) B[tid + i * NT] = bs[i];
    };
    auto compute = [&](const g3_f4* A_, const g3_u4* B_) {
        const g3_u4* B = B_ + lane;
        const g3_f4* A = A_ + wave * AT + m * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int seg = 2 * (2 * j + g);
            g3_bf8 ap_[MT][3];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const g3_f4 u = A[mt * 256 + (seg ^ sw)], v = A[mt * 256 + ((seg + 1) ^ sw)];
                const float x[8] = {u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
                uint4 p1, p2, p3;
                kgw_split3x8(x, p1, p2, p3);
                ap_[mt][0] = __builtin_bit_cast(g3_bf8, p1);
                ap_[mt][1] = __builtin_bit_cast(g3_bf8, p2);
                ap_[mt][2] = __builtin_bit_cast(g3_bf8, p3);
            }
            g3_bf8 b[3][4];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) b[p][nt] = __builtin_bit_cast(g3_bf8, B[((j * 3 + p) * 4 + nt) * 64]);
            // (piece of A, piece of B), smallest products first; 4 MT independent accumulators between two uses of one
            constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap_[mt][TA[t]], b[TB[t]][nt], acc[mt][nt], 0, 0, 0);
        }
    };

    // steady state without branches: chunk indices past the end are clamped (a redundant load of the last chunk into a
    // buffer nobody reads), so the load counters stay exact.  Iteration c: A of chunk c + 2 leaves for registers, chunk
    // c + 1 (registers since iteration c - 1) goes to the other LDS buffers, chunk c is multiplied.
    auto flip = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = -acc[mt][nt][i];
    };
    const int last = nc - 1;
    load_a(ra, 0);
    load_b(0);
    store_a(ra, smA0);
    store_b(smB0);
    load_a(ra, min(1, last));
    load_b(min(1, last));
    __syncthreads();
    for (int c = 0; c < nc; c += 2) {
        load_a(rb, min(c + 2, last));
        store_a(ra, smA1);
        store_b(smB1);
        load_b(min(c + 2, last));
        if (a.flip && c && !((c0 + c) & (a.flip - 1))) flip();  // (wavefront-uniform; at c == 0 the accumulator is zero)
        compute(smA0, smB0);
        __syncthreads();
        if (c + 1 >= nc) break;
        load_a(ra, min(c + 3, last));
        store_a(rb, smA0);
        store_b(smB0);
        load_b(min(c + 3, last));
        if (a.flip && !((c0 + c + 1) & (a.flip - 1))) flip();
        compute(smA1, smB1);
        __syncthreads();
    }

    // accumulator register r of a 32x32 tile: row 8 (r / 4) + 4 g + r % 4, column lane & 31
    float* wp = a.ws + ((long)split * a.M) * 128 + m;
    if (a.flip && (((c1 - 1) / a.flip) & 1)) flip();      // the sign the accumulator ends in
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + 32 * mt + 8 * (r >> 2) + 4 * g + (r & 3);
            if (row < a.M) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) wp[(long)row * 128 + 32 * nt] = acc[mt][nt][r];
            }
        }
}

// out[row, :] = act(sum_s ws[s, row, :] + bias), splits added in index order
__global__ void __launch_bounds__(256) k_g3_reduce(const float* __restrict__ ws, int nsplit, long M, const float* __restrict__ bias,
                                                   int relu, float* __restrict__ out, long ldo, const int32_t* __restrict__ row_map,
                                                   float* __restrict__ out_rows, long ld_rows, long out_rows_n,
                                                   const int32_t* __restrict__ out_rows_real) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (out_rows_real) {
        // rows [*out_rows_real, out_rows_n) of the batch's row block: the padding of a static layout, which no resident row maps to
        // -- zeroed here instead of by a fill launch over the whole block ahead of the product
        const long real = *out_rows_real < 0 ? 0 : *out_rows_real;
        const long zr = real + (t >> 5);
        if (zr < out_rows_n) *(float4*)(out_rows + zr * ld_rows + 4 * (t & 31)) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (t >= M * 32) return;
    const long row = t >> 5;
    const int c4 = (int)(t & 31);
    const float4* w = (const float4*)ws + t;
    float4 s = w[0];
    for (int k = 1; k < nsplit; ++k) {
        const float4 v = w[(long)k * M * 32];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) {
        const float4 b = ((const float4*)bias)[c4];
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
    *(float4*)(out + row * ldo + 4 * c4) = s;
    if (row_map) {                       // the batch's copy of the row (the loader's x[n_id] slicing of the layer's output)
        const int l = row_map[row];
        if (l >= 0) *(float4*)(out_rows + (long)l * ld_rows + 4 * c4) = s;
    }
}

// out[col, row] = sum_s ws[s, row, col]: 32-row x 32-column tiles through LDS (grid: row tiles x 4 column tiles)
__global__ void __launch_bounds__(256) k_g3_reduce_t(const float* __restrict__ ws, int nsplit, long M, float* __restrict__ out, long ldo) {
    __shared__ float tl[32][33];
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * 32;
    const int cb = blockIdx.y * 32;
    {
        const int r = tid >> 3, c4 = tid & 7;          // one float4 per thread
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + r < M) {
            const float4* w = (const float4*)ws + (r0 + r) * 32 + (cb >> 2) + c4;
            s = w[0];
            for (int k = 1; k < nsplit; ++k) {
                const float4 v = w[(long)k * M * 32];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        tl[r][4 * c4] = s.x; tl[r][4 * c4 + 1] = s.y; tl[r][4 * c4 + 2] = s.z; tl[r][4 * c4 + 3] = s.w;
    }
    __syncthreads();
    const int r = tid & 31;
    if (r0 + r < M) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = (tid >> 5) + 8 * i;
            out[(long)(cb + col) * ldo + r0 + r] = tl[r][col];
        }
    }
}

// ---- C ABI ---------------------------------------------------------------------------------------------------------------------
static int g3_nw() { return 8; }               // wavefronts per block (four, 128-row blocks two per CU, read B from L2 twice as often: 151 us against 136)

static int g3_splits(int64_t M, int64_t K) {
    const int rt = 32 * g3_nw();
    const int64_t tiles = (M + rt - 1) / rt, nch = K / 32;
    int64_t s = (G3_MAX_ITEMS * 4 / g3_nw()) / tiles;
    if (s > nch / 8) s = nch / 8;
    if (s > 16) s = 16;
    if (s < 1) s = 1;
    return (int)s;
}

extern "C" int64_t kgw_gemm3_packed_bytes(int64_t K) { return K > 0 && K % 32 == 0 ? (K / 32) * G3_CH_U4 * 16 : 0; }

extern "C" int64_t kgw_gemm3_workspace_floats(int64_t M, int64_t K) {
    return M > 0 && K > 0 ? (int64_t)g3_splits(M, K) * M * 128 : 0;
}

extern "C" int kgw_gemm3_pack(const float* S, int64_t lds_, int64_t K, int64_t k_valid, int32_t s_is_kn, void* packed,
                              kgw_stream_t stream_) {
    if (!S || !packed) return KGW_E_NULL;
    if (K <= 0 || k_valid < 0 || k_valid > K) return KGW_E_RANGE;
    if (K % 32 || ((uintptr_t)packed & 15)) return KGW_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream_;
    const int64_t nthr = K * 16;
    const int vec = !(lds_ & 3) && !((uintptr_t)S & 15);       // 16-byte loads along k (S = B^T); else element loads
    if (s_is_kn) k_g3_pack<true><<<(int)((nthr + 255) / 256), 256, 0, st>>>(S, lds_, (int)K, (long)k_valid, 0, (uint4*)packed, g3_flip());
    else k_g3_pack<false><<<(int)((nthr + 255) / 256), 256, 0, st>>>(S, lds_, (int)K, (long)k_valid, vec, (uint4*)packed, g3_flip());
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_gemm3_flip(void) { return g3_flip(); }

// transpose_out product without k_g3_reduce_t: kgw_adam_fused adds the K ranges (same order) tile by tile while it updates the
// parameter the gradient belongs to
extern "C" int kgw_gemm3_partial(const float* A, int64_t lda, int64_t M, int64_t K, const void* packed, float* workspace,
                                 int64_t workspace_floats, float* out, int64_t ldo, KgwGradSrc* src, kgw_stream_t stream_) {
    if (!A || !packed || !workspace || !out || !src) return KGW_E_NULL;
    if (M <= 0 || K <= 0 || M > (1 << 30) || K > (1 << 30)) return KGW_E_RANGE;
    if (K % 32 || M % 32 || ldo != M || lda < 0 || (lda & 3) || ((uintptr_t)A & 15) || ((uintptr_t)packed & 15) || ((uintptr_t)workspace & 15))
        return KGW_E_UNSUPPORTED;
    const int ns = g3_splits(M, K);
    if (workspace_floats < (int64_t)ns * M * 128) return KGW_E_RANGE;
    const int rt = 32 * g3_nw();
    const int tiles = (int)((M + rt - 1) / rt);
    const int per_xcd = (tiles * ns + 7) / 8;
    G3Args a{A, (long)lda, (int)M, (int)K, (const uint4*)packed, workspace, ns, tiles, per_xcd, g3_flip()};
    const G3Riders none{};
    if (g3_nw() == 8) k_g3_gemm<1, 8><<<per_xcd * 8, 512, 0, (hipStream_t)stream_>>>(G3KArgs{a, none});
    else k_g3_gemm<1, 4><<<per_xcd * 8, 256, 0, (hipStream_t)stream_>>>(G3KArgs{a, none});
    KGW_LAUNCH_CHECK();
    *src = KgwGradSrc{};
    src->ws = workspace; src->kind = KGW_GRAD_G3T; src->nblk = ns; src->M = (int32_t)M; src->N = 128;
    return KGW_OK;
}

// compute units the product of an [M, K] x [K, 128] launch leaves idle (one 512-thread block per CU): where rider blocks run for
// free.  0: none worth using (the 256-thread variant, a grid that fills the chip, fewer than 8 idle CUs).
static int g3_free_cus(int64_t M, int64_t K) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n_cu = p.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    if (g3_nw() != 8 || M <= 0 || K <= 0) return 0;
    const int ns = g3_splits(M, K);
    const int tiles = (int)((M + 255) / 256);
    const int grid = ((tiles * ns + 7) / 8) * 8;
    int idle = n_cu - grid;
    if (idle > 16) idle = 16;
    return idle >= 8 ? idle : 0;
}

extern "C" int kgw_gemm3_rider_blocks(int64_t M, int64_t K) { return g3_free_cus(M, K); }

static int gemm3_launch(const float* A, int64_t lda, int64_t M, int64_t K, const void* packed, float* workspace,
                        int64_t workspace_floats, const float* bias, int32_t relu, float* out, int64_t ldo, int32_t transpose_out,
                        const int32_t* row_map, float* out_rows, int64_t ld_rows, int64_t out_rows_n, const int32_t* out_rows_real,
                        const G3Riders& R, kgw_stream_t stream_) {
    if (!A || !packed || !workspace || !out) return KGW_E_NULL;
    if (out_rows_real && (!row_map || out_rows_n < 0 || out_rows_n > M)) return KGW_E_RANGE;   // (the padding is at most M rows)
    if (M <= 0 || K <= 0 || M > (1 << 30) || K > (1 << 30)) return KGW_E_RANGE;
    if (K % 32 || lda < 0 || (lda & 3) || ((uintptr_t)A & 15) || ((uintptr_t)packed & 15) || ((uintptr_t)workspace & 15)) return KGW_E_UNSUPPORTED;
    if (!transpose_out && ((ldo & 3) || ((uintptr_t)out & 15) || (bias && ((uintptr_t)bias & 15)))) return KGW_E_UNSUPPORTED;
    if (transpose_out && (bias || relu || row_map)) return KGW_E_UNSUPPORTED;
    if (row_map && (!out_rows || (ld_rows & 3) || ((uintptr_t)out_rows & 15))) return KGW_E_UNSUPPORTED;
    const int ns = g3_splits(M, K);
    if (workspace_floats < (int64_t)ns * M * 128) return KGW_E_RANGE;
    const int rt = 32 * g3_nw();
    const int tiles = (int)((M + rt - 1) / rt);
    hipStream_t st = (hipStream_t)stream_;
    const int per_xcd = (tiles * ns + 7) / 8;
    G3Args a{A, (long)lda, (int)M, (int)K, (const uint4*)packed, workspace, ns, tiles, per_xcd, g3_flip()};
    if (g3_nw() == 8) k_g3_gemm<1, 8><<<per_xcd * 8 + R.n_blocks, 512, 0, st>>>(G3KArgs{a, R});
    else k_g3_gemm<1, 4><<<per_xcd * 8 + R.n_blocks, 256, 0, st>>>(G3KArgs{a, R});
    KGW_LAUNCH_CHECK();
    if (transpose_out) k_g3_reduce_t<<<dim3((unsigned)((M + 31) / 32), 4), 256, 0, st>>>(workspace, ns, (long)M, out, (long)ldo);
    else k_g3_reduce<<<(int)((M * 32 + 255) / 256), 256, 0, st>>>(workspace, ns, (long)M, bias, relu, out, (long)ldo, row_map, out_rows, (long)ld_rows,
                                                                  (long)out_rows_n, out_rows_real);
    KGW_LAUNCH_CHECK();
    return KGW_OK;
}

extern "C" int kgw_gemm3(const float* A, int64_t lda, int64_t M, int64_t K, const void* packed, float* workspace,
                         int64_t workspace_floats, const float* bias, int32_t relu, float* out, int64_t ldo, int32_t transpose_out,
                         const int32_t* row_map, float* out_rows, int64_t ld_rows, int64_t out_rows_n, const int32_t* out_rows_real,
                         kgw_stream_t stream_) {
    const G3Riders none{};
    return gemm3_launch(A, lda, M, K, packed, workspace, workspace_floats, bias, relu, out, ldo, transpose_out, row_map, out_rows, ld_rows,
                        out_rows_n, out_rows_real, none, stream_);
}

extern "C" int kgw_gemm3_riders(const float* A, int64_t lda, int64_t M, int64_t K, const void* packed, float* workspace,
                                int64_t workspace_floats, const float* bias, int32_t relu, float* out, int64_t ldo, int32_t transpose_out,
                                const int32_t* row_map, float* out_rows, int64_t ld_rows, int64_t out_rows_n, const int32_t* out_rows_real,
                                int32_t n_relvec, const KgwRelvecJob* relvec, const KgwFoldArgs* fold, int32_t fold_job,
                                kgw_stream_t stream_) {
    const int nb = g3_free_cus(M, K);
    if (nb <= 0) return KGW_E_UNSUPPORTED;                  // (no idle compute unit: the caller launches the riders' own kernels)
    G3Riders R;
    int rc = g3_riders_build(n_relvec, relvec, fold, fold_job, nb, &R);
    if (rc) return rc;
    return gemm3_launch(A, lda, M, K, packed, workspace, workspace_floats, bias, relu, out, ldo, transpose_out, row_map, out_rows, ld_rows,
                        out_rows_n, out_rows_real, R, stream_);
}
