// kgw_common.h -- shared device helpers for libkgwas_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kgwas_hip.h"

#define KGW_PENDING (-2)

static constexpr int KGW_BLK  = 256;    // 4 wavefronts of 64
static constexpr int KGW_GRID = 2048;   // grid-stride launches: 8 blocks per CU on 256 CUs

#define KGW_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

#define KGW_HIP(x)                                           \
    do {                                                     \
        hipError_t e__ = (x);                                \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

// one-time set-up PER DEVICE (hipFuncSetAttribute is per device, and a process may drive more than one GPU): need() is true
// the first time it is called with a given device current.  A benign race at worst repeats the idempotent set-up.
struct KgwPerDevice {
    uint64_t done = 0;
    bool need() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return true;
        const uint64_t b = 1ull << (d & 63);
        if (done & b) return false;
        done |= b;
        return true;
    }
};

// ---- wave-level helpers (wavefront = 64 lanes) -------------------------------------------------
__device__ __forceinline__ int kgw_lane() { return threadIdx.x & 63; }

// DPP controls (GCN3+/CDNA): quad_perm = 0x00..0xFF, row_shr:n = 0x110+n, row_mirror = 0x140,
// row_half_mirror = 0x141.
template <int CTRL>
__device__ __forceinline__ float kgw_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// All-reduce (sum) inside each 32-lane half of the wavefront; every lane of a half gets its
// half's total.  4 DPP butterflies inside rows of 16 + one v_permlane16_swap across the rows.
__device__ __forceinline__ float kgw_half_allsum(float v) {
    v += kgw_dpp<0xB1>(v);    // quad_perm [1,0,3,2]  (xor 1)
    v += kgw_dpp<0x4E>(v);    // quad_perm [2,3,0,1]  (xor 2)
    v += kgw_dpp<0x141>(v);   // row_half_mirror      (xor 4 once quads are uniform)
    v += kgw_dpp<0x140>(v);   // row_mirror           (xor 8 once 8-groups are uniform)
    // rows 0<->1 and 2<->3: v_permlane16_swap_b32 exchanges the odd rows of its first operand with the
    // even rows of its second, in place.  Inline asm on purpose: hipcc (ROCm 7.2) folds the two results of
    // __builtin_amdgcn_permlane16_swap(x, x) into one register (emits v_add v, r0, r0) -- verified in the
    // ISA and by tests/test_gpu_primitives.py.  The two v_nop are the 2 wait states a VALU write needs
    // before a v_permlane*_swap reads it (guide T21).
    float a = v, b = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1\n\tv_nop" : "+v"(a), "+v"(b));
    return a + b;
}

// All-reduce (sum) across the whole wavefront.
__device__ __forceinline__ float kgw_wave_allsum(float v) {
    v = kgw_half_allsum(v);
    float a = v, b = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1\n\tv_nop" : "+v"(a), "+v"(b));
    return a + b;
}

// ---- eight dot products at once (the aggregate kernels' logits) -----------------------------------------------
// Every lane of a 32-lane half holds partial sums v[0..7] of EIGHT independent reductions.  A transposing butterfly
// halves the live values per step (xor 1, 2, 4) and two plain all-reduce steps (xor 8, 16) finish: the lane with
// (lane & 7) == p ends up with the half-wide total of reduction p -- 26 VALU ops instead of 8 x 9 for eight separate
// kgw_half_allsum calls.  xor 4 / xor 8 inside a row of 16 = a pair of bank-masked row_shl / row_shr DPP moves.
__device__ __forceinline__ float kgw_xor4(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xF, 0x5, false);       // row_shl:4 -> banks 0,2
    t = __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, v), 0x114, 0xF, 0xA, false);            // row_shr:4 -> banks 1,3
    return __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float kgw_xor8(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x108, 0xF, 0x3, false);       // row_shl:8 -> banks 0,1
    t = __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, v), 0x118, 0xF, 0xC, false);            // row_shr:8 -> banks 2,3
    return __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float kgw_xor16_sum(float v) {     // v + (value of lane ^ 16): rows swapped in place
    float a = v, b = v;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane16_swap_b32 %0, %1\n\tv_nop" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float kgw_half_reduce8(const float (&v)[8], int hl) {
    const bool b0 = hl & 1, b1 = hl & 2, b2 = hl & 4;
    float r1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float keep = b0 ? v[2 * k + 1] : v[2 * k], give = b0 ? v[2 * k] : v[2 * k + 1];
        r1[k] = keep + kgw_dpp<0xB1>(give);
    }
    float r2[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float keep = b1 ? r1[2 * k + 1] : r1[2 * k], give = b1 ? r1[2 * k] : r1[2 * k + 1];
        r2[k] = keep + kgw_dpp<0x4E>(give);
    }
    const float keep = b2 ? r2[1] : r2[0], give = b2 ? r2[0] : r2[1];
    float r = keep + kgw_xor4(give);
    r += kgw_xor8(r);
    return kgw_xor16_sum(r);
}
// all-reduce over the 8 residues (lanes differing in bits 0..2): every lane gets the max / sum of its 8-lane group
__device__ __forceinline__ float kgw_max8(float v) {
    v = fmaxf(v, kgw_dpp<0xB1>(v)); v = fmaxf(v, kgw_dpp<0x4E>(v)); return fmaxf(v, kgw_xor4(v));
}
__device__ __forceinline__ float kgw_sum8(float v) {
    v += kgw_dpp<0xB1>(v); v += kgw_dpp<0x4E>(v); return v + kgw_xor4(v);
}
// value held by lane (lane & ~7) | P of the same 8-lane group (ds_swizzle, bit mode: and 0x18, or P)
template <int P>
__device__ __forceinline__ float kgw_bcast8(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (P << 5) | 0x18));
}

__device__ __forceinline__ float kgw_xhalf(float v) {   // value held by the same lane of the other half
    return __shfl_xor(v, 32, 64);
}

// ---- exact three-way bf16 split of fp32 values (kgw_gemm3.hip, k_mlp2_fwd3) -----------------------------------------
// x = p1 + p2 + p3 with p1 = bf16(x), p2 = bf16(x - p1), p3 = x - p1 - p2: three 8-bit significands = the 24 bits of fp32, both
// residuals exact in fp32.  Eight values -> three registers-of-eight (element i in bits 16 (i & 1) of word i / 2), 44 VALU ops.
typedef __attribute__((ext_vector_type(8))) __bf16 kgw_bf8;
__device__ __forceinline__ float kgw_fxor(float x, unsigned m) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) ^ m); }
__device__ __forceinline__ uint32_t kgw_cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void kgw_split3x8(const float (&x)[8], uint4& p1, uint4& p2, uint4& p3) {
    uint32_t a[4], b[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        a[i] = kgw_cvt_pk_bf16(x0, x1);
        const float r0 = x0 - __builtin_bit_cast(float, a[i] << 16), r1 = x1 - __builtin_bit_cast(float, a[i] & 0xffff0000u);
        b[i] = kgw_cvt_pk_bf16(r0, r1);
        const float s0 = r0 - __builtin_bit_cast(float, b[i] << 16), s1 = r1 - __builtin_bit_cast(float, b[i] & 0xffff0000u);
        c[i] = kgw_cvt_pk_bf16(s0, s1);
    }
    p1 = make_uint4(a[0], a[1], a[2], a[3]);
    p2 = make_uint4(b[0], b[1], b[2], b[3]);
    p3 = make_uint4(c[0], c[1], c[2], c[3]);
}

// the eight level-1 pieces of a d u_r / d v_r value (KGW_F_DUV_PIECES: p[s * 128], s = 0..7), added in k_duv_fold's order
__device__ __forceinline__ float kgw_duv_sum8(const float* __restrict__ p) {
    return ((p[0] + p[128]) + (p[2 * 128] + p[3 * 128])) + ((p[4 * 128] + p[5 * 128]) + (p[6 * 128] + p[7 * 128]));
}

// block-wide exclusive scan of one int per thread (256 threads); returns exclusive prefix,
// *total receives the block sum.  sm must hold 256 ints.
__device__ __forceinline__ int kgw_block_exscan(int v, int* sm, int* total) {
    const int tid = threadIdx.x;
    sm[tid] = v;
    __syncthreads();
    for (int off = 1; off < KGW_BLK; off <<= 1) {
        int t = (tid >= off) ? sm[tid - off] : 0;
        __syncthreads();
        sm[tid] += t;
        __syncthreads();
    }
    int incl = sm[tid];
    *total = sm[KGW_BLK - 1];
    __syncthreads();
    return incl - v;
}
