// Issue rate of v_mfma_f32_32x32x16_bf16 on gfx950: cycles per MFMA per SIMD for NACC independent accumulators in rotation,
// W wavefronts per SIMD (timing experiment behind kgwas_amd/csrc/kgw_gemm3.hip).   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int NACC>
__global__ void __launch_bounds__(256) k(const u4* src, float* out, long long* cyc, int iters) {
    bf8 a = __builtin_bit_cast(bf8, src[threadIdx.x]), b = __builtin_bit_cast(bf8, src[256 + threadIdx.x]);
    f16v acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(const u4* src, float* out, long long* cyc, int blocks, const char* tag) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(src, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(src, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n_mfma = (double)iters * 6 * NACC;                     // per wavefront
    const double waves_per_simd = blocks / 256.0;
    printf("%s nacc %d blocks %d: %.1f us, %.1f ns per MFMA per SIMD, s_memtime ticks per MFMA of one wave %.1f, TF %.0f\n", tag, NACC, blocks,
           ms * 1e3, ms * 1e6 / (n_mfma * waves_per_simd), (double)c / n_mfma, blocks * 4.0 * n_mfma * 32768.0 / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    u4* src; float* out; long long* cyc;
    hipMalloc(&src, 512 * 16); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    unsigned h[2048];
    for (int z = 0; z < 2; ++z) {
        for (int i = 0; i < 2048; ++i) h[i] = z ? 0u : (unsigned)(i * 2654435761u) & 0x3f803f80u | 0x3c003c00u;   // bf16 pairs around 0.01..1
        hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
        const char* tag = z ? "zeros " : "random";
        run<4>(src, out, cyc, 256, tag); run<4>(src, out, cyc, 512, tag);
        run<8>(src, out, cyc, 256, tag); run<8>(src, out, cyc, 512, tag);
        run<2>(src, out, cyc, 512, tag);
    }
    return 0;
}
