// spin.hip -- diagnostic only (tools/side_queue_blocking.py): a kernel whose blocks do NOTHING but stay resident for a given time.
// It answers one question: what does a side queue cost the captured step through RESIDENCY alone -- wavefronts that hold registers
// on a compute unit and so keep a whole-CU workgroup of the step (k_g3_gemm, k_mlp2_fwd3: two 256-register wavefronts per SIMD)
// from starting there -- with no memory traffic, LDS or instruction issue of its own (s_sleep).
// build: hipcc -O2 --offload-arch=gfx950 -shared -fPIC tools/micro/spin.hip -o tools/micro/libspin.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_spin(long long ticks) {          // wall_clock64(): the 100 MHz constant clock
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int spin_launch(int blocks, int threads, long long ticks, void* stream) {
    hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, ticks);
    return (int)hipGetLastError();
}
