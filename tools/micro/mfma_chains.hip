// How many independent accumulator chains does v_mfma_f32_32x32x16_bf16 need to keep the matrix pipe busy?  256 blocks x 4 waves
// (one per SIMD), NACC accumulators used round-robin (NACC = 1: every MFMA depends on the one before), nothing else in the loop.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chains.hip -o tools/micro/mfma_chains ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
template <int NACC, bool ZERO>
__global__ void __launch_bounds__(256, 1) k(float* out, int iters, unsigned seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { x = x * 1664525u + 1013904223u; a[j] = ZERO ? (__bf16)0.f : (__bf16)((float)(x >> 8) * (1.f / 16777216.f) - 0.5f);
                                  x = x * 1664525u + 1013904223u; b[j] = ZERO ? (__bf16)0.f : (__bf16)((float)(x >> 8) * (1.f / 16777216.f) - 0.5f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % NACC], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, bool ZERO> void run(float* out, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, last = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
        const int iters = 1024;
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, ZERO>), dim3(256), dim3(256), 0, 0, out, iters, 12345u + rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        last = ms;
    }
    const double flop = 256.0 * 4 * 1024 * 16 * 2.0 * 32 * 32 * 16;
    printf("%-12s chains %2d: best %.3f ms = %6.1f TF/s (%.1f cycles/MFMA at 2.4 GHz); settled %.3f ms = %6.1f TF/s\n", what, NACC, best,
           flop / best * 1e-9, best * 1e-3 * 2.4e9 / (1024 * 16.0), last, flop / last * 1e-9);
}
int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    run<16, false>(out, "(settle)"); run<16, false>(out, "(settle)");
    run<1, false>(out, "random"); run<2, false>(out, "random"); run<4, false>(out, "random"); run<8, false>(out, "random"); run<16, false>(out, "random");
    run<2, true>(out, "zeros"); run<16, true>(out, "zeros");
    return 0;
}
