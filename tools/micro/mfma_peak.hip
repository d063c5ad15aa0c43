// Sustained v_mfma_f32_32x32x16_bf16 rate of this GPU (what "MFMA-bound" means in wall-clock terms for the backward):
// 256 blocks x 4 waves (one per SIMD), 16 independent 32x32 accumulators per wave, N back-to-back MFMAs.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
__global__ void __launch_bounds__(256, 1) k(float* out, int iters, unsigned seed) {
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { x = x * 1664525u + 1013904223u; a[j] = (__bf16)((float)(x >> 8) * (1.f / 16777216.f) - 0.5f);
                                  x = x * 1664525u + 1013904223u; b[j] = (__bf16)((float)(x >> 8) * (1.f / 16777216.f) - 0.5f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        const int iters = 2048;   // 32768 MFMAs per wave
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, iters, 12345u + rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = 256.0 * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
        double cyc_per_mfma_at_2p4 = ms * 1e-3 * 2.4e9 / (iters * 16.0);
        printf("rep %d: %.3f ms  %.1f TF/s  (%.1f cycles per MFMA if the clock were 2.4 GHz)\n", rep, ms, flop / ms * 1e-9, cyc_per_mfma_at_2p4);
    }
    return 0;
}
