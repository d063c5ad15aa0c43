// One wave per SIMD with 256 accumulators, or two with 128 each?  The same per-CU work per "tile" -- 128 MFMAs (32x32x16 bf16), 44
// LDS-DMA pieces of 1 KiB, 8 KiB-per-wave A reads + the wave's share of 64 transpose reads, 224 VALU, one barrier -- and ~SALU
// scalar instructions PER WAVE (loop control and address arithmetic do not shrink with the wave's share), dealt over NW = 4 or 8
// waves.  Models fast_bwd_dsl_kernel's tile.  build: hipcc --offload-arch=gfx950 -O3 tools/micro/waves_per_simd.hip -o tools/micro/waves_per_simd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int NW, int SALU, bool DMA, bool READS>
__global__ void __launch_bounds__(64 * NW, NW / 4) k(const unsigned char* src, float* out, int groups) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[144 * 1024];
    constexpr int NM = 128 / NW;          // MFMAs per wave and tile
    constexpr int NACC = NM / 2;          // accumulators per wave (each used twice per tile: k-steps 0 and 1)
    constexpr int NDMA = 44 / NW;         // DMA pieces per wave and tile
    constexpr int NTR = 64 / NW;          // transpose reads per wave and tile
    constexpr int NVALU = 224 / NW;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = threadIdx.x; i < 144 * 1024 / 4; i += 64 * NW) ((unsigned*)lds)[i] = 0x3c003c00u + i * 2654435761u;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 16 << 20, 0x00020000);
    const unsigned voff = lane * 16;
    const unsigned ldsbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    const unsigned rd = ldsbase + lane * 16;
    float vs[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    unsigned sx = wave;
    u32x4 A[8];
    u32x2 B[8];
    for (int i = 0; i < 8; ++i) { A[i] = u32x4{0x3c003c00u + i, 0x3c013c02u, 0x3c033c04u, (unsigned)lane * 7u + 0x3c003c00u}; B[i] = u32x2{0x3c003c00u + i, 0x3c053c06u + lane}; }
    for (int gq = 0; gq < groups; ++gq) {
        const unsigned soff = __builtin_amdgcn_readfirstlane(((gq * 37) & 255) * 32768 + wave * (32768 / NW));
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, A[i & 7]);
            const bf16x8 b = __builtin_bit_cast(bf16x8, u32x4{B[i & 7][0], B[i & 7][1], B[(i + 3) & 7][0], B[(i + 5) & 7][1]});
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i % NACC]) : "v"(a), "v"(b));
            __builtin_amdgcn_sched_barrier(0);
            if (DMA && (i * NDMA) / NM != ((i + 1) * NDMA) / NM)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 98304 + wave * (32768 / NW) + ((i * NDMA) / NM) * 1024), 16, voff, soff + ((i * NDMA) / NM) * 1024, 0, 0);
            if (READS && i < 8) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A[i]) : "v"(rd), "n"(0));      // (all issued reads are waited for at the end of the tile)
            if (READS && (i * NTR) / NM != ((i + 1) * NTR) / NM) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(B[i & 7]) : "v"(rd), "n"(8192));
            for (int j = (i * NVALU) / NM; j < ((i + 1) * NVALU) / NM; ++j) vs[j & 7] = __builtin_fmaf(vs[j & 7], 1.0001f, vs[(j + 1) & 7]);
            for (int j = (i * SALU) / NM; j < ((i + 1) * SALU) / NM; ++j) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (READS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(A[i])); asm volatile("" : "+v"(B[i])); } }
        if (DMA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(44 / NW) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = (float)sx;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += vs[i];
    out[blockIdx.x * 64 * NW + threadIdx.x] = s;
}
template <int NW, int SALU, bool DMA, bool READS> void run(const unsigned char* src, float* out, const char* what) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int groups = 1024;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NW, SALU, DMA, READS>), dim3(256), dim3(64 * NW), 0, 0, src, out, groups);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    printf("NW=%d SALU/wave=%3d DMA=%d READS=%d %-28s: %.3f us per 128-MFMA tile, %.0f TF/s\n", NW, SALU, (int)DMA, (int)READS, what, best * 1e3 / groups,
           256.0 * groups * 128 * 2.0 * 32 * 32 * 16 / best * 1e-9);
}
int main() {
    unsigned char* src; float* out;
    (void)hipMalloc(&src, 16 << 20); (void)hipMemset(src, 0x3c, 16 << 20); (void)hipMalloc(&out, 256 * 512 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<4, 0, false, false>(src, out, "bare MFMA + barrier");
        run<8, 0, false, false>(src, out, "bare MFMA + barrier");
        run<4, 70, true, true>(src, out, "whole tile");
        run<8, 70, true, true>(src, out, "whole tile");
        run<8, 50, true, true>(src, out, "whole tile");
        run<4, 70, false, true>(src, out, "no DMA");
        run<8, 70, false, true>(src, out, "no DMA");
        run<4, 0, true, true>(src, out, "no SALU");
        run<8, 0, true, true>(src, out, "no SALU");
        run<4, 35, true, true>(src, out, "half SALU");
    }
    return 0;
}
