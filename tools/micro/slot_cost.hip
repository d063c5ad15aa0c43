// What does a filler cost beside a one-wave-per-SIMD MFMA stream?  256 blocks x 4 waves; every wave runs groups of 32
// v_mfma_f32_32x32x16_bf16 (16 accumulators) and a variant-specific set of fillers pinned between the MFMAs:
//   V=0 nothing | 1: 8 LDS-DMA pieces (buffer_load_dwordx4 ... lds) | 2: 8 buffer_load_dwordx4 -> VGPR | 3: 64 ds_read_b64_tr_b16
//   4: 32 ds_read_b128 | 5: 1+3 | 6: 11 LDS-DMA + 64 tr reads + 40 VALU (the backward's tile) | 7: 40 VALU | 8: 2+3
//   9: register staging: 8 loads -> VGPR, 8 ds_write_b128 of the previous group's registers, 64 tr reads, 40 VALU | 10: 8 ds_write_b128 only
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/slot_cost.hip -o tools/micro/slot_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int V, bool SHARED>
__global__ void __launch_bounds__(256, 1) k(const unsigned char* src, float* out, int groups) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[96 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += 256) ((unsigned*)lds)[i] = 0x3c003c00u + i;
    __syncthreads();
    unsigned x = 12345u + threadIdx.x * 2654435761u + blockIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { x = x * 1664525u + 1013904223u; a[j] = (__bf16)((float)(x >> 8) * (1.f / 16777216.f) - 0.5f);
                                  x = x * 1664525u + 1013904223u; b[j] = (__bf16)((float)(x >> 8) * (1.f / 16777216.f) - 0.5f); }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 16 << 20, 0x00020000);
    const unsigned voff = lane * 16;
    const unsigned ldsbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    const unsigned rd = ldsbase + lane * 16 + wave * 8192;
    constexpr bool dma = V == 1 || V == 5 || V == 6, ld = V == 2 || V == 8, tr = V == 3 || V == 5 || V == 6 || V == 8 || V == 9 || V == 11 || V == 13, b128 = V == 4, valu = V == 6 || V == 7 || V == 9;
    constexpr bool stage = V == 9 || V == 12 || V == 13, wr = V == 9 || V == 10 || V == 11 || V == 12;   // 11: writes + tr reads; 12: loads + writes; 13: loads (consumed late by VALU) + tr reads
    u32x4 held[8];
    for (int i = 0; i < 8; ++i) held[i] = u32x4{(unsigned)i, 1u, 2u, (unsigned)lane};
    const unsigned wrp = ldsbase + 65536 + wave * 8192 + lane * 16;
    u32x4 sink = {0, 0, 0, 0};
    float vs[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    for (int gq = 0; gq < groups; ++gq) {
        const unsigned soff = __builtin_amdgcn_readfirstlane(((gq * 37 + (SHARED ? 0 : blockIdx.x)) & 511) * 32768 + wave * 8192);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            acc[i & 15] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 15], 0, 0, 0);
            if (dma && (i < 8 || (V == 6 && i < 11)))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 65536 + wave * 8192 + (i & 7) * 1024), 16, voff, soff + (i & 7) * 1024, 0, 0);
            if (ld && i < 8) {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + i * 1024, 0);
                if (i == 7) { sink ^= v; }   // one consumer per group: the compiler waits for the newest only here
            }
            if (wr && i >= 8 && i < 16) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(wrp), "v"(held[i - 8]), "n"(0) : "memory");
            if (stage && i >= 16 && i < 24) held[i - 16] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + (i - 16) * 1024, 0);
            if (V == 13 && i == 15) { for (int j = 0; j < 8; ++j) sink ^= held[j]; }
            if (tr) {
                u32x2 r0, r1;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r0) : "v"(rd), "n"(0));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r1) : "v"(rd), "n"(4096));
                if (i == 31) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1)); sink[0] ^= r0[0] ^ r1[1]; }
            }
            if (b128) {
                u32x4 r0;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r0) : "v"(rd), "n"(0));
                if (i == 31) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0)); sink ^= r0; }
            }
            if (valu && i >= 12) {
                vs[i & 7] = __builtin_fmaf(vs[i & 7], 1.0001f, vs[(i + 1) & 7]);
                vs[(i + 3) & 7] = vs[(i + 3) & 7] * 0.999f + 0.5f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (dma) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += vs[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(sink[0] ^ sink[1] ^ sink[2] ^ sink[3]);
}
template <int V, bool SHARED = true> void run(const unsigned char* src, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int groups = 1024;
    float best = 1e9f, last = 0.f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<V, SHARED>), dim3(256), dim3(256), 0, 0, src, out, groups);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        last = ms;
    }
    printf("V=%d%s: best %.3f ms last %.3f ms = %.3f us per 32-MFMA group (last), %.0f TF/s\n", V, SHARED ? " (all blocks stream the same tiles)" : " (every block its own tiles)", best, last, last * 1e3 / groups,
           256.0 * 4 * groups * 32 * 2.0 * 32 * 32 * 16 / last * 1e-9);
}
int main() {
    unsigned char* src; float* out;
    (void)hipMalloc(&src, 16 << 20); (void)hipMemset(src, 1, 16 << 20); (void)hipMalloc(&out, 256 * 256 * 4);
    run<0>(src, out); run<0>(src, out);
    run<1>(src, out); run<1, false>(src, out); run<5>(src, out); run<6>(src, out); run<9>(src, out); run<9, false>(src, out); run<12>(src, out); run<13>(src, out); run<3>(src, out); run<10>(src, out); run<11>(src, out);
    run<0>(src, out);
    return 0;
}
