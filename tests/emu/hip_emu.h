// hip_emu.h -- TEST INFRASTRUCTURE.  A minimal lane-level emulation of the HIP/gfx950 features the
// CrossCLR kernels use, so the *same kernel source* can be executed on CPU threads and checked
// against the oracle before any GPU time is spent.  One OS thread per lane; __syncthreads and
// wave-collective operations (MFMA, shuffles, transpose reads) are pthread barriers plus a per-wave
// exchange area.  Blocks run one after another, so `__shared__` is plain static storage.
//
// The MFMA and ds_read_b64_tr_b16 models follow the layouts documented for gfx950; whether the
// hardware agrees is checked ON the hardware by tests/test_hw_assumptions.py (crossclr_selftest).
// Nothing in the product package includes this file.
#pragma once
#include <pthread.h>
#include <sched.h>
#include <atomic>
#include <stdarg.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <thread>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define CROSSCLR_SHARED static

namespace emu {

// Barrier for many more threads than cores: arrive on an atomic counter, then yield until the generation flips.
// (pthread_barrier_t parks every waiter on a futex; with 64-512 threads per block on a handful of cores the wake-ups
// dominate the emulation time.)
struct YieldBarrier {
    std::atomic<int> count{0};
    std::atomic<int> gen{0};
    int n = 1;
    void init(int nthreads) { n = nthreads; count.store(0); gen.store(0); }
    void wait() {
        const int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
            count.store(0, std::memory_order_relaxed);
            gen.store(g + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (gen.load(std::memory_order_acquire) == g) {
                if (++spins < 64) continue;
                sched_yield();
            }
        }
    }
};

struct WaveCtx {
    YieldBarrier bar;
    alignas(64) unsigned char slot[64][64];  // per-lane exchange area
};
struct BlockCtx {
    YieldBarrier bar;
    std::vector<WaveCtx*> waves;
    int nthreads;
};

extern thread_local dim3 t_threadIdx;
extern thread_local dim3 t_blockIdx;
extern thread_local BlockCtx* t_block;
extern thread_local int t_tid;
extern dim3 g_blockDim, g_gridDim;

inline WaveCtx& my_wave() { return *t_block->waves[t_tid >> 6]; }
inline int my_lane() { return t_tid & 63; }
inline void wave_sync() { my_wave().bar.wait(); }

template <typename K, typename... Args>
void launch(K kernel, dim3 grid, dim3 block, Args... args) {
    const int nthreads = (int)(block.x * block.y * block.z);
    const int nwaves = (nthreads + 63) / 64;
    g_blockDim = block;
    g_gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                BlockCtx ctx;
                ctx.nthreads = nthreads;
                ctx.bar.init(nthreads);
                for (int w = 0; w < nwaves; ++w) {
                    WaveCtx* wc = new WaveCtx;
                    int lanes = nthreads - w * 64 < 64 ? nthreads - w * 64 : 64;
                    wc->bar.init(lanes);
                    ctx.waves.push_back(wc);
                }
                std::vector<std::thread> th;
                th.reserve(nthreads);
                for (int t = 0; t < nthreads; ++t) {
                    th.emplace_back([&, t]() {
                        t_tid = t;
                        t_block = &ctx;
                        t_threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                        t_blockIdx = dim3(bx, by, bz);
                        kernel(args...);
                    });
                }
                for (auto& x : th) x.join();
                for (auto* wc : ctx.waves) delete wc;
            }
}

}  // namespace emu

#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

inline void __syncthreads() { emu::t_block->bar.wait(); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
// crossclr_device.h, handoff_*: values handed from block to block inside a launch (the emulated blocks are host threads: sequentially consistent atomics)
inline void handoff_store_f64(double* p, double v) { __atomic_store(p, &v, __ATOMIC_SEQ_CST); }
inline void handoff_stores_complete() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline double handoff_load_f64(const double* p) { double v; __atomic_load(const_cast<double*>(p), &v, __ATOMIC_SEQ_CST); return v; }

namespace crossclr {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// exchange helper: every lane publishes `v`, then reads lane `src`'s value
template <typename T> inline T wave_read_lane(T v, int src) {
    static_assert(sizeof(T) <= 64, "exchange slot too small");
    emu::WaveCtx& w = emu::my_wave();
    memcpy(w.slot[emu::my_lane()], &v, sizeof(T));
    emu::wave_sync();
    T r;
    memcpy(&r, w.slot[src], sizeof(T));
    emu::wave_sync();
    return r;
}
inline float wave_xor_f32(float v, int mask) { return wave_read_lane(v, emu::my_lane() ^ mask); }
inline double wave_xor_f64(double v, int mask) { return wave_read_lane(v, emu::my_lane() ^ mask); }
inline float fast_exp2(float x) { return exp2f(x); }
template <int MASK> inline float lane_xor(float v) { return wave_read_lane(v, emu::my_lane() ^ MASK); }

// v_mfma_f32_32x32x16_bf16: A[i][k] in lane i + 32*(k/8), element k%8; B[k][j] in lane j + 32*(k/8),
// element k%8; C[i][j] in lane j + 32*((i/4)%2), register (i%4) + 4*(i/8).
inline f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    emu::WaveCtx& w = emu::my_wave();
    const int lane = emu::my_lane();
    struct AB { bf16x8 a, b; };
    AB mine{a, b};
    memcpy(w.slot[lane], &mine, sizeof(mine));
    emu::wave_sync();
    const int j = lane & 31, half = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        float s = c[r];
        for (int k = 0; k < 16; ++k) {
            AB la, lb;
            memcpy(&la, w.slot[i + 32 * (k >> 3)], sizeof(AB));
            memcpy(&lb, w.slot[j + 32 * (k >> 3)], sizeof(AB));
            s += (float)la.a[k & 7] * (float)lb.b[k & 7];
        }
        c[r] = s;
    }
    emu::wave_sync();
    return c;
}
// v_mfma_f32_16x16x32_bf16: A[i][k] in lane i + 16*(k/8), element k%8; B[k][j] in lane j + 16*(k/8), element k%8;
// C[i][j] in lane j + 16*(i/4), register i%4.
typedef __attribute__((ext_vector_type(4))) float f32x4_emu;
inline f32x4_emu mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4_emu c) {
    emu::WaveCtx& w = emu::my_wave();
    const int lane = emu::my_lane();
    struct AB { bf16x8 a, b; };
    AB mine{a, b};
    memcpy(w.slot[lane], &mine, sizeof(mine));
    emu::wave_sync();
    const int j = lane & 15, g4 = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g4 + r;
        float s = c[r];
        for (int k = 0; k < 32; ++k) {
            AB la, lb;
            memcpy(&la, w.slot[i + 16 * (k >> 3)], sizeof(AB));
            memcpy(&lb, w.slot[j + 16 * (k >> 3)], sizeof(AB));
            s += (float)la.a[k & 7] * (float)lb.b[k & 7];
        }
        c[r] = s;
    }
    emu::wave_sync();
    return c;
}
// v_mfma_f32_32x32x2_f32: A[i][k] in lane i + 32k, B[k][j] in lane j + 32k (k = 0, 1)
inline f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
    emu::WaveCtx& w = emu::my_wave();
    const int lane = emu::my_lane();
    float mine[2] = {a, b};
    memcpy(w.slot[lane], mine, sizeof(mine));
    emu::wave_sync();
    const int j = lane & 31, half = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        float s = c[r];
        for (int k = 0; k < 2; ++k) {
            float la[2], lb[2];
            memcpy(la, w.slot[i + 32 * k], sizeof(la));
            memcpy(lb, w.slot[j + 32 * k], sizeof(lb));
            s = fmaf(la[0], lb[1], s);
        }
        c[r] = s;
    }
    emu::wave_sync();
    return c;
}
// ds_read_b64_tr_b16: within each 16-lane group, lane i receives, for j = 0..3, element (i & 3) of
// the 8-byte piece addressed by lane 4j + (i >> 2) of the same group.
inline s16x4 lds_read_tr16_b64(const void* p) {
    emu::WaveCtx& w = emu::my_wave();
    const int lane = emu::my_lane();
    memcpy(w.slot[lane], p, 8);
    emu::wave_sync();
    const int base = lane & ~15, i = lane & 15;
    s16x4 r;
    for (int j = 0; j < 4; ++j) {
        short piece[4];
        memcpy(piece, w.slot[base + 4 * j + (i >> 2)], 8);
        r[j] = piece[i & 3];
    }
    emu::wave_sync();
    return r;
}

// LDS-DMA (global_load_lds): lane l's bytes land at the wave-uniform base + l*size
inline void lds_dma16(const void* gsrc, void* lds_wave_base) {
    memcpy(static_cast<unsigned char*>(lds_wave_base) + 16 * emu::my_lane(), gsrc, 16);
}
inline void lds_dma4(const void* gsrc, void* lds_wave_base) {
    memcpy(static_cast<unsigned char*>(lds_wave_base) + 4 * emu::my_lane(), gsrc, 4);
}
// a wave's DMA is complete for the whole wave once every lane has passed this point
inline void wait_dma() { emu::wave_sync(); }
template <int N> inline void wait_dma_keep() { emu::wave_sync(); }
inline int uniform(int x) { return x; }

}  // namespace crossclr
