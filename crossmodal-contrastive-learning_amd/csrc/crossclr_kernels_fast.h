// crossclr_kernels_fast.h -- register-resident bf16 kernels (the BASELINE headline path).
//
// Shape of both kernels (flash-attention-like, nothing O(B^2) ever leaves the CU):
//   * a wavefront owns 32 rows p of the batch (16 in fast_bwd16_kernel) and keeps their normalised embeddings in
//     VGPRs as MFMA B-fragments for the whole kernel (Dpad/4 VGPRs: 128 at D=512);
//   * 32-column tiles of the column operand stream through a 4-deep (2-deep for 48/64 KiB tiles) LDS ring filled
//     by LDS-DMA (global_load_lds, 16 B/lane, wave-uniform base + 32-bit lane offset), one barrier per tile, counted
//     s_waitcnt vmcnt(N) so that up to three tiles stay in flight during compute;
//   * S^T = Xq . Xp^T is computed with SWAPPED operands so lane (l&31) owns row p and 16 columns:
//     row-wise soft-max sums need no cross-lane traffic, and in the backward the 32x32 fragment,
//     turned into W = s E (1/Z_p + 1/Z_q) and packed to bf16, IS the A operand of the second MFMA
//     (k <-> q permuted consistently on both operands), so W never touches LDS;
//   * the second MFMA contracts over q, the ROW index of the LDS tile: its B fragments come from
//     ds_read_b64_tr_b16 (16-lane-group transpose read) of the same tile -- one tile, two uses.
// LDS tile layout: row q of the tile at q*RB, 16-byte chunk c at slot (c & ~15) | ((c ^ sigma(q)) & 15),
// sigma(q) = ((q&3)<<2) | ((q>>2)&3): conflict-free for BOTH the ds_read_b128 column reads of the
// first MFMA (16 rows distinct mod 16 -> 16 distinct slots) and the transpose reads of the second
// (4 consecutive rows -> 4 different 64-byte bank groups).  The DMA writes LDS lane-linearly, so the
// permutation is applied to the per-lane GLOBAL source address (same 16-chunk group of the same row:
// coalescing is unaffected).
#pragma once
#include "crossclr_device.h"
#include "../../include/crossclr.h"
#include <stdlib.h>
#include <string.h>
#include <utility>

namespace crossclr {

#ifndef CROSSCLR_EMU
// 64 lanes x 16 B (or 4 B) from per-lane global addresses to LDS at wave-uniform base + lane*size
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void lds_dma4(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most N of this wave's most recent VMEM operations are still in flight
template <int N> __device__ __forceinline__ void wait_dma_keep() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
// LDS byte address of a generic pointer into __shared__ memory
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
// Transpose read issued from inline asm: hipcc's s_waitcnt pass treats the ds_read_tr BUILTIN as "may alias every LDS-DMA
// in flight" and puts s_waitcnt vmcnt(0) in front of it, which drains the whole DMA ring in the middle of every tile.
// The asm form is invisible to that pass; its completion is counted by hand: wait_lgkm<N>(x) makes x usable once at most
// N younger LDS operations of this wave are still in flight (LDS returns in order), and -- naming x as read-write --
// keeps every consumer of x below the wait.
template <int OFF> __device__ __forceinline__ s16x4 lds_read_tr16_b64_async(unsigned addr) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int N> __device__ __forceinline__ void wait_lgkm(s16x4& a, s16x4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int OFF> __device__ __forceinline__ u32x4 lds_read_b128_async(unsigned addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// counted LDS wait + the MFMA that consumes the fragment, ONE asm statement: no compiler pad (s_nop) between an asm wait and
// a builtin MFMA, and the fragment registers are only ever read after the wait.  acc lives in AGPRs ("+a"); A comes from
// VALU results written many instructions earlier, B from the LDS reads being waited for; D -> next reader is another MFMA
// on the same accumulator or the epilogue far behind the loop, so no software wait states are needed inside the string.
template <int N> __device__ __forceinline__ void mfma_after_lgkm(f32x16& acc, bf16x8 a, bf16x8 b) {
    asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b), "n"(N));
}
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4& a, u32x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
// workgroup barrier that does NOT drain VMEM (an LDS-DMA ring stays in flight across it): LDS operations only
__device__ __forceinline__ void barrier_keep_dma() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Buffer-addressed LDS-DMA (buffer_load_dword[x4] ... offen lds): wave-uniform 128-bit descriptor + per-lane 32-bit byte offset +
// scalar byte offset.  Per tile only the scalar offset changes: no 64-bit per-lane address arithmetic, and reads past
// `bytes` return zeros instead of faulting.
struct BufRsrc { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ BufRsrc make_rsrc(const void* base, unsigned bytes) {
    BufRsrc b;
    b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
    return b;
}
__device__ __forceinline__ void lds_dma16_buf(const BufRsrc& b, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
__device__ __forceinline__ void lds_dma4_buf(const BufRsrc& b, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_wave_base, 4, (int)voff, (int)soff, 0, 0);
}
// (The only user is the forward's stash of saved exponentials -- 0.27 GB written once and read by the backward much later.  Round 3: aux = 2
//  (non-temporal), -2 %.  Round 4: sc1 | nt -- the line is DROPPED from the XCD's L2 once written (MI355X_MICROARCH.md, stores of each
//  flavour): with plain or nt stores every XCD pushes 0.5 MB of stash lines through its 4-MiB L2 per tile interval and a column tile
//  survives ~8 tiles; with sc1 the L2 belongs to the column tiles, which is what makes the XCD-aware range placement (fwd_make_perm) work:
//  FETCH_SIZE 224 -> 71 MB per launch, forward_save -7 % (profiles/r04_ab_fwd_xcd*.txt; sc1 alone: the same counters, ~1 % slower).
//  The nt hint on the backward's stash LOADS costs +3.7 %: every tile is read twice.)
#ifndef CROSSCLR_STASH_AUX
#define CROSSCLR_STASH_AUX 18      // cache policy of the stash stores: 2 = nt, 16 = sc1, 18 = sc1 | nt (A/B: tools/ab_fwd_xcd.sh)
#endif
__device__ __forceinline__ void buf_store16(const BufRsrc& b, unsigned voff, unsigned soff, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, b.r, (int)voff, (int)soff, CROSSCLR_STASH_AUX);
}
__device__ __forceinline__ void buf_store4(const BufRsrc& b, unsigned voff, unsigned soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b.r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ float buf_load4(const BufRsrc& b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// s_waitcnt vmcnt(0) as a BUILTIN: unlike the asm form hipcc's wait-count pass sees it and marks its own pending global
// loads complete -- otherwise it re-waits for them (vmcnt(N) countdowns that also drain the DMA ring) at every later use
__device__ __forceinline__ void wait_loads_visible() { __builtin_amdgcn_s_waitcnt(0x0F70); }
#else
__device__ __forceinline__ unsigned long lds_addr(const void* p) { return (unsigned long)(uintptr_t)p; }
template <int OFF> __device__ __forceinline__ s16x4 lds_read_tr16_b64_async(unsigned long addr) {
    return lds_read_tr16_b64(reinterpret_cast<const unsigned char*>(addr) + OFF);
}
template <int N> __device__ __forceinline__ void wait_lgkm(s16x4&, s16x4&) {}
template <int OFF> __device__ __forceinline__ u32x4 lds_read_b128_async(unsigned long addr) {
    return *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(addr) + OFF);
}
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4&) {}
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4&, u32x4&) {}
template <int N> __device__ __forceinline__ void mfma_after_lgkm(f32x16& acc, bf16x8 a, bf16x8 b) { acc = mfma_32x32x16_bf16(a, b, acc); }
__device__ __forceinline__ void barrier_keep_dma() { __syncthreads(); }
struct BufRsrc { const unsigned char* base; unsigned bytes; };
__device__ __forceinline__ BufRsrc make_rsrc(const void* base, unsigned bytes) { return BufRsrc{static_cast<const unsigned char*>(base), bytes}; }
__device__ __forceinline__ void lds_dma16_buf(const BufRsrc& b, unsigned voff, unsigned soff, void* lds_wave_base) {
    unsigned char z[16] = {0};
    const size_t o = (size_t)voff + soff;
    memcpy(static_cast<unsigned char*>(lds_wave_base) + 16 * emu::my_lane(), o + 16 <= b.bytes ? b.base + o : z, 16);
}
__device__ __forceinline__ void lds_dma4_buf(const BufRsrc& b, unsigned voff, unsigned soff, void* lds_wave_base) {
    unsigned char z[4] = {0};
    const size_t o = (size_t)voff + soff;
    memcpy(static_cast<unsigned char*>(lds_wave_base) + 4 * emu::my_lane(), o + 4 <= b.bytes ? b.base + o : z, 4);
}
__device__ __forceinline__ void buf_store16(const BufRsrc& b, unsigned voff, unsigned soff, u32x4 v) {
    memcpy(const_cast<unsigned char*>(b.base) + voff + soff, &v, 16);
}
__device__ __forceinline__ void buf_store4(const BufRsrc& b, unsigned voff, unsigned soff, float v) {
    if ((size_t)voff + soff + 4 <= b.bytes) memcpy(const_cast<unsigned char*>(b.base) + voff + soff, &v, 4);
}
__device__ __forceinline__ float buf_load4(const BufRsrc& b, unsigned voff, unsigned soff) {
    float v = 0.f;
    if ((size_t)voff + soff + 4 <= b.bytes) memcpy(&v, b.base + voff + soff, 4);
    return v;
}
__device__ __forceinline__ void sched_fence() {}
__device__ __forceinline__ void wait_loads_visible() {}
#endif

// compile-time loops whose index is a constant inside the body (asm immediates, register-array indices)
template <int I> struct IdxC { static constexpr int value = I; };
template <int... Is, typename F> __device__ __forceinline__ void static_for_seq(std::integer_sequence<int, Is...>, F&& f) {
    (f(IdxC<Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_seq(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// Tuning switches (compile-time; defaults = best measured on MI355X, see profiles/):
//   CROSSCLR_PF     LDS fragment reads kept in flight ahead of the MFMA that consumes them
//   CROSSCLR_SCHED  0 none, 1 iglp_opt(0), 2 iglp_opt(1), 3 explicit sched_group_barrier pipeline
//                   {PF x reads ; (1 MFMA ; reads)*}: without it hipcc re-places the reads right in
//                   front of their MFMA (lgkmcnt(0) before every MFMA)
//   CROSSCLR_ABL    timing ablations of the fast backward (results are WRONG; tuning only):
//                   bit0 skip second product, bit1 skip exp/weights, bit2 skip first product, bit3 skip DMA+barriers,
//                   bit4 second product without its LDS reads, bit5 first product without its LDS reads
#ifndef CROSSCLR_PF
#define CROSSCLR_PF 4
#endif
#ifndef CROSSCLR_SCHED
#define CROSSCLR_SCHED 3
#endif
#ifndef CROSSCLR_ABL
#define CROSSCLR_ABL 0
#endif
#ifndef CROSSCLR_G1CHAINS
#define CROSSCLR_G1CHAINS 1   // accumulator chains of the 32-row backward's first product (2: measured)
#endif
#if defined(CROSSCLR_EMU) || CROSSCLR_SCHED == 0
#define SCHED_PIPELINE(nmfma, reads_per_mfma, pf) do {} while (0)
#elif CROSSCLR_SCHED == 1
#define SCHED_PIPELINE(nmfma, reads_per_mfma, pf) __builtin_amdgcn_iglp_opt(0)
#elif CROSSCLR_SCHED == 2
#define SCHED_PIPELINE(nmfma, reads_per_mfma, pf) __builtin_amdgcn_iglp_opt(1)
#else
#define SCHED_PIPELINE(nmfma, reads_per_mfma, pf)                                                      \
    do {                                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, (reads_per_mfma) * (pf), 0);                       \
        _Pragma("unroll") for (int _i = 0; _i < (nmfma); ++_i) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
            if (_i + (pf) < (nmfma)) __builtin_amdgcn_sched_group_barrier(0x100, (reads_per_mfma), 0); \
        }                                                                                              \
    } while (0)
#endif

#ifndef CROSSCLR_TUNE
#define CROSSCLR_TUNE 0   // A/B switches: bit0 (unused), bit1 no per-MFMA fence in fast_bwd16_kernel, bit2 draining barrier there
#endif
#ifndef CROSSCLR_FWD_PF
#define CROSSCLR_FWD_PF 2
#endif
#ifndef CROSSCLR_FWD_SCHED
#define CROSSCLR_FWD_SCHED 1
#endif
// forward: per k-step 2 reads feed 2 MFMAs
#if defined(CROSSCLR_EMU) || CROSSCLR_FWD_SCHED == 0
#define SCHED_PIPELINE_FWD(nsteps, pf) do {} while (0)
#else
#define SCHED_PIPELINE_FWD(nsteps, pf)                                                       \
    do {                                                                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (pf), 0);                            \
        _Pragma("unroll") for (int _i = 0; _i < (nsteps); ++_i) {                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                               \
            if (_i + (pf) < (nsteps)) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     \
        }                                                                                    \
    } while (0)
#endif

static inline int fast_dpad(int D) {
    if (D <= 128) return 128;
    if (D <= 256) return 256;
    if (D <= 384) return 384;
    if (D <= 512) return 512;
    if (D <= 768) return 768;
    if (D <= 1024) return 1024;
    return 0;
}

// bf16 plans beyond the register-resident forward (D > 1024): the generic tiled forward saves its exponentials in the register-resident
// layout and the D-slice saved backward runs as XP column parts of 384 / 512 columns (blockIdx.z) -- Dpad = XP x part, XP = 3 ... 16 (D <= 8192)
static inline int wide_bf16_dpad(int D) {
    if (D <= 1024) return 0;
    if (D <= 1152) return 1152;     // 3 x 384
    if (D <= 1536) return 1536;     // 3 x 512
    if (D <= 2048) return 2048;     // 4 x 512
    if (D <= 2560) return 2560;     // 5 x 512
    if (D <= 3072) return 3072;     // 6 x 512
    if (D <= 4096) return 4096;     // 8 x 512
    if (D <= 5120) return 5120;     // 10 x 512   (round 6: up to 16 parts -- the part count only sets the row pitch of the operand and the gradient)
    if (D <= 6144) return 6144;     // 12 x 512
    if (D <= 8192) return 8192;     // 16 x 512
    return 0;
}

__device__ __forceinline__ int sigma16(int q) { return ((q & 3) << 2) | ((q >> 2) & 3); }
__device__ __forceinline__ int swz_slot(int chunk, int q) { return (chunk & ~15) | ((chunk ^ sigma16(q)) & 15); }

// Issue this wave's share of the LDS-DMA for one 64-row column tile (+ optionally its 64 per-column
// statistics).  NW waves cooperate; wave-instruction ii fills LDS bytes [ii*1024, ii*1024+1024).
// tile image of the 16-row backward: sigma(q) = (q & 7) << 1
__device__ __forceinline__ int swz_slot16r(int chunk, int q) { return (chunk & ~15) | ((chunk ^ ((q & 7) << 1)) & 15); }

template <int RB, int NW, int QT, int SWZ = 0>
__device__ __forceinline__ void issue_tile_dma(const unsigned char* tile_src, unsigned char* buf, int wave, int lane,
                                               const float* stat_src, unsigned char* stat_dst,
                                               const float* stat2_src = nullptr, unsigned char* stat2_dst = nullptr) {
    constexpr int kInstr = QT * RB / 1024;  // wave-instructions per tile
#pragma unroll
    for (int k = 0; k < (kInstr + NW - 1) / NW; ++k) {
        const int ii = wave + NW * k;
        if (kInstr % NW == 0 || ii < kInstr) {
            const int L = ii * 1024 + lane * 16;
            const int row = L / RB, slot = (L - row * RB) >> 4;
            // wave-uniform 64-bit base + per-lane UNSIGNED 32-bit offset: selects the SGPR-base addressing form
            // (one offset VGPR per DMA instead of 64-bit address pairs -- the backward has no registers to spare)
            const unsigned off = (unsigned)(row * RB + ((SWZ ? swz_slot16r(slot, row) : swz_slot(slot, row)) << 4));
            lds_dma16(tile_src + off, buf + ii * 1024);
        }
    }
    // every wave issues the (identical) statistics DMA so that all waves have the same VMEM count per tile:
    // the counted s_waitcnt vmcnt(N) in the backward relies on it
    if (stat_src != nullptr && lane < QT) lds_dma4(stat_src + (unsigned)lane, stat_dst);
    if (stat2_src != nullptr && lane < QT) lds_dma4(stat2_src + (unsigned)lane, stat2_dst);
}


// ---------------------------------------------------------------------------------------------
// Work list of the persistent forward.  Row block I = 32*tpr rows of the stacked operand (tpr = 8: 256-row blocks,
// Dpad <= 512; tpr = 4: 128-row blocks, Dpad <= 1024); column tile t = 32 columns.
// kind 1 (symmetric: rows and columns are the same operand): row block I owns tiles tpr*I .. NT-1;  kind 2 (rectangular): every row block owns all NC column tiles that are not in the
// skipped rank.  The (I, t) pairs, row-block major, form a flat list of `total` items; the list is cut into `nblk` contiguous
// ranges of equal COST (`per` units each: fwdw_* in crossclr_device.h) -- one persistent thread block each.
// A row block's partial row sums land in slot (block - first block touching that row block).
// ---------------------------------------------------------------------------------------------
// (FwdWork, fwd_make_work, fwd_block_begin, fwd_max_slots: crossclr_device.h -- pure integer code, unit-tested on the host)

// ---------------------------------------------------------------------------------------------
// The persistent forward itself is fast_fwd_pipe_kernel (crossclr_kernels_sym.h); what it shares with the backward kernels
// of this file lives here: the work list above and the layout of the saved exponentials.
//  ST (save for backward): every evaluated 32x32 tile of exponentials is also written to `stash` as bf16, in the layout of
//    the MFMA A fragment the backward feeds on (lane = row p, 8 k-slots per 16-column half: the C layout of the forward's
//    product, so the store is two coalesced 1-KiB wave stores and costs 8 v_cvt_pk per tile).  Symmetric launch: tile
//    (r32, t) -- 32-row group r32, 32-column tile t >= tpr * (r32 / tpr) -- lives at stash_tile_index(...) * 2 KiB
//    (tpr = 32-row groups per row block of the forward: 8 for Dpad <= 512, 4 above).
//    fast_bwd_dsl_kernel turns it into W = E (1/Z_p + 1/Z_q) without recomputing the similarity product.
// ---------------------------------------------------------------------------------------------
// (stash_tile_index / stash_tiles_total: crossclr_device.h -- the generic forward of wide bf16 plans fills the same layout)

// ---------------------------------------------------------------------------------------------
// backward: 4 waves x 32 rows per block, ONE wave per SIMD so each wave owns the whole 512-entry
// register file: 32 x Dpad fp32 gradient accumulators (256 at D=512) + Dpad/4 operand VGPRs.
// Column tiles are 32 rows (32 KiB at D=512) in a 4-deep LDS-DMA ring.  grid = (2*bpad/128, slices):
// slice y walks its share of the column tiles and writes its own gradient slice (summed by the
// finish kernel) -- that is what fills all 256 CUs at B=8192 without atomics.
// ---------------------------------------------------------------------------------------------
// SW (sample weights): intra-modal W = s E (wrz_p k_q + wrz_q k_p); the tile's k_q are a second 128-byte
// statistics DMA per tile.
template <int DK, bool SW>
__global__ void __launch_bounds__(256, 1) fast_bwd_kernel(const bf16_t* rows, const bf16_t* cols, Geo g,
                                                          const float* rz_rows, const float* wrz_rows,
                                                          const float* rz_cols, const float* wrz_cols, float* gbuf,
                                                          int accumulate, int tiles_per_slice,
                                                          const float* krows, const float* kcols) {
    constexpr int RB = DK * 32;
    constexpr int QT = 32;                 // columns per tile
    constexpr int TILE = QT * RB;
    constexpr int NST = 4;                 // ring depth (power of two): one tile consumed, up to three in flight
    constexpr int PF = (CROSSCLR_PF < DK / 2) ? CROSSCLR_PF : DK / 2;
    constexpr int DT = DK / 2;             // 32-wide output fragments
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[NST * TILE + NST * 128 * (SW ? 2 : 1)];
    unsigned char* stat = lds + NST * TILE;  // [NST][32] floats: 1/Z (or w/Z) of the tile's columns
    unsigned char* statk = stat + NST * 128; // SW: [NST][32] floats: k of the tile's columns

    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int row0w = blockIdx.x * 128 + 32 * wave;
    const int rmod = row0w / g.bpad;
    const int r_in_mod = row0w - rmod * g.bpad + l31;

    bf16x8 pf[DK];
    {
        const bf16_t* src = rows + (size_t)(row0w + l31) * (DK * 16) + 8 * half;
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) pf[ks] = *reinterpret_cast<const bf16x8*>(src + 16 * ks);
    }
    const float rzp_inter = rz_rows[row0w + l31];
    const float rzp_intra = wrz_rows[row0w + l31];
    const float kp = SW ? krows[row0w + l31] : 1.f;

    int off8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) off8[j] = l31 * RB + ((((2 * j + half) ^ sigma16(l31)) & 15) << 4);
    // transpose-read roles: in a 16-lane group lane 4j+c addresses row j, 8-byte piece c
    const int grp = lane >> 4, i16 = lane & 15, jrow = i16 >> 2, piece = i16 & 3, dsub = grp & 1;
    // address of (row q = base + 8u + 4half + jrow, col = 32dt + 16dsub + 4piece):
    //   q*RB + 256*(dt>>2) + 64*((dt&3)^jrow) + 16*((2dsub + piece/2) ^ (2u + half)) + 8*(piece&1)
    int comb[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            comb[k][u] = (4 * half + jrow) * RB + 64 * (k ^ jrow) + 16 * ((2 * dsub + (piece >> 1)) ^ (2 * u + half)) +
                         8 * (piece & 1);

    f32x16 acc2[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[dt][r] = 0.f;

    // slice y walks USABLE tiles [y*tiles_per_slice, ...): the skipped rank's segment is cut out of the numbering, so the
    // slices stay balanced whichever rank is skipped
    const int per_rank = 2 * g.bpad / QT;
    const int skip_seg = (g.skip_rank >= g.col_rank0 && g.skip_rank < g.col_rank0 + g.col_ranks) ? g.skip_rank - g.col_rank0 : -1;
    const int usable = (g.col_ranks - (skip_seg >= 0 ? 1 : 0)) * per_rank;
    int t = blockIdx.y * tiles_per_slice;   // t, t1, t2, t3: usable-tile indices
    int t_end = t + tiles_per_slice;
    if (t_end > usable) t_end = usable;
    auto tile_of = [&](int u) { return (skip_seg >= 0 && u >= skip_seg * per_rank) ? u + per_rank : u; };
    auto next = [&](int x) { return x < t_end ? x : t_end; };
    const size_t pitch = RB;
    constexpr int NOPS = DK / 4 + 1 + (SW ? 1 : 0);  // VMEM operations one tile costs each wave (DMA pieces + statistics)
    // a block's 128 rows never straddle the modality boundary (bpad is a multiple of 128), so all four
    // waves agree on which per-column statistics array (1/Z or w/Z) a tile needs
    auto issue = [&](int u, int stage) {
        const ColTile c = col_tile(g, tile_of(u), QT);
        issue_tile_dma<RB, 4, QT>(reinterpret_cast<const unsigned char*>(cols) + c.row0 * pitch, lds + stage * TILE, wave,
                                  lane, ((c.mod == rmod) ? wrz_cols : rz_cols) + c.stat0, stat + stage * 128,
                                  SW ? kcols + c.stat0 : nullptr, statk + stage * 128);
    };
    auto wait_keep = [&](int tiles_in_flight) {  // block until all but the newest `tiles_in_flight` tiles landed
        if (tiles_in_flight >= 2) wait_dma_keep<2 * NOPS>();
        else if (tiles_in_flight == 1) wait_dma_keep<NOPS>();
        else wait_dma();
    };
    // ---- S^T = Xq . Xp^T : C[q][p], lane owns row p = l31 ----
    auto gemm1 = [&](const unsigned char* bt, const ColTile& ct) {
        f32x16 acc;
        if (CROSSCLR_ABL & 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bt[r] * 0.001f;
            return acc;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#if CROSSCLR_G1CHAINS == 2
        f32x16 acc_odd;   // second accumulator chain: consecutive MFMAs independent
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_odd[r] = 0.f;
#endif
        bf16x8 ring[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) ring[i] = *reinterpret_cast<const bf16x8*>(bt + off8[i & 7] + (i >> 3) * 256);
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
            if (CROSSCLR_ABL & 32) {
                acc = mfma_32x32x16_bf16(pf[(ks + 1) % DK], pf[ks], acc);
                continue;
            }
            const bf16x8 a_cur = ring[ks % PF];
            if (ks + PF < DK)
                ring[ks % PF] = *reinterpret_cast<const bf16x8*>(bt + off8[(ks + PF) & 7] + ((ks + PF) >> 3) * 256);
#if CROSSCLR_G1CHAINS == 2
            if (ks & 1) acc_odd = mfma_32x32x16_bf16(a_cur, pf[ks], acc_odd);
            else
#endif
            acc = mfma_32x32x16_bf16(a_cur, pf[ks], acc);
        }
        if (!(CROSSCLR_ABL & 32)) SCHED_PIPELINE(DK, 1, PF);
#if CROSSCLR_G1CHAINS == 2
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc_odd[r];
#endif
        return acc;
    };
    // ---- W = s E (1/Z_p + 1/Z_q), packed to bf16: the A fragments of the second product ----
    auto weights = [&](f32x16 x, const ColTile& ct, const float* rzq, const float* kqs, bf16x8 (&af)[2]) {
        const bool same_mod = (ct.mod == rmod);
        const bool weighted = SW && same_mod;
        const float c2 = same_mod ? g.c_intra : g.c_inter;
        const float rzp = same_mod ? rzp_intra : rzp_inter;
        if (!(CROSSCLR_ABL & 2)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = x[r] * c2 - g.m2;
            // the intra-modal self pair is excluded (its exp(0) is added analytically by the forward):
            // only on the one tile that holds the diagonal, send that scaled logit to -inf
            if (same_mod && ct.rank == g.row_rank && ct.in_mod0 == (r_in_mod - l31)) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (frag_row(r, half) == l31) x[r] = -__builtin_inff();
            }
        }
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            struct { bf16_t e[8]; } pk;
#pragma unroll
            for (int r4 = 0; r4 < 2; ++r4) {
                const int q0 = 16 * th + 8 * r4 + 4 * half;  // = frag_row(8th + 4r4, half)
                const f32x4 rq = *reinterpret_cast<const f32x4*>(rzq + q0);
                f32x4 kq = {1.f, 1.f, 1.f, 1.f};
                if (weighted) kq = *reinterpret_cast<const f32x4*>(kqs + q0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = x[8 * th + 4 * r4 + j];
                    const float zz = weighted ? (rzp * kq[j] + rq[j] * kp) : (rzp + rq[j]);
                    pk.e[4 * r4 + j] = f32_to_bf16_bits((CROSSCLR_ABL & 2) ? v : fast_exp2(v) * zz);
                }
            }
            af[th] = __builtin_bit_cast(bf16x8, pk);
        }
    };
    struct Pair { s16x4 lo, hi; };
    auto trpair = [&](const unsigned char* bt, int tp, int dt) {
        if (CROSSCLR_ABL & 16) return __builtin_bit_cast(Pair, pf[(2 * dt + tp) % DK]);
        Pair p;
        p.lo = lds_read_tr16_b64(bt + comb[dt & 3][0] + (16 * tp) * RB + 256 * (dt >> 2));
        p.hi = lds_read_tr16_b64(bt + comb[dt & 3][1] + (16 * tp + 8) * RB + 256 * (dt >> 2));
        return p;
    };

    // ring of NST stages: tile t being consumed, t1 and t2 in flight
    t = next(t);
    int t1 = next(t + 1), t2 = next(t1 + 1);
    if (t < t_end) issue(t, 0);
    if (t1 < t_end) issue(t1, 1);
    if (t2 < t_end) issue(t2, 2);
    int stage = 0;
    while (t < t_end) {
        const int t3 = next(t2 + 1);
        if (!(CROSSCLR_ABL & 8)) {
            wait_keep((t1 < t_end) + (t2 < t_end));   // tile t landed (its DMA was issued before t1's and t2's)
            barrier_keep_dma();                         // ... everywhere; and every wave finished tile t-1
            if (t3 < t_end) issue(t3, (stage + 3) & (NST - 1));
        }
        const unsigned char* bt = lds + stage * TILE;
        const ColTile ct = col_tile(g, tile_of(t), QT);
        const f32x16 acc = gemm1(bt, ct);

        bf16x8 af[2];
        weights(acc, ct, reinterpret_cast<const float*>(stat + stage * 128), reinterpret_cast<const float*>(statk + stage * 128), af);
        // ---- G[p][:] += W[p][q] . Xq[q][:]  (contraction over the tile's 32 rows).  Item i = (k-step i / DT, output fragment
        // i % DT); B fragments by asm transpose reads PF items ahead of the fused {counted wait, MFMA} that consumes them
        // (the ds_read_tr BUILTIN drew a vmcnt(0) in front of every tile's second product: the whole DMA ring drained there)
        if (!(CROSSCLR_ABL & 1)) {
            constexpr int NI = 2 * DT;
            constexpr int PF2 = PF < NI / 2 ? PF : NI / 2;
            const auto xa = lds_addr(bt);
            decltype(lds_addr(bt)) base[4][2];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < 2; ++u) base[k][u] = xa + comb[k][u];
            Pair ring[PF2];
            auto fetch = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int tp = i / DT, dt = i % DT;
                ring[i % PF2].lo = lds_read_tr16_b64_async<(16 * tp) * RB + 256 * (dt >> 2)>(base[dt & 3][0]);
                ring[i % PF2].hi = lds_read_tr16_b64_async<(16 * tp + 8) * RB + 256 * (dt >> 2)>(base[dt & 3][1]);
            };
            static_for<PF2>([&](auto ic) { fetch(ic); });
            static_for<NI>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int tp = i / DT, dt = i % DT;
                constexpr int later = (NI - 1 - i) < (PF2 - 1) ? (NI - 1 - i) : (PF2 - 1);
                mfma_after_lgkm<2 * later>(acc2[dt], af[tp], __builtin_bit_cast(bf16x8, ring[i % PF2]));
                if constexpr (i + PF2 < NI) fetch(IdxC<i + PF2>{});
                sched_fence();
            });
        }
        stage = (stage + 1) & (NST - 1);
        t = t1;
        t1 = t2;
        t2 = t3;
    }
    float* gslice = gbuf + (size_t)blockIdx.y * 2 * g.bpad * (DK * 16);
    if (accumulate) {   // (hoisted: a per-element select made every store wait for its own load)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gslice[(size_t)(row0w + frag_row(r, half)) * (DK * 16) + 32 * dt + l31] += acc2[dt][r];
            sched_fence();
        }
    } else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gslice[(size_t)(row0w + frag_row(r, half)) * (DK * 16) + 32 * dt + l31] = acc2[dt][r];
            sched_fence();
        }
    }
}

// (the backward from the SAVED exponentials is fast_bwd_dsl_kernel, crossclr_kernels_dsl.h)

// ---------------------------------------------------------------------------------------------
// backward, 16-row wavefronts (v_mfma_f32_16x16x32_bf16).  A wave owns 16 rows: 16 x Dpad fp32 gradient
// accumulators (Dpad/4 registers) + Dpad/8 registers of row fragments -- half of what the 32-row kernel
// needs, so at Dpad <= 512 TWO waves fit per SIMD (8 waves x 16 rows = 128 rows per block) and hide each
// other's LDS latency, barriers and exp/weight VALU; and Dpad up to 1024 fits at all (4 waves, 64 rows).
// The price is twice the LDS read traffic per MFMA flop (a 16-row wave re-reads the whole column tile for
// half the rows).  Same dataflow as fast_bwd_kernel: S^T = Xq.Xp^T with swapped operands (lane owns row
// p = lane&15 and four columns of each 16x16 fragment) -> W in registers; the lane's eight weights of a
// 32-column tile are exactly ONE A fragment of the second product (k-slot (g,j) <-> column 4g+j / 16+4g+j-4),
// whose B fragments are two ds_read_b64_tr_b16 of the same tile.
// Tile image: row q at q*RB, 16-byte chunk c at slot (c & ~15) | ((c ^ ((q&7)<<1)) & 15): conflict-free for
// the ds_read_b128 of the first product (a 16-lane group mixes two k-groups: even/odd slots) and for the
// transpose reads of the second (8 rows x 2 chunks -> 16 distinct slots per 32 lanes).
// ---------------------------------------------------------------------------------------------
template <int DKK, int NW, bool SW>
__global__ void __launch_bounds__(64 * NW, NW / 4) fast_bwd16_kernel(const bf16_t* rows, const bf16_t* cols, Geo g,
                                                                    const float* rz_rows, const float* wrz_rows,
                                                                    const float* rz_cols, const float* wrz_cols,
                                                                    float* gbuf, int accumulate, int tiles_per_slice,
                                                                    const float* krows, const float* kcols) {
    constexpr int DP = DKK * 32;           // padded embedding width
    constexpr int RB = DP * 2;             // bytes per operand row
    constexpr int QT = 32;
    constexpr int TILE = QT * RB;
    constexpr int NST = (4 * TILE + 512 <= 160 * 1024) ? 4 : 2;   // ring depth
    constexpr int DS = DP / 16;            // 16-wide output fragments
    constexpr int PF = (CROSSCLR_PF < DKK / 2) ? CROSSCLR_PF : (DKK / 2 > 0 ? DKK / 2 : 1);
    constexpr int NOPS = QT * RB / 1024 / NW + 1 + (SW ? 1 : 0);
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[NST * TILE + NST * 128 * (SW ? 2 : 1)];
    unsigned char* stat = lds + NST * TILE;
    unsigned char* statk = stat + NST * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int i16 = lane & 15, g4 = lane >> 4;
    const int row0w = blockIdx.x * (16 * NW) + 16 * wave;
    const int rmod = row0w / g.bpad;
    const int r0_in_mod = row0w - rmod * g.bpad;         // first row of the wave inside its modality (multiple of 16)

    bf16x8 pf[DKK];
    {
        const bf16_t* src = rows + (size_t)(row0w + i16) * DP + 8 * g4;
#pragma unroll
        for (int ks = 0; ks < DKK; ++ks) pf[ks] = *reinterpret_cast<const bf16x8*>(src + 32 * ks);
    }
    const float rzp_inter = rz_rows[row0w + i16];
    const float rzp_intra = wrz_rows[row0w + i16];
    const float kp = SW ? krows[row0w + i16] : 1.f;

    // first product: row q = i16 (+16) of the tile, chunk 4ks + g4
    int offA[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) offA[j] = i16 * RB + ((((4 * j + g4) ^ ((i16 & 7) << 1)) & 15) << 4);
    // second product: in a 16-lane group (= k-group g4) lane 4j+c addresses row 4g4 + j (+16), 8-byte piece c
    const int jrow = i16 >> 2, piece = i16 & 3;
    int comb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        comb[k] = (4 * g4 + jrow) * RB + 32 * (k ^ (4 * (g4 & 1) + jrow)) + 16 * (piece >> 1) + 8 * (piece & 1);

    f32x4 acc2[DS];
#pragma unroll
    for (int d = 0; d < DS; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[d][r] = 0.f;

    const int per_rank = 2 * g.bpad / QT;
    const int skip_seg = (g.skip_rank >= g.col_rank0 && g.skip_rank < g.col_rank0 + g.col_ranks) ? g.skip_rank - g.col_rank0 : -1;
    const int usable = (g.col_ranks - (skip_seg >= 0 ? 1 : 0)) * per_rank;
    int t0 = blockIdx.y * tiles_per_slice;   // usable-tile indices (the skipped rank's segment is cut out of the numbering)
    int t_end = t0 + tiles_per_slice;
    if (t_end > usable) t_end = usable;
    auto tile_of = [&](int u) { return (skip_seg >= 0 && u >= skip_seg * per_rank) ? u + per_rank : u; };
    auto next = [&](int x) { return x < t_end ? x : t_end; };
    const size_t pitch = RB;
    auto issue = [&](int u, int stage) {
        const ColTile c = col_tile(g, tile_of(u), QT);
        issue_tile_dma<RB, NW, QT, 1>(reinterpret_cast<const unsigned char*>(cols) + c.row0 * pitch, lds + stage * TILE,
                                      wave, lane, ((c.mod == rmod) ? wrz_cols : rz_cols) + c.stat0, stat + stage * 128,
                                      SW ? kcols + c.stat0 : nullptr, statk + stage * 128);
    };
    auto wait_keep = [&](int tiles_in_flight) {
        if (tiles_in_flight >= 2) wait_dma_keep<2 * NOPS>();
        else if (tiles_in_flight == 1) wait_dma_keep<NOPS>();
        else wait_dma();
    };
    struct Pair { s16x4 lo, hi; };
    int tl[NST];  // tl[0]: tile being consumed; tl[1..NST-2]: in flight; tl[NST-1]: issued after the barrier
    tl[0] = next(t0);
#pragma unroll
    for (int k = 1; k < NST; ++k) tl[k] = next(tl[k - 1] + 1);
#pragma unroll
    for (int k = 0; k < NST - 1; ++k)
        if (tl[k] < t_end) issue(tl[k], k);
    int stage = 0;
    while (tl[0] < t_end) {
        int inflight = 0;
#pragma unroll
        for (int k = 1; k < NST - 1; ++k) inflight += (tl[k] < t_end);
        wait_keep(inflight);
#if CROSSCLR_TUNE & 4
        __syncthreads();
#else
        barrier_keep_dma();
#endif
        if (tl[NST - 1] < t_end) issue(tl[NST - 1], (stage + NST - 1) % NST);
        const unsigned char* bt = lds + stage * TILE;
        const ColTile ct = col_tile(g, tile_of(tl[0]), QT);
        // ---- S^T = Xq . Xp^T: two 16x16 fragments (columns 0-15 and 16-31 of the tile), independent chains ----
        f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
        {
            bf16x8 r0[PF], r1[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const unsigned char* a = bt + offA[i & 3] + (i >> 2) * 256;
                r0[i] = *reinterpret_cast<const bf16x8*>(a);
                r1[i] = *reinterpret_cast<const bf16x8*>(a + 16 * RB);
            }
#pragma unroll
            for (int ks = 0; ks < DKK; ++ks) {
                const bf16x8 a0 = r0[ks % PF], a1 = r1[ks % PF];
                if (ks + PF < DKK) {
                    const unsigned char* a = bt + offA[(ks + PF) & 3] + ((ks + PF) >> 2) * 256;
                    r0[ks % PF] = *reinterpret_cast<const bf16x8*>(a);
                    r1[ks % PF] = *reinterpret_cast<const bf16x8*>(a + 16 * RB);
                }
                x0 = mfma_16x16x32_bf16(a0, pf[ks], x0);
                x1 = mfma_16x16x32_bf16(a1, pf[ks], x1);
            }
            SCHED_PIPELINE_FWD(DKK, PF);
        }
        // ---- W = s E (1/Z_p + 1/Z_q) -> ONE bf16 A fragment for the whole tile ----
        const bool same_mod = (ct.mod == rmod);
        const float c2 = same_mod ? g.c_intra : g.c_inter;
        const float rzp = same_mod ? rzp_intra : rzp_inter;
#pragma unroll
        for (int r = 0; r < 4; ++r) { x0[r] = x0[r] * c2 - g.m2; x1[r] = x1[r] * c2 - g.m2; }
        if (same_mod && ct.rank == g.row_rank && ct.in_mod0 == (r0_in_mod & ~31)) {  // tile holds the wave's diagonal
            const int pt = (r0_in_mod & 31) + i16;   // this lane's row as a column index of the tile
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (4 * g4 + r == pt) x0[r] = -__builtin_inff();
                if (16 + 4 * g4 + r == pt) x1[r] = -__builtin_inff();
            }
        }
        const float* rzq = reinterpret_cast<const float*>(stat + stage * 128);
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(rzq + 4 * g4);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(rzq + 16 + 4 * g4);
        struct { bf16_t e[8]; } pk;
        if (SW && same_mod) {
            const float* kqs = reinterpret_cast<const float*>(statk + stage * 128);
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(kqs + 4 * g4);
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(kqs + 16 + 4 * g4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pk.e[r] = f32_to_bf16_bits(fast_exp2(x0[r]) * (rzp * k0[r] + q0[r] * kp));
                pk.e[4 + r] = f32_to_bf16_bits(fast_exp2(x1[r]) * (rzp * k1[r] + q1[r] * kp));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pk.e[r] = f32_to_bf16_bits(fast_exp2(x0[r]) * (rzp + q0[r]));
                pk.e[4 + r] = f32_to_bf16_bits(fast_exp2(x1[r]) * (rzp + q1[r]));
            }
        }
        const bf16x8 af = __builtin_bit_cast(bf16x8, pk);
        // ---- G[p][:] += W[p][q] . Xq[q][:] : one k-step (32 columns) per 16-wide output fragment; B fragments by asm
        // transpose reads PF2 fragments ahead of their MFMA (the builtin form drained the DMA ring: see fast_bwd_kernel) ----
        {
            constexpr int PF2 = CROSSCLR_PF < DS / 2 ? CROSSCLR_PF : DS / 2;
            const auto xa = lds_addr(bt);
            decltype(lds_addr(bt)) base[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) base[k] = xa + comb[k];
            Pair ring[PF2];
            auto fetch = [&](auto ic) {
                constexpr int ds = decltype(ic)::value;
                ring[ds % PF2].lo = lds_read_tr16_b64_async<256 * (ds >> 3)>(base[ds & 7]);
                ring[ds % PF2].hi = lds_read_tr16_b64_async<256 * (ds >> 3) + 16 * RB>(base[ds & 7]);
            };
            static_for<PF2>([&](auto ic) { fetch(ic); });
            static_for<DS>([&](auto ic) {
                constexpr int ds = decltype(ic)::value;
                constexpr int later = (DS - 1 - ds) < (PF2 - 1) ? (DS - 1 - ds) : (PF2 - 1);
                wait_lgkm<2 * later>(ring[ds % PF2].lo, ring[ds % PF2].hi);
                acc2[ds] = mfma_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, ring[ds % PF2]), acc2[ds]);
                if constexpr (ds + PF2 < DS) fetch(IdxC<ds + PF2>{});
                if (!(CROSSCLR_TUNE & 2)) sched_fence();
            });
        }
#pragma unroll
        for (int k = 0; k < NST - 1; ++k) tl[k] = tl[k + 1];
        tl[NST - 1] = next(tl[NST - 2] + 1);
        stage = (stage + 1) % NST;
    }
    float* gslice = gbuf + (size_t)blockIdx.y * 2 * g.bpad * DP;
    if (accumulate) {   // 16 loads in flight, then add + store
#pragma unroll
        for (int d4 = 0; d4 < DS; d4 += 4) {
            float o[4][4];
#pragma unroll
            for (int ds = d4; ds < d4 + 4; ++ds)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[ds - d4][r] = gslice[(size_t)(row0w + 4 * g4 + r) * DP + 16 * ds + i16];
            sched_fence();
#pragma unroll
            for (int ds = d4; ds < d4 + 4; ++ds)
#pragma unroll
                for (int r = 0; r < 4; ++r) gslice[(size_t)(row0w + 4 * g4 + r) * DP + 16 * ds + i16] = o[ds - d4][r] + acc2[ds][r];
            sched_fence();
        }
    } else {
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gslice[(size_t)(row0w + 4 * g4 + r) * DP + 16 * ds + i16] = acc2[ds][r];
            if ((ds & 3) == 3) sched_fence();
        }
    }
}

}  // namespace crossclr
#include "crossclr_kernels_sym.h"
#include "crossclr_kernels_dsl.h"
#include "crossclr_kernels_dslp.h"
#include "crossclr_kernels_symp.h"
namespace crossclr {

// ---------------------------------------------------------------------------------------------
// host-side launchers (called from crossclr_api.cpp)
// ---------------------------------------------------------------------------------------------
#ifndef CROSSCLR_KERNELS_ONLY   // (tools/kernel_asm.py compiles single kernels without the launchers)
#ifdef CROSSCLR_EMU
#define CROSSCLR_FAST_LAUNCH(kernel, grid, block, stream, ...) emu::launch(kernel, grid, block, __VA_ARGS__)
#else
#define CROSSCLR_FAST_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, 0, (hipStream_t)(stream), __VA_ARGS__)
#endif

// The launchers that instantiate the heavy kernel templates are "leaves": in the default single-translation-unit build (tests/emu, tools/
// build_variant.py) they are static inline like everything else here; the product build (build.py, -DCROSSCLR_SPLIT) compiles
// crossclr_api.cpp with the leaves only DECLARED and one small translation unit per leaf (csrc/tu_*.cpp: -DCROSSCLR_TU_<LEAF>) that
// defines it -- the instantiations then compile in parallel instead of one after the other (6+ minutes -> the longest leaf).
#ifdef CROSSCLR_SPLIT
#define CROSSCLR_LEAF
#else
#define CROSSCLR_LEAF static inline
#endif
#if !defined(CROSSCLR_SPLIT) || defined(CROSSCLR_TU_FWD)
#define CROSSCLR_DEF_FWD 1
#endif
#if !defined(CROSSCLR_SPLIT) || defined(CROSSCLR_TU_FWDP)
#define CROSSCLR_DEF_FWDP 1
#endif
#if !defined(CROSSCLR_SPLIT) || defined(CROSSCLR_TU_SAVED_LDS)
#define CROSSCLR_DEF_SAVED_LDS 1
#endif
#if !defined(CROSSCLR_SPLIT) || defined(CROSSCLR_TU_SAVED_WIDE)
#define CROSSCLR_DEF_SAVED_WIDE 1
#endif
#if !defined(CROSSCLR_SPLIT) || defined(CROSSCLR_TU_SAVED_XF1)
#define CROSSCLR_DEF_SAVED_XF1 1
#endif
#if !defined(CROSSCLR_SPLIT) || defined(CROSSCLR_TU_SAVED_XFP)
#define CROSSCLR_DEF_SAVED_XFP 1
#endif
#if !defined(CROSSCLR_SPLIT) || defined(CROSSCLR_TU_RECOMP)
#define CROSSCLR_DEF_RECOMP 1
#endif

static inline int fast_fwd_tpr(int Dpad) { return Dpad <= 512 ? 8 : 4; }   // waves per block = 32-column tiles per row block

static inline FwdWork fast_forward_work(const crossclr_plan* p, int col_ranks, int skip_rank, bool symmetric, bool pairs = false) {
    const int usable = (col_ranks - (skip_rank >= 0 ? 1 : 0)) * (2 * p->bpad / 32);
    return fwd_make_work(symmetric ? 1 : (pairs ? 3 : 2), p->bpad, usable, p->fwd_blocks, fast_fwd_tpr(p->Dpad));
}

// the forward of whole batches (crossclr_kernels_symp.h: one unbroken MFMA stream per wave): Dpad <= 1024, b a multiple of 128, a stash below 4 GiB
#ifndef CROSSCLR_DEF_FWDP
int fast_forward_pair(const crossclr_plan* p, const Geo& g, const FwdWork& wk, const void* rows, const void* cols, float* part, float* colpart,
                      int* header, int kind, const float* krows, const float* kcols, void* stash, size_t stash_bytes, const FwdPerm& perm,
                      void* stream);
#else
CROSSCLR_LEAF int fast_forward_pair(const crossclr_plan* p, const Geo& g, const FwdWork& wk, const void* rows, const void* cols, float* part,
                                    float* colpart, int* header, int kind, const float* krows, const float* kcols, void* stash,
                                    size_t stash_bytes, const FwdPerm& perm, void* stream) {
    note_kernel(0, "fast_fwd_pair_kernel");
    const bf16_t* r = (const bf16_t*)rows;
    const bf16_t* c = (const bf16_t*)cols;
    unsigned char* st = (unsigned char*)stash;
    const unsigned sb = (unsigned)stash_bytes;
    const bool sw = krows != nullptr && kcols != nullptr;
    dim3 grid(wk.nblk), block(256);
    (void)r; (void)c; (void)st; (void)sb; (void)grid; (void)block; (void)sw;
#define CROSSCLR_LZ4(DK, NH, KS, KIND, SW, ST) \
    CROSSCLR_FAST_LAUNCH((fast_fwd_pair_kernel<DK, ST, NH, KS, KIND, SW>), grid, block, stream, r, c, g, wk, part, colpart, header, st, sb, perm, krows, kcols)
    // (sample weights: the wide instantiations only -- NH = 1 leaves room for the 16 column scales of a tile; with both 32-row halves' accumulators
    //  resident they spill, and the caller keeps those launches with fast_fwd_pipe_kernel)
#define CROSSCLR_LZ3(DK, NH, KS, KIND)                                            \
    do {                                                                          \
        if (sw && NH == 2) return CROSSCLR_E_ARG;                                 \
        if (sw && st) CROSSCLR_LZ4(DK, NH, KS, KIND, (NH == 1), true);            \
        else if (sw) CROSSCLR_LZ4(DK, NH, KS, KIND, (NH == 1), false);            \
        else if (st) CROSSCLR_LZ4(DK, NH, KS, KIND, false, true);                 \
        else CROSSCLR_LZ4(DK, NH, KS, KIND, false, false);                        \
    } while (0)
#define CROSSCLR_LZ(DK, NH, KS)                          \
    do {                                                 \
        if (kind == 1) CROSSCLR_LZ3(DK, NH, KS, 1);      \
        else if (kind == 2) CROSSCLR_LZ3(DK, NH, KS, 2); \
        else CROSSCLR_LZ3(DK, NH, KS, 3);                \
    } while (0)
    switch (p->Dpad) {
        case 128: CROSSCLR_LZ(8, 2, 1); break;
        case 256: CROSSCLR_LZ(16, 2, 1); break;
        case 384: CROSSCLR_LZ(24, 2, 1); break;
        case 512: CROSSCLR_LZ(32, 2, 1); break;
        case 768: CROSSCLR_LZ(24, 1, 2); break;      // wide operands: one 32-row half per wave, the tile in two ring stages
        case 1024: CROSSCLR_LZ(32, 1, 2); break;
        default: return CROSSCLR_E_ARG;
    }
#undef CROSSCLR_LZ
#undef CROSSCLR_LZ3
#undef CROSSCLR_LZ4
    return CROSSCLR_OK;
}
#endif   // CROSSCLR_DEF_FWDP

// the software-pipelined forward (Dpad <= 512), all three kinds, with or without saving the exponentials
#ifndef CROSSCLR_DEF_FWD
int fast_forward_pipe(const crossclr_plan* p, const Geo& g, const FwdWork& wk, const void* rows, const void* cols,
                      float* part, float* colpart, int* header, int kind, const float* krows, const float* kcols,
                      void* stash, void* stream);
#else
CROSSCLR_LEAF int fast_forward_pipe(const crossclr_plan* p, const Geo& g, const FwdWork& wk, const void* rows, const void* cols,
                                    float* part, float* colpart, int* header, int kind, const float* krows, const float* kcols,
                                    void* stash, void* stream) {
    if (wk.total <= 0) return CROSSCLR_OK;
    const bf16_t* r = (const bf16_t*)rows;
    const bf16_t* c = (const bf16_t*)cols;
    unsigned char* st = (unsigned char*)stash;
    dim3 grid(wk.nblk), block(256);
    const bool sw = krows != nullptr && kcols != nullptr;
    // XCD-aware placement of the ranges (crossclr_device.h: fwd_make_perm), computed once per work list and thread
    static const bool xcd_aware = [] { const char* e = getenv("CROSSCLR_FWD_XCD"); return !(e && e[0] == '0'); }();
    static thread_local struct { FwdWork wk; FwdPerm perm; bool valid; } pcache[4];
    static thread_local int pnext = 0;
    const FwdPerm* permp = nullptr;
    for (int i = 0; i < 4; ++i)
        if (pcache[i].valid && memcmp(&pcache[i].wk, &wk, sizeof(FwdWork)) == 0) { permp = &pcache[i].perm; break; }
    if (!permp) {
        auto& e = pcache[pnext];
        pnext = (pnext + 1) & 3;
        memset(&e.wk, 0, sizeof(FwdWork));
        e.wk = wk;
        fwd_make_perm(wk, xcd_aware, &e.perm);
        e.valid = true;
        permp = &e.perm;
    }
    const FwdPerm perm = *permp;
    // whole batches of the local block without sample weights: the kernel with one unbroken MFMA stream per wave (crossclr_kernels_symp.h;
    // the same bits as the kernel below, CROSSCLR_FWD_PAIR=0 keeps that one: A/B)
    const char* pair_env = getenv("CROSSCLR_FWD_PAIR");      // (read per launch: the bit-identity tests flip it inside one process)
    const bool pair_kernel = !(pair_env && pair_env[0] == '0');
    if (pair_kernel && (!sw || p->Dpad > 512) && g.b == g.bpad && p->Dpad <= 1024 && wk.tpr == (p->Dpad <= 512 ? 8 : 4)) {
        // rectangular / pair launches: only over OTHER ranks' columns (a launch that contains the rows' own rank needs the self-pair masks)
        bool own = false;
        if (kind != 1) {
            const int W = g.col_wrap;
            for (int i = 0; i < g.col_ranks; ++i) {
                if (kind == 2 && g.col_rank0 + i == g.skip_rank) continue;
                int rk = g.col_rank0 + i;
                if (W > 0 && rk >= W) rk -= W;
                own = own || rk == g.row_rank;
            }
        }
        const size_t per_rank = (size_t)(2 * p->bpad / 32);
        const size_t sbytes = !st ? 0 : (kind == 1 ? stash_tiles_total(wk.tpr, 2 * p->bpad / 32) * 2048 : per_rank * (size_t)wk.NT * 2048);
        const size_t col_segs = kind == 1 ? 1 : (size_t)(g.col_wrap > 0 ? g.col_wrap : g.col_ranks);
        if (!own && sbytes < ((size_t)1 << 32) && (size_t)wk.NB * wk.NT * 128 < ((size_t)1 << 32) &&
            col_segs * 2 * p->bpad * p->Dpad * 2 < ((size_t)1 << 32))
            return fast_forward_pair(p, g, wk, rows, cols, part, colpart, header, kind, krows, kcols, stash, sbytes, perm, stream);
    }
    note_kernel(0, "fast_fwd_pipe_kernel");
#define CROSSCLR_LP3(DK, KIND, SW, ST) \
    CROSSCLR_FAST_LAUNCH((fast_fwd_pipe_kernel<DK, KIND, SW, ST>), grid, block, stream, r, c, g, wk, part, colpart, header, krows, kcols, st, perm)
#define CROSSCLR_LP2(DK, KIND)                                   \
    do {                                                          \
        if (sw && st) CROSSCLR_LP3(DK, KIND, true, true);         \
        else if (sw) CROSSCLR_LP3(DK, KIND, true, false);         \
        else if (st) CROSSCLR_LP3(DK, KIND, false, true);         \
        else CROSSCLR_LP3(DK, KIND, false, false);                \
    } while (0)
#define CROSSCLR_LP(DK)                                 \
    do {                                                 \
        if (kind == 1) CROSSCLR_LP2(DK, 1);              \
        else if (kind == 2) CROSSCLR_LP2(DK, 2);         \
        else CROSSCLR_LP2(DK, 3);                        \
    } while (0)
#define CROSSCLR_LPW3(DK, KIND, SW, ST) \
    CROSSCLR_FAST_LAUNCH((fast_fwd_pipe_kernel<DK, KIND, SW, ST, 1>), grid, block, stream, r, c, g, wk, part, colpart, header, krows, kcols, st, perm)
#define CROSSCLR_LPW2(DK, KIND)                                   \
    do {                                                           \
        if (sw && st) CROSSCLR_LPW3(DK, KIND, true, true);         \
        else if (sw) CROSSCLR_LPW3(DK, KIND, true, false);         \
        else if (st) CROSSCLR_LPW3(DK, KIND, false, true);         \
        else CROSSCLR_LPW3(DK, KIND, false, false);                \
    } while (0)
#define CROSSCLR_LPW(DK)                                 \
    do {                                                  \
        if (kind == 1) CROSSCLR_LPW2(DK, 1);              \
        else if (kind == 2) CROSSCLR_LPW2(DK, 2);         \
        else CROSSCLR_LPW2(DK, 3);                        \
    } while (0)
    switch (p->Dpad) {
        case 128: CROSSCLR_LP(8); break;
        case 256: CROSSCLR_LP(16); break;
        case 384: CROSSCLR_LP(24); break;
        case 512: CROSSCLR_LP(32); break;
        case 768: CROSSCLR_LPW(48); break;     // wide operands: one 32-row half per wave
        case 1024: CROSSCLR_LPW(64); break;
        default: return CROSSCLR_E_ARG;
    }
#undef CROSSCLR_LPW
#undef CROSSCLR_LPW2
#undef CROSSCLR_LPW3
#undef CROSSCLR_LP
#undef CROSSCLR_LP2
#undef CROSSCLR_LP3
    return CROSSCLR_OK;
}
#endif   // CROSSCLR_DEF_FWD

static inline int fast_forward(const crossclr_plan* p, const Geo& g, const void* rows, const void* cols, float* part,
                               float* colpart, int* header, bool symmetric, const float* krows, const float* kcols,
                               void* stream, bool pairs = false) {
    const bool skipping = g.skip_rank >= g.col_rank0 && g.skip_rank < g.col_rank0 + g.col_ranks;
    const FwdWork wk = fast_forward_work(p, g.col_ranks, skipping ? g.skip_rank : -1, symmetric, pairs);
    if (wk.total <= 0) return CROSSCLR_OK;
    const bf16_t* r = (const bf16_t*)rows;
    const bf16_t* c = (const bf16_t*)cols;
    dim3 grid(wk.nblk);
    const bool sw = krows != nullptr && kcols != nullptr;
    (void)r; (void)c; (void)grid; (void)sw;
    // software-pipelined 4-wave kernel (crossclr_kernels_sym.h): symmetric, rectangular and pairs
    if (p->Dpad > 1024) return CROSSCLR_E_ARG;
    return fast_forward_pipe(p, g, wk, rows, cols, part, colpart, header, symmetric ? 1 : (pairs ? 3 : 2), krows, kcols, nullptr, stream);
}

// the saved-exponentials pair (symmetric local block, Dpad <= 512): bytes of the stash, forward that fills it, backward that reads it
static inline size_t fast_stash_bytes(int bpad, int Dpad) {
    return Dpad <= 512 ? stash_tiles_total(8, 2 * bpad / 32) * 2048 : (Dpad <= 1024 ? stash_tiles_total(4, 2 * bpad / 32) * 2048 : 0);
}
// wide bf16 plans (wide_bf16_dpad): the generic symmetric forward fills the 128-row-block layout
static inline size_t wide_stash_bytes(int bpad) { return stash_tiles_total(4, 2 * bpad / 32) * 2048; }
// bytes of the stash of a rectangular (remote / pairs) launch over `nranks` column ranks
static inline size_t fast_stash_bytes_rect(int bpad, int Dpad, int nranks) {
    return Dpad <= 1024 ? (size_t)(2 * bpad / 32) * (size_t)(2 * bpad / 32) * (size_t)nranks * 2048 : 0;
}
// wide bf16 plans (Dpad > 1024): the same 2-KiB records, written by the generic forward over the rank range
static inline size_t wide_stash_bytes_rect(int bpad, int nranks) {
    return (size_t)(2 * bpad / 32) * (size_t)(2 * bpad / 32) * (size_t)nranks * 2048;
}
static inline int fast_forward_save(const crossclr_plan* p, const Geo& g, const void* x, float* part, float* colpart,
                                    int* header, const float* ks, void* stash, void* stream) {
    const FwdWork wk = fast_forward_work(p, 1, -1, true);
    return fast_forward_pipe(p, g, wk, x, x, part, colpart, header, 1, ks, ks, stash, stream);
}
// mode 0: the symmetric local block (x is both operands).  mode 1: a rectangular block whose forward saved its exponentials (g describes
// the column ranks; cols / rz_cols / wrz_cols / kc use the column operand's indexing).  mode 2: the transpose of ONE rectangular block
// (crossclr_backward_rect_saved_t): output rows = the partner's, cols / *_cols / kc = this rank's LOCAL operand and statistics,
// rz / wrz / ks = the partner's statistics; g.col_ranks = rank segments per stash row, g.skip_rank = the partner's segment.
struct SavedLaunch {      // one launch of a saved backward, as the leaves below take it
    const crossclr_plan* p; Geo g; const void* cols; const void* stash; const float *rz, *wrz, *rz_cols, *wrz_cols; float* gbuf; int accumulate;
    const float *ks, *kc; int mode; int tps; void* stream; size_t stash_bytes;
};
// the pair kernel (crossclr_kernels_dslp.h): local block, fragment-major operand, two tiles per barrier interval
#ifndef CROSSCLR_DEF_SAVED_XFP
int launch_saved_xfp(const SavedLaunch& a);
#else
CROSSCLR_LEAF int launch_saved_xfp(const SavedLaunch& a) {
    note_kernel(1, "fast_bwd_xfp_kernel");
    const crossclr_plan* p = a.p;
    const Geo& g = a.g;
    dim3 grid(2 * p->bpad / 128, p->bwd_slices), block(256);
    void* stream = a.stream;
    const unsigned char* st = (const unsigned char*)a.stash;
    const unsigned sb = (unsigned)a.stash_bytes;
    const unsigned char* xfo = (const unsigned char*)a.cols;
    const float *rz = a.rz, *wrz = a.wrz, *rz_cols = a.rz_cols, *wrz_cols = a.wrz_cols, *ks = a.ks, *kc = a.kc;
    float* gbuf = a.gbuf;
    const int accumulate = a.accumulate, tps2 = a.tps, mode = a.mode;
    (void)grid; (void)block; (void)stream; (void)st; (void)sb; (void)xfo; (void)rz; (void)wrz; (void)rz_cols; (void)wrz_cols; (void)kc; (void)gbuf;
    (void)accumulate; (void)tps2; (void)mode;
#ifdef CROSSCLR_DSL_MINIMAL   // tuning builds (tools/build_variant.py): only the headline instantiation is compiled
    if (p->Dpad != 512 || ks || mode != 0) return CROSSCLR_E_ARG;
    CROSSCLR_FAST_LAUNCH((fast_bwd_xfp_kernel<32, false, 0, 1, 8>), grid, block, stream, xfo, st, sb, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps2, ks, kc);
    return CROSSCLR_OK;
#else
#define CROSSCLR_LBP3(DK, SW, XP, TPRF, GRID)                                                                                                                                    \
    do {                                                                                                                                                                         \
        if (mode == 0) CROSSCLR_FAST_LAUNCH((fast_bwd_xfp_kernel<DK, SW, 0, XP, TPRF>), GRID, block, stream, xfo, st, sb, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps2, ks, kc);      \
        else if (mode == 1) CROSSCLR_FAST_LAUNCH((fast_bwd_xfp_kernel<DK, SW, 1, XP, TPRF>), GRID, block, stream, xfo, st, sb, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps2, ks, kc); \
        else CROSSCLR_FAST_LAUNCH((fast_bwd_xfp_kernel<DK, SW, 2, XP, TPRF>), GRID, block, stream, xfo, st, sb, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps2, ks, kc);                \
    } while (0)
#define CROSSCLR_LBP(DK, XP, TPRF, GRID) do { if (ks) CROSSCLR_LBP3(DK, true, XP, TPRF, GRID); else CROSSCLR_LBP3(DK, false, XP, TPRF, GRID); } while (0)
    dim3 gridp2(2 * p->bpad / 128, p->bwd_slices, 2);
    switch (p->Dpad) {
        case 128: CROSSCLR_LBP(8, 1, 8, grid); break;
        case 256: CROSSCLR_LBP(16, 1, 8, grid); break;
        case 384: CROSSCLR_LBP(24, 1, 8, grid); break;
        case 512: CROSSCLR_LBP(32, 1, 8, grid); break;
        case 768: CROSSCLR_LBP(24, 2, 4, gridp2); break;      // two column parts of Dpad / 2 (blockIdx.z)
        case 1024: CROSSCLR_LBP(32, 2, 4, gridp2); break;
        default: return CROSSCLR_E_ARG;
    }
#undef CROSSCLR_LBP
#undef CROSSCLR_LBP3
    return CROSSCLR_OK;
#endif
}
#endif   // CROSSCLR_DEF_SAVED_XFP

// fast_bwd_dsl_kernel<..., XF>: local block, fragment-major operand, one tile per barrier interval
#ifndef CROSSCLR_DEF_SAVED_XF1
int launch_saved_xf1(const SavedLaunch& a);
#else
CROSSCLR_LEAF int launch_saved_xf1(const SavedLaunch& a) {
    note_kernel(1, "fast_bwd_dsl_kernel (fragment-major, one tile)");
    const crossclr_plan* p = a.p;
    const Geo& g = a.g;
    dim3 grid(2 * p->bpad / 128, p->bwd_slices), block(256);
    void* stream = a.stream;
    const bf16_t* c = (const bf16_t*)a.cols;
    const unsigned char* st = (const unsigned char*)a.stash;
    const float *rz = a.rz, *wrz = a.wrz, *rz_cols = a.rz_cols, *wrz_cols = a.wrz_cols, *ks = a.ks, *kc = a.kc;
    float* gbuf = a.gbuf;
    const int accumulate = a.accumulate, tps = a.tps;
    (void)grid; (void)block; (void)stream; (void)c; (void)st; (void)rz; (void)wrz; (void)rz_cols; (void)wrz_cols; (void)kc; (void)gbuf; (void)accumulate; (void)tps;
#ifdef CROSSCLR_DSL_MINIMAL
    if (p->Dpad != 512 || ks) return CROSSCLR_E_ARG;
    CROSSCLR_FAST_LAUNCH((fast_bwd_dsl_kernel<32, false, 0, 1, 8, true>), grid, block, stream, c, st, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps, ks, kc);
    return CROSSCLR_OK;
#else
#define CROSSCLR_LBX2(DK) do { dim3 grid2(2 * p->bpad / 128, p->bwd_slices, 2);                                                                  \
                               if (ks) CROSSCLR_FAST_LAUNCH((fast_bwd_dsl_kernel<DK, true, 0, 2, 4, true>), grid2, block, stream, c, st, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps, ks, kc); \
                               else CROSSCLR_FAST_LAUNCH((fast_bwd_dsl_kernel<DK, false, 0, 2, 4, true>), grid2, block, stream, c, st, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps, ks, kc); } while (0)
#define CROSSCLR_LBX(DK) do { if (ks) CROSSCLR_FAST_LAUNCH((fast_bwd_dsl_kernel<DK, true, 0, 1, 8, true>), grid, block, stream, c, st, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps, ks, kc); \
                              else CROSSCLR_FAST_LAUNCH((fast_bwd_dsl_kernel<DK, false, 0, 1, 8, true>), grid, block, stream, c, st, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps, ks, kc); } while (0)
    switch (p->Dpad) {
        case 128: CROSSCLR_LBX(8); break;
        case 256: CROSSCLR_LBX(16); break;
        case 384: CROSSCLR_LBX(24); break;
        case 512: CROSSCLR_LBX(32); break;
        case 768: CROSSCLR_LBX2(24); break;      // two column parts of Dpad / 2 (blockIdx.z), like the LDS-staged launch
        case 1024: CROSSCLR_LBX2(32); break;
        default: return CROSSCLR_E_ARG;
    }
#undef CROSSCLR_LBX
#undef CROSSCLR_LBX2
    return CROSSCLR_OK;
#endif
}
#endif   // CROSSCLR_DEF_SAVED_XF1

// fast_bwd_dsl_kernel, column tiles staged through LDS: the local block (mode 0), rectangular blocks (1) and their transposes (2)
#ifndef CROSSCLR_DEF_SAVED_LDS
int launch_saved_lds(const SavedLaunch& a);
#else
CROSSCLR_LEAF int launch_saved_lds(const SavedLaunch& a) {
    note_kernel(1, "fast_bwd_dsl_kernel (LDS-staged)");
    const crossclr_plan* p = a.p;
    const Geo& g = a.g;
    dim3 grid(2 * p->bpad / 128, p->bwd_slices), block(256);
    void* stream = a.stream;
    const bf16_t* c = (const bf16_t*)a.cols;
    const unsigned char* st = (const unsigned char*)a.stash;
    const float *rz = a.rz, *wrz = a.wrz, *rz_cols = a.rz_cols, *wrz_cols = a.wrz_cols, *ks = a.ks, *kc = a.kc;
    float* gbuf = a.gbuf;
    const int accumulate = a.accumulate, tps = a.tps, mode = a.mode;
    (void)grid; (void)block; (void)stream; (void)c; (void)st; (void)rz; (void)wrz; (void)rz_cols; (void)wrz_cols; (void)kc; (void)gbuf; (void)accumulate; (void)tps; (void)mode;
#ifdef CROSSCLR_DSL_MINIMAL
    if (p->Dpad != 512 || ks || mode != 0) return CROSSCLR_E_ARG;
    CROSSCLR_FAST_LAUNCH((fast_bwd_dsl_kernel<32, false, 0>), grid, block, stream, c, st, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps, ks, kc);
    return CROSSCLR_OK;
#else
#define CROSSCLR_LB3(DK, SW, XP, TPRF, GRID)                                                                                                   \
    do {                                                                                                                                        \
        if (mode == 0) CROSSCLR_FAST_LAUNCH((fast_bwd_dsl_kernel<DK, SW, 0, XP, TPRF>), GRID, block, stream, c, st, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps, ks, kc);      \
        else if (mode == 1) CROSSCLR_FAST_LAUNCH((fast_bwd_dsl_kernel<DK, SW, 1, XP, TPRF>), GRID, block, stream, c, st, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps, ks, kc); \
        else CROSSCLR_FAST_LAUNCH((fast_bwd_dsl_kernel<DK, SW, 2, XP, TPRF>), GRID, block, stream, c, st, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps, ks, kc);                \
    } while (0)
#define CROSSCLR_LB(DK, XP, TPRF, GRID) do { if (ks) CROSSCLR_LB3(DK, true, XP, TPRF, GRID); else CROSSCLR_LB3(DK, false, XP, TPRF, GRID); } while (0)
    if (p->Dpad > 512) {   // two column parts of Dpad/2
        dim3 grid2(2 * p->bpad / 128, p->bwd_slices, 2);
        switch (p->Dpad) {
            case 768: CROSSCLR_LB(24, 2, 4, grid2); break;
            case 1024: CROSSCLR_LB(32, 2, 4, grid2); break;
            default: return CROSSCLR_E_ARG;
        }
        return CROSSCLR_OK;
    }
    switch (p->Dpad) {
        case 128: CROSSCLR_LB(8, 1, 8, grid); break;
        case 256: CROSSCLR_LB(16, 1, 8, grid); break;
        case 384: CROSSCLR_LB(24, 1, 8, grid); break;
        case 512: CROSSCLR_LB(32, 1, 8, grid); break;
        default: return CROSSCLR_E_ARG;
    }
#undef CROSSCLR_LB
#undef CROSSCLR_LB3
    return CROSSCLR_OK;
#endif
}
#endif   // CROSSCLR_DEF_SAVED_LDS

// fast_bwd_dsl_kernel for wide bf16 plans (Dpad > 1024: 3 ... 8 column parts, blockIdx.z): the local symmetric block from the records the
// generic forward saved (fwd_sums_kernel<bf16_t, ..., ST, SYM>)
#ifndef CROSSCLR_DEF_SAVED_WIDE
int launch_saved_wide(const SavedLaunch& a);
int launch_saved_wide_xfp(const SavedLaunch& a);
#else
CROSSCLR_LEAF int launch_saved_wide(const SavedLaunch& a) {
    note_kernel(1, "fast_bwd_dsl_kernel (wide, column parts)");
    const crossclr_plan* p = a.p;
    const Geo& g = a.g;
    dim3 block(256);
    void* stream = a.stream;
    const bf16_t* c = (const bf16_t*)a.cols;
    const unsigned char* st = (const unsigned char*)a.stash;
    const float *rz = a.rz, *wrz = a.wrz, *rz_cols = a.rz_cols, *wrz_cols = a.wrz_cols, *ks = a.ks, *kc = a.kc;
    float* gbuf = a.gbuf;
    const int accumulate = a.accumulate, tps = a.tps;
    (void)block; (void)stream; (void)c; (void)st; (void)rz; (void)wrz; (void)rz_cols; (void)wrz_cols; (void)kc; (void)gbuf; (void)accumulate; (void)tps;
    const int mode = a.mode;
    if (mode != 0 && mode != 1) return CROSSCLR_E_ARG;       // (1: a rectangular block against other ranks' columns, crossclr_backward_rect_saved)
#ifdef CROSSCLR_DSL_MINIMAL
    return CROSSCLR_E_ARG;
#else
#define CROSSCLR_LBW2(DK, XP, SW) do { if (mode == 0) CROSSCLR_FAST_LAUNCH((fast_bwd_dsl_kernel<DK, SW, 0, XP, 4>), gridw, block, stream, c, st, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps, ks, kc); \
                                       else CROSSCLR_FAST_LAUNCH((fast_bwd_dsl_kernel<DK, SW, 1, XP, 4>), gridw, block, stream, c, st, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps, ks, kc); } while (0)
#define CROSSCLR_LBW(DK, XP) do { dim3 gridw(2 * p->bpad / 128, p->bwd_slices, XP);                                                            \
                                  if (ks) CROSSCLR_LBW2(DK, XP, true); else CROSSCLR_LBW2(DK, XP, false); } while (0)
    switch (p->Dpad) {
        case 1152: CROSSCLR_LBW(24, 3); break;
        case 1536: CROSSCLR_LBW(32, 3); break;
        case 2048: CROSSCLR_LBW(32, 4); break;
        case 2560: CROSSCLR_LBW(32, 5); break;
        case 3072: CROSSCLR_LBW(32, 6); break;
        case 4096: CROSSCLR_LBW(32, 8); break;
        case 5120: CROSSCLR_LBW(32, 10); break;
        case 6144: CROSSCLR_LBW(32, 12); break;
        case 8192: CROSSCLR_LBW(32, 16); break;
        default: return CROSSCLR_E_ARG;
    }
#undef CROSSCLR_LBW
#undef CROSSCLR_LBW2
    return CROSSCLR_OK;
#endif
}
// the same on the fragment-major operand with the pair kernel (crossclr_kernels_dslp.h): what the module takes from 4096 padded rows on
CROSSCLR_LEAF int launch_saved_wide_xfp(const SavedLaunch& a) {
    note_kernel(1, "fast_bwd_xfp_kernel (wide, column parts)");
    const crossclr_plan* p = a.p;
    const Geo& g = a.g;
    dim3 block(256);
    void* stream = a.stream;
    const unsigned char* st = (const unsigned char*)a.stash;
    const unsigned sb = (unsigned)a.stash_bytes;
    const unsigned char* xfo = (const unsigned char*)a.cols;
    const float *rz = a.rz, *wrz = a.wrz, *rz_cols = a.rz_cols, *wrz_cols = a.wrz_cols, *ks = a.ks, *kc = a.kc;
    float* gbuf = a.gbuf;
    const int accumulate = a.accumulate, tps2 = a.tps;
    (void)block; (void)stream; (void)st; (void)sb; (void)xfo; (void)rz; (void)wrz; (void)rz_cols; (void)wrz_cols; (void)kc; (void)gbuf; (void)accumulate; (void)tps2;
    if (a.mode != 0) return CROSSCLR_E_ARG;
#ifdef CROSSCLR_DSL_MINIMAL
    return CROSSCLR_E_ARG;
#else
#define CROSSCLR_LBWP(DK, XP) do { dim3 gridw(2 * p->bpad / 128, p->bwd_slices, XP);                                                            \
                                   if (ks) CROSSCLR_FAST_LAUNCH((fast_bwd_xfp_kernel<DK, true, 0, XP, 4>), gridw, block, stream, xfo, st, sb, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps2, ks, kc); \
                                   else CROSSCLR_FAST_LAUNCH((fast_bwd_xfp_kernel<DK, false, 0, XP, 4>), gridw, block, stream, xfo, st, sb, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tps2, ks, kc); } while (0)
    switch (p->Dpad) {
        case 1152: CROSSCLR_LBWP(24, 3); break;
        case 1536: CROSSCLR_LBWP(32, 3); break;
        case 2048: CROSSCLR_LBWP(32, 4); break;
        case 2560: CROSSCLR_LBWP(32, 5); break;
        case 3072: CROSSCLR_LBWP(32, 6); break;
        case 4096: CROSSCLR_LBWP(32, 8); break;
        case 5120: CROSSCLR_LBWP(32, 10); break;
        case 6144: CROSSCLR_LBWP(32, 12); break;
        case 8192: CROSSCLR_LBWP(32, 16); break;
        default: return CROSSCLR_E_ARG;
    }
#undef CROSSCLR_LBWP
    return CROSSCLR_OK;
#endif
}
#endif   // CROSSCLR_DEF_SAVED_WIDE

// mode 0 / 1 / 2: the LDS-staged kernel on the row-major operand -- the local symmetric block, a rectangular block, the transpose of one
// rectangular block.  mode 3: mode 0 on the fragment-major operand, one tile per barrier interval.  mode 4 / 5 / 6: modes 0 / 1 / 2 on the
// fragment-major operand with the pair kernel (crossclr_kernels_dslp.h; stash below 4 GiB: 32-bit scalar offsets, even slices).
static inline int fast_backward_saved(const crossclr_plan* p, const Geo& g, const void* cols, const void* stash, const float* rz,
                                      const float* wrz, const float* rz_cols, const float* wrz_cols, float* gbuf, int accumulate,
                                      const float* ks, const float* kc, int mode, void* stream) {
    const bool xfp = mode >= 4;
    const bool xf = mode == 3 || xfp;
    if (xfp) mode -= 4;
    if (mode == 3) mode = 0;
    const bool rect = mode == 1;
    const bool skipping = rect && g.col_wrap == 0 && g.skip_rank >= g.col_rank0 && g.skip_rank < g.col_rank0 + g.col_ranks;
    const int per_rank = 2 * p->bpad / 32;
    const int ntiles = (rect ? g.col_ranks - (skipping ? 1 : 0) : 1) * per_rank;
    if (ntiles <= 0) return CROSSCLR_OK;
    int tps = (ntiles + p->bwd_slices - 1) / p->bwd_slices;
    // every slice holds an EVEN number of tiles (the pair kernel needs it; the other kernels take the same cut so that all of them
    // produce the same bits slice by slice)
    tps = (tps + 1) & ~1;
    SavedLaunch a = {p, g, cols, stash, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, ks, kc, mode, tps, stream, 0};
    if (xfp) {
        // the saved exponentials are addressed with 32-bit scalar offsets: local triangle / [row group][usable tile] / [row group][segments x tile]
        a.stash_bytes = mode == 0 ? p->stash_bytes
                                  : (size_t)per_rank * (size_t)(mode == 1 ? ntiles : g.col_ranks * per_rank) * 2048;
        if (a.stash_bytes >= ((size_t)1 << 32)) return CROSSCLR_E_ARG;
        return p->Dpad > 1024 ? launch_saved_wide_xfp(a) : launch_saved_xfp(a);
    }
    if (p->Dpad > 1024) return xf ? CROSSCLR_E_ARG : launch_saved_wide(a);     // (wide plans: the pair kernel or the LDS-staged one)
    return xf ? launch_saved_xf1(a) : launch_saved_lds(a);
}
// which backward the fast path uses: 16-row wavefronts (rows per block 128 at Dpad <= 512, 64 above) or the
// 32-row kernel (Dpad <= 512 only)
static inline int fast_bwd_rows_per_block(int Dpad, int use16) { return use16 ? (Dpad <= 512 ? 128 : 64) : 128; }

#ifndef CROSSCLR_DEF_RECOMP
int fast_backward16(const crossclr_plan* p, const Geo& g, const void* rows, const void* cols, const float* rz_rows, const float* wrz_rows,
                    const float* rz_cols, const float* wrz_cols, float* gbuf, int accumulate, const float* krows, const float* kcols, void* stream);
int fast_backward(const crossclr_plan* p, const Geo& g, const void* rows, const void* cols, const float* rz_rows, const float* wrz_rows,
                  const float* rz_cols, const float* wrz_cols, float* gbuf, int accumulate, const float* krows, const float* kcols, void* stream);
#else
CROSSCLR_LEAF int fast_backward16(const crossclr_plan* p, const Geo& g, const void* rows, const void* cols,
                                  const float* rz_rows, const float* wrz_rows, const float* rz_cols,
                                  const float* wrz_cols, float* gbuf, int accumulate, const float* krows,
                                  const float* kcols, void* stream) {
    note_kernel(1, "fast_bwd16_kernel (recomputing)");
    const bool sw = krows != nullptr && kcols != nullptr;
    const bool skipping = g.skip_rank >= g.col_rank0 && g.skip_rank < g.col_rank0 + g.col_ranks;
    const int ntiles = (g.col_ranks - (skipping ? 1 : 0)) * 2 * p->bpad / 32;   // usable column tiles
    const int tps = (ntiles + p->bwd_slices - 1) / p->bwd_slices;
    const bf16_t* r = (const bf16_t*)rows;
    const bf16_t* c = (const bf16_t*)cols;
#define CROSSCLR_L16B(DKK, NW, SW)                                                                                  \
    CROSSCLR_FAST_LAUNCH((fast_bwd16_kernel<DKK, NW, SW>), dim3(2 * p->bpad / (16 * NW), p->bwd_slices), dim3(64 * NW), stream, r, c, g, \
                         rz_rows, wrz_rows, rz_cols, wrz_cols, gbuf, accumulate, tps, krows, kcols)
#define CROSSCLR_L16(DKK, NW)                  \
    do {                                        \
        if (sw) CROSSCLR_L16B(DKK, NW, true);   \
        else CROSSCLR_L16B(DKK, NW, false);     \
    } while (0)
    switch (p->Dpad) {
        case 128: CROSSCLR_L16(4, 8); break;
        case 256: CROSSCLR_L16(8, 8); break;
        case 384: CROSSCLR_L16(12, 8); break;
        case 512: CROSSCLR_L16(16, 8); break;
        case 768: CROSSCLR_L16(24, 4); break;
        case 1024: CROSSCLR_L16(32, 4); break;
        default: return CROSSCLR_E_ARG;
    }
#undef CROSSCLR_L16
#undef CROSSCLR_L16B
    return CROSSCLR_OK;
}

CROSSCLR_LEAF int fast_backward(const crossclr_plan* p, const Geo& g, const void* rows, const void* cols,
                                const float* rz_rows, const float* wrz_rows, const float* rz_cols,
                                const float* wrz_cols, float* gbuf, int accumulate, const float* krows,
                                const float* kcols, void* stream) {
    note_kernel(1, "fast_bwd_kernel (recomputing)");
    const bool sw = krows != nullptr && kcols != nullptr;
    const bool skipping = g.skip_rank >= g.col_rank0 && g.skip_rank < g.col_rank0 + g.col_ranks;
    const int ntiles = (g.col_ranks - (skipping ? 1 : 0)) * 2 * p->bpad / 32;   // usable column tiles
    const int tps = (ntiles + p->bwd_slices - 1) / p->bwd_slices;
    dim3 grid(2 * p->bpad / 128, p->bwd_slices), block(256);
    const bf16_t* r = (const bf16_t*)rows;
    const bf16_t* c = (const bf16_t*)cols;
#define CROSSCLR_L32(DK)                                                                                              \
    do {                                                                                                               \
        if (sw) CROSSCLR_FAST_LAUNCH((fast_bwd_kernel<DK, true>), grid, block, stream, r, c, g, rz_rows, wrz_rows, rz_cols, wrz_cols, gbuf, accumulate, tps, krows, kcols); \
        else CROSSCLR_FAST_LAUNCH((fast_bwd_kernel<DK, false>), grid, block, stream, r, c, g, rz_rows, wrz_rows, rz_cols, wrz_cols, gbuf, accumulate, tps, krows, kcols);  \
    } while (0)
    switch (p->Dpad) {
        case 128: CROSSCLR_L32(8); break;
        case 256: CROSSCLR_L32(16); break;
        case 384: CROSSCLR_L32(24); break;
        case 512: CROSSCLR_L32(32); break;
        default: return CROSSCLR_E_ARG;
    }
#undef CROSSCLR_L32
    return CROSSCLR_OK;
}
#endif   // CROSSCLR_DEF_RECOMP

#endif  // CROSSCLR_KERNELS_ONLY

}  // namespace crossclr
