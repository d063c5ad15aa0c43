// crossclr_device.h -- device-side building blocks shared by every kernel.
//
// Written for gfx950 (CDNA4): 64-lane wavefronts, v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32,
// 160 KiB LDS with 16-byte-chunk XOR swizzles sized for ds_read_b128's 64-bank rows,
// ds_read_b64_tr_b16 for operands that are contracted over their ROW index.
//
// The same source is also compiled by the host clang against tests/emu/hip_emu.h (CROSSCLR_EMU):
// that build is test infrastructure (lane-level index-math checks on CPU), never the product.
#pragma once

#ifdef CROSSCLR_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace crossclr {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef unsigned short bf16_t;  // storage type of a bf16 element in memory

constexpr int kRowPad = 128;     // rows of every packed operand are padded to this
// host side: the launchers record which kernel template a forward (0) / gradient-product (1) launch of this thread went to
// (crossclr_last_kernel, include/crossclr.h; defined in crossclr_api.cpp)
void note_kernel(int which, const char* name);
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// ---------------------------------------------------------------------------------------------
// geometry of one launch (passed by value)
// ---------------------------------------------------------------------------------------------
struct Geo {
    int b;          // valid rows per modality per rank
    int bpad;       // padded rows per modality
    int D, Dpad;
    int col_ranks;  // rank segments in the column operand
    int col_rank0;  // rank id of column segment 0
    int row_rank;   // rank id owning the row operand
    int skip_rank;  // column rank to skip (-1: none)
    int col_wrap;   // 0: column segment i is rank col_rank0+i at offset i; W > 0: the column operand is the WHOLE gathered
                    // array of W ranks and segment i is rank (col_rank0+i) mod W at offset of that rank
    float c_inter;  // log2(e)/tau
    float c_intra;  // negative_weight*log2(e)/tau
    float m2;       // soft-max shift, log2 domain (0 when row_shift is set)
    int row_shift;  // 1: max |logit| is beyond any fixed shift in fp32 (temperature < ~0.008): the generic kernels take a per-row
                    // shift (the row maximum, found by a first pass) like the reference's float64 soft-max does (loss.py:60)
};

// ---------------------------------------------------------------------------------------------
// portability layer: the handful of gfx950 builtins the kernels use
// ---------------------------------------------------------------------------------------------
#ifndef CROSSCLR_EMU
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// 16x16x32: A[i][k] in lane i + 16*(k/8), element k%8; B[k][j] in lane j + 16*(k/8); C[i][j] in lane j + 16*(i/4), reg i%4
__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// 16-lane-group transpose read: lane i of a group receives element (i&3) of the 8-byte pieces
// addressed by lanes 4j+(i>>2), j = 0..3  (i.e. column i of the 4x16 b16 matrix whose row j is
// made of the four pieces addressed by lanes 4j..4j+3).
__device__ __forceinline__ s16x4 lds_read_tr16_b64(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float wave_xor_f32(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ double wave_xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }
// value of lane (l ^ MASK) for MASK in {1, 2, 7, 15, 16}: DPP quad_perm / row_half_mirror / row_mirror
// (one VALU slot, no LDS) and a ds_swizzle for the cross-row 16
template <int MASK> __device__ __forceinline__ float lane_xor(float v) {
    const int b = __builtin_bit_cast(int, v);
    int r;
    if (MASK == 1) r = __builtin_amdgcn_mov_dpp(b, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if (MASK == 2) r = __builtin_amdgcn_mov_dpp(b, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if (MASK == 7) r = __builtin_amdgcn_mov_dpp(b, 0x141, 0xF, 0xF, true);  // row_half_mirror: i <-> 7-i
    else if (MASK == 15) r = __builtin_amdgcn_mov_dpp(b, 0x140, 0xF, 0xF, true); // row_mirror: i <-> 15-i
    else r = __builtin_amdgcn_ds_swizzle(b, 0x401F);                             // bit mode: xor 16 within 32 lanes
    return __builtin_bit_cast(float, r);
}
#define CROSSCLR_SHARED __shared__
// A few values handed from thread block to thread block inside one launch WITHOUT an agent-scope fence (whose release writes back every dirty
// line of the XCD's L2 and whose acquire drops the CU's L1: ~5 us in fwd_finish_kernel, profiles/r05k_kernel_stats.csv): the producer's store is
// written through to the coherence point (sc1) and COMPLETE before its ticket atomic is issued; the consumer, having seen every ticket, reads
// with sc1 loads (past its L1; MI355X_MICROARCH.md, inter-workgroup visibility: "16 B sc1 stores AND sc1 loads").
// This ordering is a property of the TARGET, not of the HIP memory model (relaxed atomics give no happens-before): it rests on gfx942 / gfx950
// writing agent-scope atomic stores through to the coherence point and on vmcnt counting stores (gfx10+ count them in vscnt).  The device pass
// of any other target gets the portable form: a release fence in front of the ticket, an acquire fence behind it.
#if !defined(__HIP_DEVICE_COMPILE__) || defined(__gfx950__) || defined(__gfx942__)
__device__ __forceinline__ void handoff_store_f64(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void handoff_stores_complete() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ double handoff_load_f64(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
__device__ __forceinline__ void handoff_store_f64(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void handoff_stores_complete() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__device__ __forceinline__ double handoff_load_f64(const double* p) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

#endif

// ---------------------------------------------------------------------------------------------
// Work list of the persistent forward, in COST UNITS (shared by the forward kernels, the plan and fwd_finish_kernel).
// Items = (row block, column tile) pairs, row-block major; a thread block owns the items whose first unit falls into its
// contiguous range of `per` units.  Rectangular lists (kind 2 / 3): one unit per item.  Symmetric lists (kind 1): the tpr tiles
// of a row block's diagonal block take the masked, un-overlapped epilogue and entering a row block means loading its fragments.
// Costs measured on fast_fwd_pair_kernel by regressing the thread blocks' main-loop times of one launch on their ranges' composition
// (tools/fwd_balance.py, profiles/r05u_fwd_balance.txt): a masked tile = 1.50 plain tiles, a row-block start = 1.82 -- in half-tile
// units: plain 2, masked 3, item 0 (start + masked) 7.  (Round 4's kernel, whose plain tiles were slower: 1 / 2 / 3.)
// ---------------------------------------------------------------------------------------------
#ifndef CROSSCLR_FWD_COST_PLAIN
#define CROSSCLR_FWD_COST_PLAIN 2
#define CROSSCLR_FWD_COST_MASKED 3
#define CROSSCLR_FWD_COST_FIRST 7
#endif
constexpr int kFwdCostPlain = CROSSCLR_FWD_COST_PLAIN, kFwdCostMasked = CROSSCLR_FWD_COST_MASKED, kFwdCostFirst = CROSSCLR_FWD_COST_FIRST;
__host__ __device__ __forceinline__ int fwdw_items(int kind, int tpr, int NT, int rb) { return kind == 1 ? NT - tpr * rb : NT; }
__host__ __device__ __forceinline__ int fwdw_item_prefix(int kind, int tpr, int NT, int rb) {      // items before row block rb
    return kind == 1 ? rb * NT - (tpr / 2) * rb * (rb - 1) : rb * NT;
}
__host__ __device__ __forceinline__ int fwdw_unit_prefix(int kind, int tpr, int NT, int rb) {      // units before row block rb
    // every row block of a symmetric list has its tpr diagonal tiles (NT = tpr * NB): first + (tpr - 1) masked + plain for the rest
    return kind == 1 ? rb * (kFwdCostFirst + (tpr - 1) * kFwdCostMasked - tpr * kFwdCostPlain) + kFwdCostPlain * (rb * NT - (tpr / 2) * rb * (rb - 1))
                     : rb * NT;
}
__host__ __device__ __forceinline__ int fwdw_unit_of_item(int kind, int tpr, int j) {              // first unit of item j of its row block
    if (kind != 1) return j;
    if (j == 0) return 0;
    return j < tpr ? kFwdCostFirst + (j - 1) * kFwdCostMasked : kFwdCostFirst + (tpr - 1) * kFwdCostMasked + (j - tpr) * kFwdCostPlain;
}
__host__ __device__ __forceinline__ int fwdw_item_at_unit(int kind, int tpr, int u) {              // first item whose first unit is >= u
    if (kind != 1) return u;
    if (u <= 0) return 0;
    const int base = kFwdCostFirst + (tpr - 1) * kFwdCostMasked;      // first unit of item tpr
    if (u > base - kFwdCostMasked) {                                  // beyond the first unit of item tpr - 1
        const int j = tpr + (u - base + kFwdCostPlain - 1) / kFwdCostPlain;
        return j > tpr ? j : tpr;
    }
    const int j = 1 + (u - kFwdCostFirst + kFwdCostMasked - 1) / kFwdCostMasked;
    return u <= kFwdCostFirst ? 1 : j;
}
__host__ __device__ __forceinline__ int fwdw_first_block(int kind, int tpr, int NT, int per, int rb) {
    return fwdw_unit_prefix(kind, tpr, NT, rb) / per;
}
__host__ __device__ __forceinline__ int fwdw_last_block(int kind, int tpr, int NT, int per, int rb) {
    return (fwdw_unit_prefix(kind, tpr, NT, rb) + fwdw_unit_of_item(kind, tpr, fwdw_items(kind, tpr, NT, rb) - 1)) / per;
}

struct FwdWork {
    int kind;   // 1 symmetric, 2 rectangular, 3 rectangular + column sums (0 in the workspace header = dense slots)
    int tpr;    // 32-column tiles per row block
    int NB;     // row blocks
    int NT;     // symmetric: column tiles of the operand; rectangular: usable column tiles
    int per;    // cost units per thread block
    int nblk;   // thread blocks
    int total;  // items
};
__host__ __device__ __forceinline__ int fwd_prefix(const FwdWork& w, int rb) { return fwdw_item_prefix(w.kind, w.tpr, w.NT, rb); }
__host__ __device__ __forceinline__ int fwd_first_block(const FwdWork& w, int rb) { return fwdw_first_block(w.kind, w.tpr, w.NT, w.per, rb); }
__host__ __device__ __forceinline__ int fwd_last_block(const FwdWork& w, int rb) { return fwdw_last_block(w.kind, w.tpr, w.NT, w.per, rb); }
// first item of thread block b (= the end of block b - 1's range; `total` past the end of the list)
__host__ __device__ __forceinline__ int fwd_block_begin(const FwdWork& w, int b) {
    const int P = b * w.per;
    if (P >= fwdw_unit_prefix(w.kind, w.tpr, w.NT, w.NB)) return w.total;
    int rb = 0;
    while (fwdw_unit_prefix(w.kind, w.tpr, w.NT, rb + 1) <= P) ++rb;
    const int j = fwdw_item_at_unit(w.kind, w.tpr, P - fwdw_unit_prefix(w.kind, w.tpr, w.NT, rb));
    return j >= fwdw_items(w.kind, w.tpr, w.NT, rb) ? fwd_prefix(w, rb + 1) : fwd_prefix(w, rb) + j;
}
static inline FwdWork fwd_make_work(int kind, int bpad, int usable_col_tiles, int max_blocks, int tpr) {
    FwdWork w;
    w.kind = kind;
    w.tpr = tpr;
    w.NB = 2 * bpad / (32 * tpr);
    w.NT = kind == 1 ? 2 * bpad / 32 : usable_col_tiles;
    w.total = fwd_prefix(w, w.NB);
    const int units = fwdw_unit_prefix(kind, tpr, w.NT, w.NB);
    // every thread block between a row block's first and last one must hold at least one of its items (its slot is summed):
    // a range is never shorter than the longest item (item 0 of a symmetric list)
    const int min_per = kind == 1 ? kFwdCostFirst : 2;
    int nb = units / min_per;
    if (nb > max_blocks) nb = max_blocks;
    if (nb < 1) nb = 1;
    w.per = (units + nb - 1) / nb;
    if (w.per < min_per) w.per = min_per;
    w.nblk = (units + w.per - 1) / w.per;
    return w;
}
static inline int fwd_max_slots(const FwdWork& w) {
    int m = 1;
    for (int rb = 0; rb < w.NB; ++rb) {
        const int n = fwd_last_block(w, rb) - fwd_first_block(w, rb) + 1;
        if (n > m) m = n;
    }
    return m;
}

// XCD-aware placement of the forward's work ranges.  The launch keeps one persistent thread block per CU and block b runs on XCD b % 8
// (observed, relied on for speed only), each XCD with its own 4-MiB L2.  Handing range c to block c spreads the ranges that stream the SAME
// column tiles over all eight XCDs (L2 hit rate 0.36 at B = 8192: every tile is fetched from the Infinity Cache by ~6 of 8 XCDs).  The
// permutation below sorts the ranges by the column tile they start at and deals consecutive runs of the sorted list to one XCD each: the
// thread blocks of an XCD then walk neighbouring column ranges at the same pace and a tile is fetched once per XCD.  A block takes range
// v[blockIdx.x]; slots, column sums and the stash are indexed by the RANGE, so results are bit-identical to the identity placement.
struct FwdPerm { unsigned short v[256]; };
static inline void fwd_make_perm(const FwdWork& w, bool xcd_aware, FwdPerm* out) {
    const int n = w.nblk < 256 ? w.nblk : 256;
    for (int b = 0; b < 256; ++b) out->v[b] = (unsigned short)b;
    if (!xcd_aware || w.nblk > 256 || n < 16) return;
    int key[256], order[256];
    for (int c = 0; c < n; ++c) {
        const int w0 = fwd_block_begin(w, c);
        order[c] = c;
        if (w0 >= w.total) { key[c] = 1 << 30; continue; }
        int rb = 0;
        while (fwd_prefix(w, rb + 1) <= w0) ++rb;
        const int j = w0 - fwd_prefix(w, rb);
        key[c] = w.kind == 1 ? w.tpr * rb + j : j;          // the column tile the range starts at
    }
    for (int i = 1; i < n; ++i) {                            // insertion sort by (key, range): stable, n <= 256, once per plan
        const int c = order[i];
        int k = i - 1;
        while (k >= 0 && (key[order[k]] > key[c] || (key[order[k]] == key[c] && order[k] > c))) { order[k + 1] = order[k]; --k; }
        order[k + 1] = c;
    }
    int off = 0;
    for (int x = 0; x < 8; ++x) {
        const int cnt = (n - x + 7) / 8;                     // blocks b < n with b % 8 == x
        for (int i = 0; i < cnt; ++i) out->v[x + 8 * i] = (unsigned short)order[off + i];
        off += cnt;
    }
}

// Saved bf16 exponentials of a symmetric local block (crossclr_kernels_fast.h, "stash"): one 2-KiB record per 32 x 32 tile (r32, t) with
// t >= tpr * (r32 / tpr) -- the upper triangle at the granularity of the forward's row blocks (tpr 32-row groups each), row group by row group.
__host__ __device__ __forceinline__ size_t stash_tile_index(int tpr, int NT, int r32, int t) {
    const size_t rb = (size_t)(r32 / tpr), w = (size_t)(r32 % tpr);
    const size_t before = (size_t)tpr * (rb * NT - (size_t)(tpr / 2) * rb * (rb - 1));   // tiles of row blocks < rb
    return before + w * ((size_t)NT - tpr * rb) + ((size_t)t - tpr * rb);
}
static inline size_t stash_tiles_total(int tpr, int NT) { return stash_tile_index(tpr, NT, NT, NT); }

// Timeline instrumentation of the two pipelined kernels (variant builds only: tools/build_variant.py NAME -DCROSSCLR_TIMING;
// read back with crossclr_debug_timing, tools/timeline.py): thread 0 of every block stamps the constant 100 MHz counter
// (s_memrealtime) and the shader-clock counter (s_memtime) at a few marks -- block start / first tile ready / main loop done /
// exit -- which shows the dispatch ramp, the tail, and the clock the kernel actually ran at.
#if defined(CROSSCLR_TIMING) && !defined(CROSSCLR_EMU)
__device__ unsigned long long g_timing[1024 * 8];
__device__ __forceinline__ void timing_mark(int slot) {
    if (threadIdx.x == 0 && blockIdx.x + gridDim.x * blockIdx.y < 1024) {
        unsigned long long* t = g_timing + (blockIdx.x + gridDim.x * blockIdx.y) * 8;
        t[slot] = __builtin_amdgcn_s_memrealtime();
        if (slot == 0) { t[4] = __builtin_amdgcn_s_memtime(); t[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); t[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20); }
        if (slot == 3) t[5] = __builtin_amdgcn_s_memtime();
    }
}
#else
__device__ __forceinline__ void timing_mark(int) {}
#endif

__device__ __forceinline__ bf16_t f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;  // round-to-nearest-even (v_cvt_pk_bf16_f32 on gfx950)
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float bf16_bits_to_f32(bf16_t h) {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
}

// C/D fragment of a 32x32 MFMA: lane l, register r holds C[row][col] with
__device__ __forceinline__ int frag_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
// col = l & 31, half = l >> 5.

// ---------------------------------------------------------------------------------------------
// operand traits: how a 128-byte K-chunk of an LDS tile row turns into MFMA operands
// ---------------------------------------------------------------------------------------------
// LDS "K-tile": R rows x 128 bytes; 16-byte chunk c of row r lives at r*128 + ((c ^ ((r>>1)&7)) << 4).
// ds_read_b128 is serviced per 16-lane group over a 256-byte (64-bank) row, i.e. TWO tile rows:
// a group reading chunk c of 16 rows that are distinct mod 16 hits 16 distinct (row parity, slot)
// pairs -> conflict-free (the linear layout would be 8-way).
__device__ __forceinline__ int ktile_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <typename T> struct Operand;

template <> struct Operand<float> {
    typedef float elem;
    static constexpr int kChunkElems = 32;  // 128 B
    typedef f32x4 frag;                     // 4 consecutive k values of one row
    // one step s (0..3) of a chunk consumes 16-byte piece 2*s+half of each row and issues 4 MFMAs
    static __device__ __forceinline__ frag load(const unsigned char* tile, int row, int s, int half) {
        return *reinterpret_cast<const f32x4*>(tile + ktile_off(row, 2 * s + half));
    }
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = mfma_32x32x2_f32(a[j], b[j], c);
        return c;
    }
};

template <> struct Operand<bf16_t> {
    typedef bf16_t elem;
    static constexpr int kChunkElems = 64;  // 128 B
    typedef bf16x8 frag;                    // 8 consecutive k values of one row
    static __device__ __forceinline__ frag load(const unsigned char* tile, int row, int s, int half) {
        return *reinterpret_cast<const bf16x8*>(tile + ktile_off(row, 2 * s + half));
    }
    static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return mfma_32x32x16_bf16(a, b, c); }
};

// Stage R rows x 128 bytes (byte column kb of rows r0.. of a row-major array with row pitch
// `pitch` bytes) into registers, then into a swizzled K-tile.  NT threads cooperate.
template <int R, int NT> struct KTileStage {
    static constexpr int kPieces = R * 8 / NT;  // 16-byte pieces per thread
    u32x4 v[kPieces];
    __device__ __forceinline__ void fetch(const unsigned char* base, size_t pitch, int kb, int tid) {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            int id = tid + i * NT;
            int row = id >> 3, c = id & 7;
            v[i] = *reinterpret_cast<const u32x4*>(base + (size_t)row * pitch + kb + c * 16);
        }
    }
    __device__ __forceinline__ void commit(unsigned char* tile, int tid) const {
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            int id = tid + i * NT;
            int row = id >> 3, c = id & 7;
            *reinterpret_cast<u32x4*>(tile + ktile_off(row, c)) = v[i];
        }
    }
};

// ---------------------------------------------------------------------------------------------
// column-tile bookkeeping shared by forward and backward
// ---------------------------------------------------------------------------------------------
struct ColTile {
    int rank;     // global rank of the segment
    int mod;      // modality of the columns
    int in_mod0;  // index of the first column inside its modality block
    size_t row0;  // first row of the tile inside the column operand
    size_t stat0; // first index of the tile inside column statistics arrays
};
// tile t of width W over a column operand Xcols[col_ranks][2][bpad][Dpad]
__device__ __forceinline__ ColTile col_tile(const Geo& g, int t, int W) {
    int per_rank = 2 * g.bpad / W;
    int seg = t / per_rank, in_seg = t - seg * per_rank;
    int per_mod = g.bpad / W;
    ColTile ct;
    ct.rank = g.col_rank0 + seg;
    int seg_mem = seg;
    if (g.col_wrap > 0) {
        if (ct.rank >= g.col_wrap) ct.rank -= g.col_wrap;
        seg_mem = ct.rank;
    }
    ct.mod = in_seg / per_mod;
    ct.in_mod0 = (in_seg - ct.mod * per_mod) * W;
    ct.row0 = (size_t)seg_mem * 2 * g.bpad + (size_t)in_seg * W;
    ct.stat0 = ct.row0;
    return ct;
}

// ---------------------------------------------------------------------------------------------
// Sum e[0..15] over the 32 lanes of each wave half by recursive halving: after step k a lane keeps
// only the half of its values selected by one bit of its lane id and adds the partner's copy of that
// half (16 adds + 30 selects instead of 80 adds for 16 independent butterflies; fp32, fixed order).
// Every lane ends up with the total of element  8*b3 + 4*b2 + 2*b1 + b0  (b_k = bit k of l31).
// Partners: l^15, l^7 (DPP row_mirror / row_half_mirror: they also flip the lower bits, which have
// not been used as selectors yet at that point), l^2, l^1 (DPP quad_perm), l^16 (ds_swizzle).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float halving_sum16(const float (&e)[16], int l31) {
    float k8[8], k4[4], k2[2];
    {
        const bool up = (l31 >> 3) & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) k8[i] = (up ? e[8 + i] : e[i]) + lane_xor<15>(up ? e[i] : e[8 + i]);
    }
    {
        const bool up = (l31 >> 2) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) k4[i] = (up ? k8[4 + i] : k8[i]) + lane_xor<7>(up ? k8[i] : k8[4 + i]);
    }
    {
        const bool up = (l31 >> 1) & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) k2[i] = (up ? k4[2 + i] : k4[i]) + lane_xor<2>(up ? k4[i] : k4[2 + i]);
    }
    const bool up = l31 & 1;
    const float k1 = (up ? k2[1] : k2[0]) + lane_xor<1>(up ? k2[0] : k2[1]);
    return k1 + lane_xor<16>(k1);
}
// the same exchange pattern for maxima (row maxima of the two-pass soft-max, mirrored tiles)
__device__ __forceinline__ float halving_max16(const float (&e)[16], int l31) {
    float k8[8], k4[4], k2[2];
    {
        const bool up = (l31 >> 3) & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) k8[i] = fmaxf(up ? e[8 + i] : e[i], lane_xor<15>(up ? e[i] : e[8 + i]));
    }
    {
        const bool up = (l31 >> 2) & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) k4[i] = fmaxf(up ? k8[4 + i] : k8[i], lane_xor<7>(up ? k8[i] : k8[4 + i]));
    }
    {
        const bool up = (l31 >> 1) & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) k2[i] = fmaxf(up ? k4[2 + i] : k4[i], lane_xor<2>(up ? k4[i] : k4[2 + i]));
    }
    const bool up = l31 & 1;
    const float k1 = fmaxf(up ? k2[1] : k2[0], lane_xor<1>(up ? k2[0] : k2[1]));
    return fmaxf(k1, lane_xor<16>(k1));
}
__device__ __forceinline__ int halving_elem16(int l31) {
    return 8 * ((l31 >> 3) & 1) + 4 * ((l31 >> 2) & 1) + 2 * ((l31 >> 1) & 1) + (l31 & 1);
}

}  // namespace crossclr
