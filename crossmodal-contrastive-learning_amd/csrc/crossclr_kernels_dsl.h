// crossclr_kernels_dsl.h -- backward from the saved exponentials, "D-slice" decomposition (Dpad <= 512; two column parts up to 1024).
//
// G[p][:] = sum_q W[p][q] X[q][:] with W = E (omega_p/Z_p + omega_q/Z_q) formed from the forward's bf16 stash (autograd of reference
// trainer/loss.py:83-112): 8 B^2 D flop executed = algorithmic, no similarity recompute.  128-row blocks x column slices (grid =
// (2 bpad / 128, slices[, 2 column parts])), 4 waves, ONE wave per SIMD (256 accumulator registers + ~250 others).
//
//   who owns what     a wave = all 128 rows x a Dpad/4 column slice (the square 128 x 128 wave tile of a GEMM: 4 A + 4 B fragments per 16
//                     MFMAs).  Each wave reads only ITS quarter of the 32 x Dpad column tile (16 transpose reads per tile); W -- 128 x 32
//                     bf16 = 8 KiB per tile -- is shared through LDS: wave w turns the saved exponentials of row group w into W (56 VALU per
//                     tile), writes them with two ds_write_b128 into a double-buffered W slot, and after the tile's one barrier every wave
//                     reads all four groups' A fragments (8 ds_read_b128).  ~36 LDS instructions per wave and tile.
//                     (Round 2's kernel gave a wave 32 rows x all Dpad columns: W private, but every wave read the WHOLE column tile
//                     through 64 transpose reads -- 74 LDS instructions, 330 instructions per wave and tile against 215 now.)
//   mirrored tiles    (column tile left of the block's own 256-row forward block: stash tile (t, r32) holds E^T)  The wave weighs the tile in
//                     its STORED orientation (the lane's row is a column q of ours: row statistic from the tile's vector, column statistics
//                     from the block's own rows) and writes W^T with a chunk permutation that makes the four rows a transpose read
//                     gathers sit in one 256-byte line; the A fragments are then 16 ds_read_b64_tr_b16.  Direct and mirrored tiles are two
//                     contiguous runs of the slice, so the loop is split in three bodies (M->M, M->D, D->D): no per-tile selects.
//   software pipeline iteration t runs the 4 DI MFMAs of k-step 1 of tile t-1 FIRST (their fragments were read before the barrier) and only
//                     then k-step 0 of tile t: the LDS reads that follow a barrier land behind 4 DI MFMAs of independent work instead of
//                     stalling the pipe.  k-step 1 of tile t is read in the second half (complete before the next barrier) and carried
//                     across the back edge.  Every LDS read is asm; the only waits are lgkmcnt(0) in the last slot of the first half
//                     (everything it covers was issued >= 5 MFMAs earlier) and at the end of the iteration.
//   VMEM              per iteration and wave NXO pieces of X(t+2) and then 2 pieces + statistics of the stash tile t+1+PE, dealt evenly over the
//                     MFMA slots.  X before E: vmcnt retires in order, so the wait that proves X(t+1) landed covers everything older --
//                     with E first every iteration waited for an HBM access it had issued one tile earlier.
//                     s_waitcnt vmcnt(NXO + 2 NEO) per tile (NXO + NEO in the first iteration, whose X(t+1) is the prologue's last operation).
// LDS (Dpad = 512): X ring 3 x 32 KiB | W 2 x 8 KiB | E ring (PE+1) x 4 x 2 KiB | statistics rings | the block's own statistics = 150 KiB.
// Measured and ablated in DESIGN.md section 3.
#pragma once

namespace crossclr {

#ifndef CROSSCLR_DSL_PE
#define CROSSCLR_DSL_PE 3      // tiles between the DMA of a saved-exponential tile and its use (HBM latency)
#endif
#ifndef CROSSCLR_DSL_STAGGER
#define CROSSCLR_DSL_STAGGER 0  // 1: four copies of the loop, one per wave, whose VMEM instructions sit in different MFMA slots (measured: the
                               // addresser's FIFO-full events drop 5x, the time does not move; 4x the code and compile time) -- off
#endif
#ifndef CROSSCLR_DABL
#define CROSSCLR_DABL 0        // timing ablations (WRONG results): bit0 no E DMA, bit1 no X DMA, bit2 no weight VALU, bit3 no W write,
                               // bit4 no B reads, bit5 no barrier, bit6 no A reads, bit7 no MFMA, bit8 E from a 2-MiB window, bit9 X from 8 tiles,
                               // bit10 no mid-iteration LDS wait, bit11 no closing LDS wait, bit12 no closing VMEM wait
#endif

#ifndef CROSSCLR_EMU
template <int OFF> __device__ __forceinline__ unsigned lds_read_b32_async(unsigned addr) {
    unsigned r;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
// orders every later use of x behind the preceding asm wait (emits nothing)
template <typename T> __device__ __forceinline__ void after_wait(T& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void wait_lgkm_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// materialise a scalar value HERE (hipcc otherwise sinks address arithmetic down to its first use, across sched_barriers)
template <typename T> __device__ __forceinline__ void pin_s(T& x) {
    static_assert(sizeof(T) == 4, "32-bit scalars");
    x = (T)__builtin_amdgcn_readfirstlane((int)x);     // (folds away when the value already lives in an SGPR)
    asm volatile("" : "+s"(x));
}
template <typename T> __device__ __forceinline__ void pin_v(T& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void mfma_acc(f32x16& acc, bf16x8 a, bf16x8 b) {
    if (CROSSCLR_DABL & 128) { asm volatile("" : "+a"(acc) : "v"(a), "v"(b)); return; }
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// XF: an MFMA B fragment straight from global memory into VGPRs (buffer_load_dwordx4 ... offen: wave-uniform descriptor in SGPRs,
// per-lane byte offset, scalar byte offset, 12-bit immediate).  Asm for the same reason as the LDS reads: hipcc's wait-count pass
// drains the LDS-DMA ring (vmcnt(0)) in front of the first use of an ordinary load; this one it does not see -- its completion is
// counted by hand (s_waitcnt vmcnt(N): VMEM returns in order) and after_wait() orders every use behind that wait.
struct RawRsrc { u32x4 v; };
__device__ __forceinline__ RawRsrc make_raw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)(uintptr_t)base;
    RawRsrc r;
    r.v[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.v[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.v[2] = (unsigned)__builtin_amdgcn_readfirstlane((int)bytes);
    r.v[3] = 0x00020000u;      // (the flags word __builtin_amdgcn_make_buffer_rsrc is given everywhere else in this library)
    return r;
}
template <int OFF> __device__ __forceinline__ u32x4 buf_load_b128_async(const RawRsrc& rs, unsigned voff, unsigned soff) {
    static_assert(OFF >= 0 && OFF < 4096, "12-bit immediate");
    u32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(r) : "v"(voff), "s"(rs.v), "s"(soff), "n"(OFF));
    return r;
}
#else
struct RawRsrc { const unsigned char* base; unsigned bytes; };
__device__ __forceinline__ RawRsrc make_raw_rsrc(const void* base, unsigned bytes) { return RawRsrc{static_cast<const unsigned char*>(base), bytes}; }
template <int OFF> __device__ __forceinline__ u32x4 buf_load_b128_async(const RawRsrc& rs, unsigned voff, unsigned soff) {
    u32x4 r = {0u, 0u, 0u, 0u};
    const size_t o = (size_t)voff + soff + OFF;
    if (o + 16 <= rs.bytes) memcpy(&r, rs.base + o, 16);
    return r;
}
template <int OFF> __device__ __forceinline__ unsigned lds_read_b32_async(unsigned long addr) {
    return *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(addr) + OFF);
}
template <typename T> __device__ __forceinline__ void after_wait(T&) {}
template <typename T> __device__ __forceinline__ void pin_s(T&) {}
template <typename T> __device__ __forceinline__ void pin_v(T&) {}
__device__ __forceinline__ void wait_lgkm_all() {}
__device__ __forceinline__ void mfma_acc(f32x16& acc, bf16x8 a, bf16x8 b) { acc = mfma_32x32x16_bf16(a, b, acc); }
#endif

// One wave's share of the block (WV = its index, a compile-time constant: the four waves run four copies of the loop whose
// VMEM instructions sit in DIFFERENT MFMA slots -- see the schedule below -- and whose LDS addresses are immediates).
template <int DK, bool SW, int MODE, int XP, int TPRF, int WV, bool XF = false>
__device__ __forceinline__ void dsl_wave(unsigned char* lds, const bf16_t* cols, const unsigned char* stash, const Geo& g, const float* rz, const float* wrz,
        const float* rz_cols, const float* wrz_cols, float* gbuf, int accumulate, int tiles_per_slice, const float* ks, const float* kc) {
    constexpr int RB = DK * 32;            // bytes per row of the (part of the) operand a block multiplies
    constexpr int QT = 32;
    constexpr int TILE = QT * RB;
    constexpr int TPR = TPRF;
    constexpr int RBG = XP * RB;           // bytes per operand row in memory
    constexpr int DI = DK / 8;             // 32-wide output fragments per wave (its column slice)
    constexpr int H = 4 * DI;              // MFMAs per k-step: 4 row groups x DI fragments
    constexpr int NSX = XF ? 0 : 3;        // column-tile ring (XF: the column tile never touches LDS)
    constexpr int PE = XF ? 2 : CROSSCLR_DSL_PE;
    constexpr int NSE = PE + 1;            // saved-exponential + statistics rings (private to a wave)
    constexpr int ESTG = 4 * 2048;         // one stage of the E ring: [4 waves][2 KiB]
    constexpr int SSTG = 4 * 256;          // one stage of a statistics ring: [4 waves][64 floats] (the tile's 32 + 32 spare)
    constexpr int NXO = DK / 4;            // VMEM operations per wave and tile: column-tile pieces (XF: 2 DI fragment loads -- the same number) ...
    constexpr int NEO = 3 + (SW ? 1 : 0);  // ... saved-exponential pieces + statistics
    constexpr int WBUF = 4 * 2048;         // one W slot: [4 row groups][2 KiB]
    constexpr int W0 = NSX * TILE, E0 = W0 + 2 * WBUF, S0 = E0 + NSE * ESTG, K0 = S0 + NSE * SSTG;
    constexpr int O0 = K0 + (SW ? NSE * SSTG : 0);      // the block's own statistics: rz[128] | wrz[128] | k[128]
    static_assert(DK % 8 == 0 && DK >= 8 && DK <= 32, "Dpad (per part) in {128, 256, 384, 512}");
    static_assert(XF || PE >= 3, "E / statistics of tile t+2 must be older than the pieces of X(t+1)");
    static_assert(!XF || (MODE == 0 && WV < 0 && NXO == 2 * DI), "XF: the local symmetric block, one copy of the loop");
    static_assert(O0 + 3 * 512 <= 160 * 1024, "LDS budget");
    // XF ("fragment-major" column operand, crossclr_normalize_xf): `cols` is not the row-major packed operand but XF[tile u = 32 stacked
    // rows][dt = Dpad/32 column fragments][k-step][lane][8 bf16] -- 1 KiB per (u, dt, k-step), inside it lane (n = lane & 31, kg = lane >> 5)
    // holds X[32 u + 16 ks + 8 kg + 0..7][32 dt + n]: exactly the B fragment of v_mfma_f32_32x32x16_bf16 for G += W X.  A wave loads the
    // 2 DI fragments of ITS column slice with 2 DI fully coalesced buffer_load_dwordx4 straight into VGPRs: no LDS-DMA of the column
    // tile (NXO pieces per wave and tile), no transpose reads (2 DI pairs), 96 KiB of LDS less, and the k-step skew keeps working on
    // registers.  Two register sets (tile parity): set par holds tile t, set par^1 tile t-1 whose k-step-1 half is multiplied in the
    // first half of iteration t (fragment-major MFMA order: fragment di is dead after its 4 MFMAs) while tile t+1 lands in it --
    // k-step-0 fragments from slot 0, the k-step-1 fragment di from slot 4 (di + 1).  Every load completes inside the iteration that
    // issues it (s_waitcnt vmcnt(NEO) at its end: only the saved-exponential pieces issued behind the last load stay in flight), so no
    // asm-loaded register is ever live across a branch; the loop is unrolled by two for the static register naming.
    // MODE 0: the local symmetric block; 1 (RECT): this rank's rows x other ranks' columns, rectangular stash; 2 (TR): the TRANSPOSE of one
    // rectangular block -- output rows = the partner rank's rows, contraction over THIS rank's rows, every tile read mirrored from
    // the stash of block (this rank x partner): what the partner would otherwise recompute (crossclr_backward_rect_saved_t).  The host
    // passes TR launches the local operand / statistics as "columns" and the partner's statistics as "rows"; g.col_ranks = rank
    // segments per stash row, g.skip_rank = the partner's segment inside it.
    constexpr bool RECT = MODE == 1, TR = MODE == 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = WV >= 0 ? WV : uniform(tid >> 6);        // WV = -1: one copy of the code for all four waves
    constexpr int WVS = WV >= 0 ? WV : 0;                     // the wave's place in the VMEM schedule
    const int half = lane >> 5, l31 = lane & 31;
    const int row0b = blockIdx.x * 128;
    const int row0w = row0b + 32 * wave;
    const int r32 = uniform(row0w >> 5);
    const int per_rank = 2 * g.bpad / QT, per_mod = g.bpad / QT;
    const int skip_seg = (RECT && g.col_wrap == 0 && g.skip_rank >= g.col_rank0 && g.skip_rank < g.col_rank0 + g.col_ranks) ? g.skip_rank - g.col_rank0 : -1;
    const int NT = RECT ? (g.col_ranks - (skip_seg >= 0 ? 1 : 0)) * per_rank : per_rank;
    const int rmod = (2 * r32 >= per_rank) ? 1 : 0;
    const int rb0 = (r32 / TPR) * TPR;           // first tile the forward evaluated for this block's rows
    const int part = XP > 1 ? (int)blockIdx.z : 0;

    const float rzp_inter = rz[row0w + l31];
    const float rzp_intra = wrz[row0w + l31];
    const float kp = SW ? ks[row0w + l31] : 1.f;
    wait_loads_visible();   // (the compiler then never waits for these loads inside the DMA stream)
    // transpose-read roles: in a 16-lane group lane 4j+c addresses row j, 8-byte piece c
    const int grp = lane >> 4, i16 = lane & 15, jrow = i16 >> 2, piece = i16 & 3, dsub = grp & 1;
    // column tile: B fragment (k-step tp, output fragment dt = DI*wave + di) = rows 16tp + 4half + jrow (+8), columns 32dt + 16dsub + 4piece..
    int bo[DI][2];
#pragma unroll
    for (int di = 0; di < DI; ++di) {
        const int dt = DI * wave + di;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            bo[di][u] = (4 * half + jrow) * RB + 256 * (dt >> 2) + 64 * ((dt & 3) ^ jrow) +
                        16 * ((2 * dsub + (piece >> 1)) ^ (2 * u + half)) + 8 * (piece & 1);
    }
    // W slot, direct image: group pi at pi*2048, lane-linear fragment (k-step th at +1024*th).  Mirrored image: 16-byte chunk (th_s, hf, rho)
    // of the stored tile [row rho = a column q of ours, chunk = 8 of OUR rows] at slot 16*(rho>>2) + (rho&3) + 4*hf + 8*th_s; the reader
    // (lane (half, g1 = dsub, jj = jrow, c = piece)) addresses th*1024 + u*512 + [half*256 + (jj + 4(c&1) + 8 g1)*16 + 8(c>>1)]
    const int wr_dir = wave * 2048 + lane * 16;                                           // + 1024*th
    const int wr_mir = wave * 2048 + 16 * (16 * (l31 >> 2) + (l31 & 3) + 4 * half);       // + 128*th
    const int rd_dir = lane * 16;                                                         // + pi*2048 + th*1024
    const int rd_mir = half * 256 + (jrow + 4 * (piece & 1) + 8 * dsub) * 16 + 8 * (piece >> 1);   // + pi*2048 + th*1024 (+512)

    const int col_segs = !RECT ? 1 : (g.col_wrap > 0 ? g.col_wrap : g.col_ranks);
    const BufRsrc rs_x = make_rsrc(cols, (unsigned)((size_t)col_segs * 2 * g.bpad * RBG));
    unsigned voffx[NXO];
#pragma unroll
    for (int k = 0; k < NXO; ++k) {
        const int L = (wave + 4 * k) * 1024 + lane * 16;
        const int row = L / RB, slot = (L - row * RB) >> 4;
        voffx[k] = (unsigned)(row * RBG + part * RB + (swz_slot(slot, row) << 4));
    }
    // XF: fragment (dt = 4 DI part + DI wave + di, k-step ks) of tile u at u * QT * RBG + (2 dt + ks) * 1024 + 16 lane
    const RawRsrc rs_xf = make_raw_rsrc(cols, (unsigned)((size_t)2 * g.bpad * RBG));
    const unsigned xfv0 = (unsigned)((4 * DI * part + DI * wave) * 2048 + lane * 16), xfv1 = xfv0 + 4096u;
    u32x4 BS[2][DI][2];          // [tile parity][fragment][k-step]
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int di = 0; di < DI; ++di) { BS[a][di][0] = z; BS[a][di][1] = z; }
    }
    const unsigned stat_bytes = (unsigned)((size_t)col_segs * 2 * g.bpad * 4);
    const BufRsrc rs_rz = make_rsrc(rz_cols, stat_bytes), rs_wrz = make_rsrc(wrz_cols, stat_bytes);
    const BufRsrc rs_k = make_rsrc(SW ? kc : rz_cols, stat_bytes);
    unsigned char* ebuf = lds + E0 + wave * 2048;      // + stage * ESTG
    unsigned char* sbuf = lds + S0 + wave * 256;       // + stage * SSTG: omega/Z (or w omega/Z) of the tile's 32 columns
    unsigned char* kbuf = lds + K0 + wave * 256;       // SW: k of the tile's columns

    f32x16 acc[4][DI];
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
#pragma unroll
        for (int di = 0; di < DI; ++di)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pi][di][r] = 0.f;

    int t = blockIdx.y * tiles_per_slice;
    int t_end = t + tiles_per_slice;
    if (t_end > NT) t_end = NT;
    // (Equal-length slices on purpose.  A mirrored tile takes ~8-13 % longer than a direct one, so the slice-0 blocks of the upper half of the row
    // blocks run longest; cutting the slices at equal cost instead levelled the block times (tools/timeline.py) and changed the
    // launch time by nothing measurable (profiles/r03_timeline_*.txt): under the package power limit the blocks still running speed up
    // as the others retire.)
    auto clampt = [&](int u) { return u < t_end ? u : t_end - 1; };   // past the end: re-fetch the last tile (fixed VMEM count)
    struct Col { int u, mt, seg, in_seg; };
    auto col_seg_start = [&](Col& c) {
        c.in_seg = 0;
        int r = c.seg;
        if (g.col_wrap > 0) { r += g.col_rank0; if (r >= g.col_wrap) r -= g.col_wrap; }
        c.mt = r * per_rank;
    };
    auto col_at = [&](int u) {
        Col c;
        c.u = u = clampt(u);
        if (!RECT) { c.mt = u; c.seg = 0; c.in_seg = u; return c; }
        const int su = u / per_rank;
        c.seg = su + ((skip_seg >= 0 && su >= skip_seg) ? 1 : 0);
        col_seg_start(c);
        c.in_seg = u - su * per_rank;
        c.mt += c.in_seg;
        return c;
    };
    auto col_next = [&](Col& c) {
        if (c.u + 1 >= t_end) return;                   // stay on the last tile
        ++c.u; ++c.mt; ++c.in_seg;
        if (RECT && c.in_seg == per_rank) { ++c.seg; if (c.seg == skip_seg) ++c.seg; col_seg_start(c); }
    };
    auto col_mod = [&](const Col& c) { return c.in_seg >= per_mod ? 1 : 0; };
    // the local block's cursors are functions of the tile index alone (two scalar instructions); rectangular launches walk segments
    auto col_of = [&](int u) { u = clampt(u); return Col{u, u, 0, u}; };
    auto issue_x_piece = [&](const Col& c, int stage, int k) {
        if (CROSSCLR_DABL & 2) return;
        lds_dma16_buf(rs_x, voffx[k], (unsigned)((CROSSCLR_DABL & 512) ? (c.mt & 7) : c.mt) * (unsigned)(QT * RBG), lds + stage * TILE + (wave + 4 * k) * 1024);
    };
    // XF: fragment (di, ks) of column tile c into register set `setc`
    auto load_xf = [&](auto setc, auto dic, auto ksc, const Col& c) {
        constexpr int S = decltype(setc)::value, di = decltype(dic)::value, ks = decltype(ksc)::value;
        if (CROSSCLR_DABL & 2) return;
        constexpr int off = (2 * di + ks) * 1024;
        unsigned so = (unsigned)((CROSSCLR_DABL & 512) ? (c.mt & 7) : c.mt) * (unsigned)(QT * RBG);
        pin_s(so);
        BS[S][di][ks] = buf_load_b128_async<(off & 4095)>(rs_xf, off >= 4096 ? xfv1 : xfv0, so);
    };
    auto xf_landed = [&](auto setc) {      // (the caller has waited)
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int di = 0; di < DI; ++di) { after_wait(BS[S][di][0]); after_wait(BS[S][di][1]); }
    };
    // Stash tile of this wave for column tile u: (r32, u) where the forward evaluated it for these rows (u >= rb0: index Cd + u),
    // (u, r32) left of that (mirrored).  stash_tile_index in 32-bit scalar arithmetic (the host refuses plans with 2^31 tiles or
    // more), cut into three stages so that the ~15 scalar instructions can be dealt over several MFMA slots.
    const unsigned Cd = RECT ? (unsigned)r32 * (unsigned)NT
                             : TPR * (unsigned)(r32 / TPR) * ((unsigned)NT - (TPR / 2) * (unsigned)(r32 / TPR) + (TPR / 2)) +
                               (unsigned)(r32 % TPR) * ((unsigned)NT - TPR * (unsigned)(r32 / TPR)) - TPR * (unsigned)(r32 / TPR);
    struct EAddr { unsigned R8, wp, m1, b; };
    // known_direct: the caller knows that u >= rb0 (every tile behind a direct one is direct): one addition
    auto eaddr_stage = [&](int stage, int u, EAddr& a, const unsigned char*& out, bool known_direct) {
        if (TR) {       // stash tile (this rank's row group u, item = partner segment * per_rank + the output row group)
            if (stage == 2) { unsigned idx = (unsigned)u * (unsigned)(g.col_ranks * per_rank) + (unsigned)(g.skip_rank * per_rank + r32); pin_s(idx); out = stash + (size_t)idx * 2048; }
        } else if (RECT || known_direct) {
            if (stage == 2) { unsigned idx = Cd + (unsigned)u; pin_s(idx); out = stash + (size_t)idx * 2048; }
        } else if (stage == 0) {
            a.R8 = (unsigned)u & ~(unsigned)(TPR - 1);
            a.wp = (unsigned)u & (unsigned)(TPR - 1);
            pin_s(a.R8); pin_s(a.wp);
        } else if (stage == 1) {
            a.m1 = a.R8 * ((unsigned)NT + (TPR / 2) - a.R8 / 2);
            a.b = (unsigned)NT - a.R8;
            pin_s(a.m1); pin_s(a.b);
        } else {
            // branch-free select (a branch here would split the basic block that holds the asm loads and their waits)
            const unsigned dmask = (unsigned)-(int)(u >= rb0);
            const unsigned mir = a.m1 + a.wp * a.b + ((unsigned)r32 - a.R8), dir = Cd + (unsigned)u;
            unsigned idx = (dir & dmask) | (mir & ~dmask);
            pin_s(idx);
            out = stash + (size_t)idx * 2048;
        }
    };
    auto eaddr_of = [&](int u) {
        EAddr a = {0, 0, 0, 0};
        const unsigned char* out = stash;
        eaddr_stage(0, u, a, out, false); eaddr_stage(1, u, a, out, false); eaddr_stage(2, u, a, out, false);
        return out;
    };
    // piece 0 / 1: the wave's 2-KiB stash tile as stored (lane-linear); piece 2: the tile's statistics; piece 3 (SW): its k
    auto issue_e_piece = [&](const Col& c, const unsigned char* etile, int estage, int k) {
        if (CROSSCLR_DABL & 1) return;
        if (k < 2) {
            const BufRsrc rs_e = make_rsrc((CROSSCLR_DABL & 256) ? stash + ((size_t)(etile - stash) & (size_t)0x1FF800) : etile, 2048u);
            lds_dma16_buf(rs_e, (unsigned)(lane * 16 + 1024 * k), 0u, ebuf + estage * ESTG + 1024 * k);
        } else if (k == 2) {
            const bool same = col_mod(c) == rmod;
            lds_dma4_buf(same ? rs_wrz : rs_rz, (unsigned)(lane * 4), (unsigned)(c.mt * QT * 4), sbuf + estage * SSTG);
        } else if (SW) {
            lds_dma4_buf(rs_k, (unsigned)(lane * 4), (unsigned)(c.mt * QT * 4), kbuf + estage * SSTG);
        }
    };
    struct Pair { s16x4 lo, hi; };
    struct Bits8 { bf16_t e[8]; };
    // tile u as this wave weighs it: saved exponentials, row statistic(s) of the lane, column statistics of its 16 columns
    struct Staged { u32x4 e[2]; u32x4 cs[4]; u32x4 kc[4]; unsigned rs, kr; };
    // MIR: the stored tile is E^T -- the lane's row is column q = l31 of the tile, its 16 columns are rows of this wave's group
    // part 0: the exponentials (+ the lane's own row statistic of a mirrored tile); part 1: column statistics; part 2 (SW): column k
    auto read_staged = [&](auto mir, auto partc, const Col& c, int estage, Staged& st) {
        constexpr bool MIR = decltype(mir)::value;
        constexpr int P = decltype(partc)::value;
        if constexpr (P == 0) {
            const auto ed = lds_addr(ebuf + estage * ESTG + 16 * lane);
            st.e[0] = lds_read_b128_async<0>(ed);
            st.e[1] = lds_read_b128_async<1024>(ed);
            if (MIR) {
                st.rs = lds_read_b32_async<0>(lds_addr(sbuf + estage * SSTG + 4 * l31));
                if (SW) st.kr = lds_read_b32_async<0>(lds_addr(kbuf + estage * SSTG + 4 * l31));
            }
        } else if constexpr (P == 1) {
            // quad (th, r4): columns 16th + 8r4 + 4half ..+3 -- of the tile (direct) or of this wave's own row group (mirrored)
            const bool same = col_mod(c) == rmod;
            const auto sa = MIR ? lds_addr(lds + O0 + (same ? 512 : 0) + 128 * wave + 16 * half) : lds_addr(sbuf + estage * SSTG + 16 * half);
            st.cs[0] = lds_read_b128_async<0>(sa);
            st.cs[1] = lds_read_b128_async<32>(sa);
            st.cs[2] = lds_read_b128_async<64>(sa);
            st.cs[3] = lds_read_b128_async<96>(sa);
        } else if constexpr (SW) {
            const auto ka = MIR ? lds_addr(lds + O0 + 1024 + 128 * wave + 16 * half) : lds_addr(kbuf + estage * SSTG + 16 * half);
            st.kc[0] = lds_read_b128_async<0>(ka);
            st.kc[1] = lds_read_b128_async<32>(ka);
            st.kc[2] = lds_read_b128_async<64>(ka);
            st.kc[3] = lds_read_b128_async<96>(ka);
        }
    };
    auto staged_landed = [&](auto mir, Staged& st) {       // (the caller has waited: lgkmcnt(0))
        constexpr bool MIR = decltype(mir)::value;
        after_wait(st.e[0]); after_wait(st.e[1]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { after_wait(st.cs[k]); if (SW) after_wait(st.kc[k]); }
        if (MIR) { after_wait(st.rs); if (SW) after_wait(st.kr); }
    };
    // W for two columns (k-step th, register quad r4, pair hh) of the staged tile, packed to bf16 in place
    auto weigh2 = [&](auto mir, const Col& c, const Staged& st, Bits8 (&pk)[2], int th, int r4, int hh) {
        constexpr bool MIR = decltype(mir)::value;
        const bool same_mod = col_mod(c) == rmod;
        const bool weighted = SW && same_mod;
        const float rs = MIR ? __builtin_bit_cast(float, st.rs) : (same_mod ? rzp_intra : rzp_inter);
        const float kr = MIR ? (SW ? __builtin_bit_cast(float, st.kr) : 1.f) : kp;
        const f32x4 cs = __builtin_bit_cast(f32x4, st.cs[2 * th + r4]);
        f32x4 kq = {1.f, 1.f, 1.f, 1.f};
        if (weighted) kq = __builtin_bit_cast(f32x4, st.kc[2 * th + r4]);
        const Bits8 ev = __builtin_bit_cast(Bits8, st.e[th]);
#pragma unroll
        for (int j = 2 * hh; j < 2 * hh + 2; ++j) {
            const float v = bf16_bits_to_f32(ev.e[4 * r4 + j]);
            // (one expression for every tile of a weighted launch: scales of 1.0 for the other modality's tiles give rs + cs exactly;
            //  crossclr_kernels_dslp.h forms the same FMA from ones it reads out of LDS)
            const float zz = SW ? __builtin_fmaf(rs, kq[j], cs[j] * (weighted ? kr : 1.f)) : (rs + cs[j]);
            pk[th].e[4 * r4 + j] = (CROSSCLR_DABL & 4) ? ev.e[4 * r4 + j] : f32_to_bf16_bits(v * zz);
        }
    };
    auto write_w = [&](auto mir, int wslot, const Bits8 (&pk)[2]) {
        constexpr bool MIR = decltype(mir)::value;
        if (CROSSCLR_DABL & 8) return;
        unsigned char* wb = lds + W0 + wslot * WBUF;
        if (MIR) {
            *reinterpret_cast<u32x4*>(wb + wr_mir) = __builtin_bit_cast(u32x4, pk[0]);
            *reinterpret_cast<u32x4*>(wb + wr_mir + 128) = __builtin_bit_cast(u32x4, pk[1]);
        } else {
            *reinterpret_cast<u32x4*>(wb + wr_dir) = __builtin_bit_cast(u32x4, pk[0]);
            *reinterpret_cast<u32x4*>(wb + wr_dir + 1024) = __builtin_bit_cast(u32x4, pk[1]);
        }
    };

    bf16x8 A1c[4], B1c[DI];     // k-step 1 of the previous tile: read before the barrier, multiplied after it
    {
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) A1c[pi] = z;
#pragma unroll
        for (int di = 0; di < DI; ++di) B1c[di] = z;
    }

    if (t < t_end) {
        const int tm = RECT ? t : (TR ? t_end : (t_end < rb0 ? t_end : rb0));        // tiles [t, tm) are mirrored, [tm, t_end) direct
        Col cw, cx, ce;    // the next tile to weigh (t+1), to fetch (t+2; XF: t+1), to fetch saved exponentials for (t+1+PE)
        if constexpr (XF) {     // X(t) first: the wait for E(t) below then covers it, and E(t+1) / E(t+2) stay in flight behind both
            cx = col_at(t);
            static_for<NXO>([&](auto jc) { constexpr int j = decltype(jc)::value; load_xf(IdxC<0>{}, IdxC<j / 2>{}, IdxC<j % 2>{}, cx); });
        }
        ce = col_at(t);
#pragma unroll
        for (int k = 0; k < NSE; ++k) {
            const unsigned char* et = eaddr_of(ce.u);
#pragma unroll
            for (int j = 0; j < NEO; ++j) issue_e_piece(ce, et, k, j);
            col_next(ce);
        }
        const unsigned char* etile = eaddr_of(ce.u);      // stash tile of the E DMA the first iteration issues
        if constexpr (!XF) {
            cx = col_at(t);
#pragma unroll
            for (int k = 0; k < NSX - 1; ++k) {
#pragma unroll
                for (int j = 0; j < NXO; ++j) issue_x_piece(cx, k, j);
                col_next(cx);
            }
        }
        cw = col_at(t);
        if (!(CROSSCLR_DABL & 3)) wait_dma_keep<(NSE - 1) * NEO + (XF ? 0 : (NSX - 1) * NXO)>();   // E / statistics of the first tile (XF: and X(t))
        if constexpr (XF) xf_landed(IdxC<0>{});
        {
            Bits8 pk[2];
            Staged st;
            if (t < tm) {
                static_for<3>([&](auto pc) { read_staged(IdxC<true>{}, pc, cw, 0, st); });
                wait_lgkm_all();
                staged_landed(IdxC<true>{}, st);
#pragma unroll
                for (int q = 0; q < 8; ++q) weigh2(IdxC<true>{}, cw, st, pk, q >> 2, (q >> 1) & 1, q & 1);
                write_w(IdxC<true>{}, 0, pk);
            } else {
                static_for<3>([&](auto pc) { read_staged(IdxC<false>{}, pc, cw, 0, st); });
                wait_lgkm_all();
                staged_landed(IdxC<false>{}, st);
#pragma unroll
                for (int q = 0; q < 8; ++q) weigh2(IdxC<false>{}, cw, st, pk, q >> 2, (q >> 1) & 1, q & 1);
                write_w(IdxC<false>{}, 0, pk);
            }
        }
        col_next(cw);
        if constexpr (XF) { if (!(CROSSCLR_DABL & 3)) wait_dma_keep<(NSE - 2) * NEO>(); }     // E(t+1): the first iteration stages it at once
        else if (!(CROSSCLR_DABL & 3)) wait_dma_keep<(NSX - 2) * NXO>();     // X(t) (and every E piece: they were issued first)
        barrier_keep_dma();
        timing_mark(1);
        int sx = 0, se = 1 % NSE, wslot = 0;
        bool first_iter = true;
        int sx_free = 0, se_free = 0, ue_next = 0;
        decltype(lds_addr(lds)) xa = 0, wa_dir = 0, wa_mir = 0;
        auto setup_a = [&]() {     // cursors and ring stages of iteration t
            if constexpr (!RECT) { cw = col_of(t + 1); cx = col_of(XF ? t + 1 : t + 2); ce = col_of(t + 1 + PE); }
            sx_free = sx == 0 ? NSX - 1 : sx - 1;        // stage of tile t-1 = stage of tile t+2
            se_free = se == 0 ? NSE - 1 : se - 1;        // stage of tile t (weighed an iteration ago) = stage of tile t+1+PE
            ue_next = RECT ? 0 : clampt(t + 2 + PE);     // (RECT: the cursor itself is advanced, see below)
            pin_s(sx_free); pin_s(se_free); pin_s(ue_next);
            if constexpr (!RECT) { pin_s(cw.u); pin_s(cx.u); pin_s(ce.u); cw = Col{cw.u, cw.u, 0, cw.u}; cx = Col{cx.u, cx.u, 0, cx.u}; ce = Col{ce.u, ce.u, 0, ce.u}; }
        };
        auto setup_b = [&]() {     // LDS addresses of iteration t
            xa = lds_addr(lds + sx * TILE);
            wa_dir = lds_addr(lds + W0 + wslot * WBUF + rd_dir);
            wa_mir = lds_addr(lds + W0 + wslot * WBUF + rd_mir);
            pin_s(xa); pin_v(wa_dir); pin_v(wa_mir);
        };
        setup_a(); setup_b();

        // One iteration = 2 H MFMA slots { MFMA ; a share of the chores }, pinned by sched_fence() on both sides of the MFMA (scalar
        // address arithmetic stays in the slot of the chore that needs it instead of piling up at the top of the loop).
        //   first half  (k-step 1 of tile t-1): E / statistics DMA of tile t+1+PE, staged reads of tile t+1, reads of k-step 0 of tile t,
        //               pieces of X(t+2); its last slot waits for the reads (issued many MFMAs earlier: a formality)
        //   second half (k-step 0 of tile t): reads of k-step 1 of tile t (carried across the barrier), the 56 VALU of W(t+1), the
        //               next E address, the W write; the last slot moves the cursors and rotates the rings
        auto body = [&](auto mc_, auto mn_, auto par_) {
            constexpr bool MC = decltype(mc_)::value, MN = decltype(mn_)::value;
            constexpr int PAR = decltype(par_)::value;       // XF: register set of tile t (the other one: tile t-1, then tile t+1)
            // (sx_free, se_free, xa, wa_dir / wa_mir, ue_next and the cursors of THIS iteration were set in the last two slots of the
            // previous one -- setup_a / setup_b below -- so that no address arithmetic sits between the barrier and the first MFMA)
            const auto wa = MC ? wa_mir : wa_dir;
            u32x4 Ad0[4], Ad1[4];
            Pair Am0[4], Am1[4], B0[DI], B1[DI];
            Staged st;
            Bits8 pk[2];
            EAddr ea = {0, 0, 0, 0};
            const unsigned char* etile_next = etile;
            auto read_a = [&](auto pic, auto thc) {
                constexpr int pi = decltype(pic)::value, th = decltype(thc)::value;
                if (MC) {
                    Pair p;
                    if (CROSSCLR_DABL & 64) p = __builtin_bit_cast(Pair, A1c[pi]);
                    else {
                        p.lo = lds_read_tr16_b64_async<pi * 2048 + th * 1024>(wa);
                        p.hi = lds_read_tr16_b64_async<pi * 2048 + th * 1024 + 512>(wa);
                    }
                    if (th) Am1[pi] = p; else Am0[pi] = p;
                } else {
                    const u32x4 v = (CROSSCLR_DABL & 64) ? __builtin_bit_cast(u32x4, A1c[pi]) : lds_read_b128_async<pi * 2048 + th * 1024>(wa);
                    if (th) Ad1[pi] = v; else Ad0[pi] = v;
                }
            };
            auto read_b = [&](auto dic, auto thc) {
                constexpr int di = decltype(dic)::value, th = decltype(thc)::value;
                Pair p;
                if (CROSSCLR_DABL & 16) p = __builtin_bit_cast(Pair, B1c[di]);
                else {
                    p.lo = lds_read_tr16_b64_async<(16 * th) * RB>(xa + bo[di][0]);
                    p.hi = lds_read_tr16_b64_async<(16 * th + 8) * RB>(xa + bo[di][1]);
                }
                if (th) B1[di] = p; else B0[di] = p;
            };
            // ---- VMEM: NV = NXO + NEO pieces per wave and iteration, dealt evenly over the 2 H slots: first the pieces of X(t+2), THEN the
            // saved exponentials / statistics of tile t+1+PE.  The order matters: vmcnt retires in order, so the wait that proves
            // X(t+1) landed also waits for everything issued before it -- with E first, every iteration waited for an HBM access it had
            // issued only one tile earlier (measured: 0.024 ms of the kernel; E from an L2-resident window took it away).
            // (CROSSCLR_DSL_STAGGER: piece j of wave WV sits in slot floor((NV WV + j) 2H / (4 NV)) instead -- each wave in its own
            // quarter of the iteration, one wave at a time talking to the texture addresser.)
            constexpr int NV = NEO + NXO;
            constexpr int R2 = H / 2 > 0 ? H / 2 : 1;                // second half: slots for the reads / for the half-quads
            constexpr int WSLOT = 1 + R2 <= H - 1 ? 1 + R2 : H - 1;  // ... slot of the W write
            constexpr int ADV = 2 * H - 4 > H + WSLOT ? 2 * H - 4 : H + WSLOT;   // slot that moves on to the next tile: behind every user of the cursors
            constexpr int SETB = ADV + 1 <= 2 * H - 1 ? ADV + 1 : 2 * H - 1;
            auto vmem_items = [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                static_for<NV>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (XF) {
                        // fragment loads of tile t+1 into the set of tile t-1: k-step 0 (dead since the last iteration) from slot 0, k-step 1 of
                        // fragment di right behind its last MFMA (slot 4 di + 3); the saved-exponential pieces behind ALL of them, up to slot ADV
                        if constexpr (j < NXO) {
                            constexpr int di = j >> 1, ks = j & 1;
                            if constexpr ((ks == 0 ? di : 4 * (di + 1)) == s) load_xf(IdxC<PAR ^ 1>{}, IdxC<di>{}, IdxC<ks>{}, cx);
                        } else {
                            constexpr int je = j - NXO;
                            if constexpr (H + 1 + (je * (ADV - H - 1)) / (NEO - 1) == s) issue_e_piece(ce, etile, se_free, je);
                        }
                    } else if constexpr ((WV >= 0 ? ((NV * WVS + j) * ADV) / (4 * NV) : (j * ADV) / NV) == s) {
                        if constexpr (j < NXO) issue_x_piece(cx, sx_free, j);
                        else issue_e_piece(ce, etile, se_free, j - NXO);
                    }
                });
            };
            // ---- first half: LDS items, one per slot from slot 0 (the last one well ahead of the wait in slot H-1) ----
            constexpr int NS1 = 2 + (SW ? 1 : 0);                    // staged reads, in two or three instalments
            constexpr int N1 = NS1 + 4 + (XF ? 0 : DI);
            constexpr int L1 = (2 * H) / 3 > 0 ? (2 * H) / 3 : 1;    // slots that carry them
            auto item1 = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < NS1) read_staged(IdxC<MN>{}, IdxC<i>{}, cw, se, st);
                else if constexpr (i < NS1 + 4) read_a(IdxC<i - NS1>{}, IdxC<0>{});
                else read_b(IdxC<i - NS1 - 4>{}, IdxC<0>{});
            };
            auto items1 = [&](auto sc) {
                constexpr int s = decltype(sc)::value;
                static_for<N1>([&](auto ic) { if constexpr ((decltype(ic)::value * L1) / N1 == s) item1(ic); });
            };
            // ---- second half: reads of k-step 1 from its first slot on, the half-quads of W(t+1) one slot behind, then the write and the
            // next E address; the last slots carry nothing that the closing lgkmcnt(0) would have to wait for ----
            constexpr int NR = 4 + (XF ? 0 : DI);
            auto items2 = [&](auto sc) {
                constexpr int s2 = decltype(sc)::value;
                static_for<NR>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr ((i * R2) / NR == s2) { if constexpr (i < 4) read_a(IdxC<i>{}, IdxC<1>{}); else read_b(IdxC<i - 4>{}, IdxC<1>{}); }
                });
                static_for<8>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    if constexpr (1 + (q * R2) / 8 == s2 || (1 + (q * R2) / 8 > H - 1 && s2 == H - 1)) weigh2(IdxC<MN>{}, cw, st, pk, q >> 2, (q >> 1) & 1, q & 1);
                });
                if constexpr (s2 == WSLOT) write_w(IdxC<MN>{}, wslot ^ 1, pk);
                if constexpr (!RECT) static_for<3>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr ((WSLOT + 1 + k <= ADV - H ? WSLOT + 1 + k : ADV - H) == s2) eaddr_stage(k, ue_next, ea, etile_next, !MC);
                });
            };
            static_for<2 * H>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s < H) {
                    constexpr int pi = XF ? s % 4 : s / DI, di = XF ? s / 4 : s % DI;     // XF: fragment-major (see vmem_items)
                    if constexpr (XF) mfma_acc(acc[pi][di], A1c[pi], __builtin_bit_cast(bf16x8, BS[PAR ^ 1][di][1]));
                    else mfma_acc(acc[pi][di], A1c[pi], B1c[di]);
                    sched_fence();
                    items1(sc);
                    vmem_items(sc);
                    if constexpr (s == H - 1) {
                        if (!(CROSSCLR_DABL & 1024)) wait_lgkm_all();
                        staged_landed(IdxC<MN>{}, st);
#pragma unroll
                        for (int p = 0; p < 4; ++p) { if (MC) { after_wait(Am0[p].lo); after_wait(Am0[p].hi); } else after_wait(Ad0[p]); }
                        if constexpr (!XF) {
#pragma unroll
                            for (int d = 0; d < DI; ++d) { after_wait(B0[d].lo); after_wait(B0[d].hi); }
                        }
                    }
                } else {
                    constexpr int s2 = s - H, pi = s2 / DI, di = s2 % DI;
                    mfma_acc(acc[pi][di], MC ? __builtin_bit_cast(bf16x8, Am0[pi]) : __builtin_bit_cast(bf16x8, Ad0[pi]),
                             XF ? __builtin_bit_cast(bf16x8, BS[PAR][di][0]) : __builtin_bit_cast(bf16x8, B0[di]));
                    sched_fence();
                    items2(IdxC<s2>{});
                    vmem_items(sc);
                    if constexpr (!RECT && s == ADV) {     // nothing of this iteration reads the cursors or the ring stages any more
                        etile = etile_next;
                        sx = sx + 1 == NSX ? 0 : sx + 1;
                        se = se + 1 == NSE ? 0 : se + 1;
                        wslot ^= 1;
                        ++t;
                        setup_a();
                    }
                    if constexpr (!RECT && s == SETB) setup_b();
                }
                sched_fence();
            });
            // k-step 1 of this tile has been read (and W(t+1) written): all LDS traffic of the iteration is complete ...
            if (!(CROSSCLR_DABL & 2048)) wait_lgkm_all();
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (MC) { after_wait(Am1[p].lo); after_wait(Am1[p].hi); A1c[p] = __builtin_bit_cast(bf16x8, Am1[p]); }
                else { after_wait(Ad1[p]); A1c[p] = __builtin_bit_cast(bf16x8, Ad1[p]); }
            }
            if constexpr (!XF) {
#pragma unroll
                for (int d = 0; d < DI; ++d) { after_wait(B1[d].lo); after_wait(B1[d].hi); B1c[d] = __builtin_bit_cast(bf16x8, B1[d]); }
            }
            // ... X(t+1) has landed (what this iteration issued may still be in flight), for every wave
            if constexpr (RECT) {   // the segment walk of a rectangular launch branches: only here, where no asm load is in flight
                col_next(cw); col_next(cx); col_next(ce);
                etile = eaddr_of(ce.u);
                sx = sx + 1 == NSX ? 0 : sx + 1;
                se = se + 1 == NSE ? 0 : se + 1;
                wslot ^= 1;
                ++t;
                setup_a(); setup_b();
            }
            // (operations issued after X(t+1)'s last piece: the E pieces of the previous iteration and everything of this one; the
            // first iteration's X(t+1) is the last operation of the prologue -- only this iteration's operations follow it)
            if constexpr (XF) {      // tile t+1's fragments have landed: only this iteration's NEO pieces were issued behind the last of them
                if (!(CROSSCLR_DABL & (3 | 4096))) wait_dma_keep<NEO>();
                xf_landed(IdxC<PAR ^ 1>{});
            } else if (!(CROSSCLR_DABL & (3 | 4096))) {
                if (first_iter) wait_dma_keep<NXO + NEO>(); else wait_dma_keep<NXO + 2 * NEO>();
            }
            first_iter = false;
            if (!(CROSSCLR_DABL & 32)) barrier_keep_dma();
        };
        if constexpr (XF) {
            // The loop unrolled by two: iteration parity = register set of its tile.  Every phase (M->M, M->D, D->D) starts on set 0; a phase
            // that ends after an odd number of iterations swaps the two sets (twice per block at most) -- a dispatch on a run-time parity
            // instead (six bodies reachable from each other) made hipcc split the accumulators' live ranges and spill.
            auto swap_sets = [&]() {
#pragma unroll
                for (int di = 0; di < DI; ++di)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) { const u32x4 x = BS[0][di][ks]; BS[0][di][ks] = BS[1][di][ks]; BS[1][di][ks] = x; }
            };
            // (an odd iteration is peeled off IN FRONT of each loop: a second exit in the middle of the unrolled loop made hipcc split the
            // accumulators' live ranges as well)
            const int n_mm = tm - 1 - t > 0 ? tm - 1 - t : 0;
            if (n_mm & 1) { body(IdxC<true>{}, IdxC<true>{}, IdxC<0>{}); swap_sets(); }
            for (int k = n_mm >> 1; k > 0; --k) {
                body(IdxC<true>{}, IdxC<true>{}, IdxC<0>{});
                body(IdxC<true>{}, IdxC<true>{}, IdxC<1>{});
            }
            if (t < tm) { body(IdxC<true>{}, IdxC<false>{}, IdxC<0>{}); swap_sets(); }
            const int n_dd = t_end - t;
            if (n_dd & 1) { body(IdxC<false>{}, IdxC<false>{}, IdxC<0>{}); swap_sets(); }
            for (int k = n_dd >> 1; k > 0; --k) {
                body(IdxC<false>{}, IdxC<false>{}, IdxC<0>{});
                body(IdxC<false>{}, IdxC<false>{}, IdxC<1>{});
            }
            // k-step 1 of the last tile: after the swaps the next tile would use set 0, the last one sits in set 1
            static_for<H>([&](auto sc) { constexpr int s = decltype(sc)::value; mfma_acc(acc[s % 4][s / 4], A1c[s % 4], __builtin_bit_cast(bf16x8, BS[1][s / 4][1])); });
        } else {
        if constexpr (TR) {            // every tile mirrored (the last iteration weighs a re-fetch of the last tile: never consumed)
            while (t < t_end) body(IdxC<true>{}, IdxC<true>{}, IdxC<0>{});
        } else {
            if constexpr (!RECT) {     // (a rectangular block's tiles are all direct)
                while (t + 1 < tm) body(IdxC<true>{}, IdxC<true>{}, IdxC<0>{});
                if (t < tm) body(IdxC<true>{}, IdxC<false>{}, IdxC<0>{});
            }
            while (t < t_end) body(IdxC<false>{}, IdxC<false>{}, IdxC<0>{});
        }
        // k-step 1 of the last tile
        static_for<H>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            mfma_acc(acc[s / DI][s % DI], A1c[s / DI], B1c[s % DI]);
        });
        }
        wait_dma();   // the re-fetches past the end must not outlive the block's LDS
    }
    timing_mark(2);
    // G[row][d]: lane holds column d = 32 (DI wave + di) + l31 of fragment (pi, di) and 16 rows; buffer addressing (one per-lane offset, the row / fragment part as a scalar)
    constexpr unsigned GP = XP * DK * 16 * 4;           // bytes per gradient row
    const BufRsrc rs_g = make_rsrc(gbuf + (size_t)blockIdx.y * 2 * g.bpad * (XP * DK * 16) + (size_t)row0b * (XP * DK * 16) + part * (DK * 16),
                                   128u * GP);          // this block's 128 rows
    const unsigned vg = (unsigned)((4 * half) * GP + l31 * 4 + 128 * DI * wave);
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {
#pragma unroll
        for (int di = 0; di < DI; ++di) {
            float o[16];
            if (accumulate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = buf_load4(rs_g, vg, (unsigned)((32 * pi + 8 * (r >> 2) + (r & 3)) * GP + 128 * di));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.f;
            }
            sched_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r)
                buf_store4(rs_g, vg, (unsigned)((32 * pi + 8 * (r >> 2) + (r & 3)) * GP + 128 * di), o[r] + acc[pi][di][r]);
            sched_fence();
        }
    }
    timing_mark(3);
}

template <int DK, bool SW, int MODE, int XP = 1, int TPRF = 8, bool XF = false>
__global__ void __launch_bounds__(256, 1) fast_bwd_dsl_kernel(const bf16_t* cols, const unsigned char* stash, Geo g,
                                                              const float* rz, const float* wrz,
                                                              const float* rz_cols, const float* wrz_cols, float* gbuf,
                                                              int accumulate, int tiles_per_slice, const float* ks,
                                                              const float* kc) {
    constexpr int RB = DK * 32;            // bytes per row of the (part of the) operand a block multiplies
    constexpr int QT = 32;
    constexpr int TILE = QT * RB;
    constexpr int TPR = TPRF;
    constexpr int RBG = XP * RB;           // bytes per operand row in memory
    constexpr int DI = DK / 8;             // 32-wide output fragments per wave (its column slice)
    constexpr int H = 4 * DI;              // MFMAs per k-step: 4 row groups x DI fragments
    constexpr int NSX = XF ? 0 : 3;        // column-tile ring
    constexpr int PE = XF ? 2 : CROSSCLR_DSL_PE;
    constexpr int NSE = PE + 1;            // saved-exponential + statistics rings (private to a wave)
    constexpr int ESTG = 4 * 2048;         // one stage of the E ring: [4 waves][2 KiB]
    constexpr int SSTG = 4 * 256;          // one stage of a statistics ring: [4 waves][64 floats] (the tile's 32 + 32 spare)
    constexpr int NXO = DK / 4;            // VMEM operations per wave and tile: column-tile pieces ...
    constexpr int NEO = 3 + (SW ? 1 : 0);  // ... saved-exponential pieces + statistics
    constexpr int WBUF = 4 * 2048;         // one W slot: [4 row groups][2 KiB]
    constexpr int W0 = NSX * TILE, E0 = W0 + 2 * WBUF, S0 = E0 + NSE * ESTG, K0 = S0 + NSE * SSTG;
    constexpr int O0 = K0 + (SW ? NSE * SSTG : 0);      // the block's own statistics: rz[128] | wrz[128] | k[128]
    static_assert(DK % 8 == 0 && DK >= 8 && DK <= 32, "Dpad (per part) in {128, 256, 384, 512}");
    static_assert(XF || PE >= 3, "E / statistics of tile t+2 must be older than the pieces of X(t+1)");
    static_assert(O0 + 3 * 512 <= 160 * 1024, "LDS budget");
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[O0 + 3 * 512];
    const int tid = threadIdx.x;
    const int row0b = blockIdx.x * 128;
    timing_mark(0);
    // the block's own statistics (mirrored tiles read them as COLUMN statistics)
    if (tid < 128) {
        float* own = reinterpret_cast<float*>(lds + O0);
        own[tid] = rz[row0b + tid];
        own[128 + tid] = wrz[row0b + tid];
        own[256 + tid] = SW ? ks[row0b + tid] : 1.f;
    }
    __syncthreads();        // (before any LDS-DMA is in flight: this barrier may drain VMEM)
#if !CROSSCLR_DSL_STAGGER
    if constexpr (XF) { dsl_wave<DK, SW, MODE, XP, TPRF, -1, true>(lds, cols, stash, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tiles_per_slice, ks, kc); return; }
    dsl_wave<DK, SW, MODE, XP, TPRF, -1>(lds, cols, stash, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tiles_per_slice, ks, kc);
#else
    switch (uniform(tid >> 6)) {
        case 0: dsl_wave<DK, SW, MODE, XP, TPRF, 0>(lds, cols, stash, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tiles_per_slice, ks, kc); break;
        case 1: dsl_wave<DK, SW, MODE, XP, TPRF, 1>(lds, cols, stash, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tiles_per_slice, ks, kc); break;
        case 2: dsl_wave<DK, SW, MODE, XP, TPRF, 2>(lds, cols, stash, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tiles_per_slice, ks, kc); break;
        default: dsl_wave<DK, SW, MODE, XP, TPRF, 3>(lds, cols, stash, g, rz, wrz, rz_cols, wrz_cols, gbuf, accumulate, tiles_per_slice, ks, kc); break;
    }
#endif
}

}  // namespace crossclr
