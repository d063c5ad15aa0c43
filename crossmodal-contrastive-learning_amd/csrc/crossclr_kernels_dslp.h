// crossclr_kernels_dslp.h -- the saved backward of the local block on the fragment-major operand, TWO column tiles per barrier interval.
//
// Same math, same data and the same MFMA sequence per accumulator as fast_bwd_dsl_kernel<..., XF> (crossclr_kernels_dsl.h; autograd of
// reference trainer/loss.py:83-112 from the forward's bf16 stash): G[p][:] = sum_q W[p][q] X[q][:], W = E (omega_p/Z_p + omega_q/Z_q),
// a wave = all 128 rows x a Dpad/4 column slice, W shared through LDS, column tiles as MFMA B fragments straight from the fragment-major
// copy of the operand.  What changed is the bookkeeping around the 32 MFMAs of a tile, which round 3's PMC showed to be the kernel's
// limit (5.2 non-MFMA issues per MFMA, MFMA-busy 0.52; profiles/r03_pmc.json):
//
//   one iteration = one PAIR of 32-column tiles = 4 quarters of H = 4 DI MFMAs:
//       Q0  k-step 1 of the previous pair's second tile (fragments read before the barrier)      B: set 1, k-step 1
//       Q1  k-step 0 of tile a                                                                    B: set 0, k-step 0
//       Q2  k-step 1 of tile a                                                                    B: set 0, k-step 1
//       Q3  k-step 0 of tile b                                                                    B: set 1, k-step 0
//     -> ONE barrier, ONE closing wait pair and ONE set of cursor updates per 64 MFMAs (was: per 32); the register set of a tile is
//     static (a -> set 0, b -> set 1): no parity unrolling, no peeled iterations, three loop bodies (M->M, M->D, D->D) instead of seven.
//   scalar side: every address that moves per pair is ONE loop-carried SGPR offset bumped by a constant (fragment offset, statistics
//     offset, direct stash offset); the saved exponentials are addressed through ONE descriptor over the whole stash (32-bit scalar
//     offsets: the host takes this kernel for stashes below 4 GiB), so no 64-bit address arithmetic and no descriptor rebuild per tile;
//     reads past the end of a slice need no clamp (in bounds of the buffer or answered with zeros by the descriptor's range check, and
//     what they produce is never multiplied).  Rings are powers of two: stage offsets are `(x + step) & mask`.
//   fragment loads: every B fragment is fetched right behind the last MFMA that reads its predecessor (fragment-major MFMA order) and
//     lands inside the iteration that issues it: one counted wait in front of Q3 (tile b's k-step 0) and one at the end; the saved
//     exponentials of pair j+3 are the LAST VMEM operations of iteration j, so the closing s_waitcnt vmcnt(NE) leaves exactly them in
//     flight and they have a whole iteration to come from HBM.
//   W^T image of mirrored tiles: the 16-byte chunks of a lane group now land in 8 distinct bank groups (bit 2 of the chunk slot is
//     XORed with bit 4 on the writing and on the reading side): the 2-way ds_write_b128 conflicts of the round-3 layout are gone.
//
// LDS: W 2 x 16 KiB | E ring 4 x 4 waves x 4 KiB | statistics ring 4 x 4 waves x 256 B (| k ring) | the block's own statistics = 101 KiB.
#pragma once

namespace crossclr {

#ifndef CROSSCLR_PABL
#define CROSSCLR_PABL 0        // timing ablations (WRONG results): bit0 no E DMA, bit1 no fragment loads, bit2 no weight VALU, bit3 no W write,
                               // bit5 no barrier, bit6 no A reads, bit7 no MFMA, bit8 no VMEM wait in front of Q3, bit9 no closing VMEM wait,
                               // bit10 no LDS waits, bit11 E from a 2-MiB window (L2-resident), bit12 fragments from 8 tiles (always L2 hits)
                               // bit13 every block starts its walk over the column tiles' FRAGMENTS at another tile (L2-channel hot spots?)
                               // MODELS of other block shapes (instruction and byte mix only; profiles/r05_pabl.txt):
                               // bit14 half the fragment loads, bit15 every E piece fetched twice (same address: the second hits L2),
                               // bit16 (with 15) the second from another part of the stash (a second HBM stream), bit17 weights, staged
                               // reads, W writes and A reads twice  -> 14+15(+16)+17 = a 256-row x Dpad/2 block;
                               // bit18 no k-step-1 A reads of mirrored tiles (their fragments kept in registers)
                               // bit19 / bit20: wave w idles 32 w / 64 w cycles behind every barrier (are the four waves' VMEM bursts colliding?)
#endif
#ifndef CROSSCLR_PSPREAD
#define CROSSCLR_PSPREAD 0
#endif
// (Measured and dropped: the saved exponentials straight into VGPRs instead of the per-wave LDS ring -- four buffer_load_dwordx4 per pair
//  in place of four LDS-DMA pieces and four ds_read_b128: 0.250 against 0.232 ms, profiles/r04_pabl.txt.)

// MODE 0: the local symmetric block (xf = this rank's fragment-major operand; rz_cols / wrz_cols / kc = rz / wrz / ks).
// MODE 1 (RECT): this rank's rows x other ranks' columns from a rectangular stash -- every tile direct; xf = the fragment-major copy of the
//   GATHERED operand (crossclr_pack_xf_from_packed), rz_cols / wrz_cols / kc = the gathered statistics; column tile u of the usable
//   ranks maps to memory tile mt(u) (segment walk with the skipped rank and the wrap of crossclr_kernels_dsl.h, division by a reciprocal).
// MODE 2 (TR): the TRANSPOSE of one rectangular block (partner gradients) -- every tile mirrored; output rows = the partner's (rz / wrz /
//   ks = its statistics), xf / rz_cols / wrz_cols / kc = this rank's LOCAL operand and statistics; g.col_ranks = rank segments per stash
//   row, g.skip_rank = the partner's segment in it.
template <int DK, bool SW, int MODE, int XP, int TPRF>
__global__ void __launch_bounds__(256, 1) fast_bwd_xfp_kernel(const unsigned char* xf, const unsigned char* stash, unsigned stash_bytes, Geo g,
                                                              const float* rz, const float* wrz, const float* rz_cols, const float* wrz_cols,
                                                              float* gbuf, int accumulate, int tiles_per_slice, const float* ks, const float* kc) {
    constexpr bool RECT = MODE == 1, TR = MODE == 2;
    constexpr int RB = DK * 32;            // bytes per row of the (part of the) operand a block multiplies
    constexpr int QT = 32;
    constexpr int TPR = TPRF;
    constexpr int RBG = XP * RB;           // bytes per operand row in memory
    constexpr int DI = DK / 8;             // 32-wide output fragments per wave (its column slice)
    constexpr int H = 4 * DI;              // MFMAs per quarter: 4 row groups x DI fragments
    constexpr int NSE = 4;                 // stages of the saved-exponential / statistics rings (pairs): j+1 read, j+2 landing, j+3 issued
    constexpr int WPAIR = 2 * 4 * 2048;    // one W slot: [2 tiles][4 row groups][2 KiB]
    constexpr int ESTG = 4 * 4096;         // one stage of the E ring: [4 waves][2 tiles x 2 KiB]
    constexpr int SSTG = 4 * 256;          // one stage of a statistics ring: [4 waves][64 floats = the pair's columns]
    constexpr int NEO = ((CROSSCLR_PABL & 32768) ? 8 : 4) + 1 + (SW ? 1 : 0);   // VMEM operations per wave and pair behind the fragment loads: E pieces + statistics (+ k)
    constexpr int W0 = 0, E0 = W0 + 2 * WPAIR, S0 = E0 + NSE * ESTG, K0 = S0 + NSE * SSTG;
    constexpr int O0 = K0 + (SW ? NSE * SSTG : 0);      // the block's own statistics: rz[128] | wrz[128] | k[128]
    constexpr int O1 = O0 + 3 * 512;                    // SW: 64 floats of 1.0 -- the negative scales of a tile of the OTHER modality
    static_assert(DK % 8 == 0 && DK >= 8 && DK <= 32, "Dpad (per part) in {128, 256, 384, 512}");
    static_assert(O1 + 256 <= 160 * 1024, "LDS budget");
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[O1 + 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int row0b = blockIdx.x * 128;
    const int row0w = row0b + 32 * wave;
    const int r32 = uniform(row0w >> 5);
    const int per_rank = 2 * g.bpad / QT, per_mod = g.bpad / QT;
    const int skip_seg = (RECT && g.col_wrap == 0 && g.skip_rank >= g.col_rank0 && g.skip_rank < g.col_rank0 + g.col_ranks) ? g.skip_rank - g.col_rank0 : -1;
    const int NT = RECT ? (g.col_ranks - (skip_seg >= 0 ? 1 : 0)) * per_rank : per_rank;
    const int rmod = (2 * r32 >= per_rank) ? 1 : 0;
    const int col_segs = !RECT ? 1 : (g.col_wrap > 0 ? g.col_wrap : g.col_ranks);
    // RECT: usable column tile u -> memory tile (rank segment * per_rank + tile inside the segment) and "columns of the rows' modality";
    // u / per_rank by a 32-bit reciprocal (exact for u * per_rank < 2^32: the host refuses larger launches)
    const unsigned rcp_per_rank = (unsigned)((((unsigned long long)1 << 32) + (unsigned)per_rank - 1u) / (unsigned)per_rank);
    struct Cur { unsigned mt; bool same; };
    auto cur_of = [&](int u) {
        Cur c;
        if constexpr (RECT) {
            const unsigned su = (unsigned)(((unsigned long long)(unsigned)u * rcp_per_rank) >> 32);
            const unsigned in_seg = (unsigned)u - su * (unsigned)per_rank;
            unsigned r = su + ((skip_seg >= 0 && (int)su >= skip_seg) ? 1u : 0u);
            if (g.col_wrap > 0) { r += (unsigned)g.col_rank0; if (r >= (unsigned)g.col_wrap) r -= (unsigned)g.col_wrap; }
            c.mt = r * (unsigned)per_rank + in_seg;
            c.same = ((int)in_seg >= per_mod ? 1 : 0) == rmod;
        } else {
            c.mt = (unsigned)u;
            c.same = (u >= per_mod ? 1 : 0) == rmod;
        }
        return c;
    };
    const int rb0 = (r32 / TPR) * TPR;           // first tile the forward evaluated for this block's rows (the same for its four waves)
    const int part = XP > 1 ? (int)blockIdx.z : 0;
    timing_mark(0);
    // the block's own statistics (mirrored tiles read them as COLUMN statistics)
    if (tid < 128) {
        float* own = reinterpret_cast<float*>(lds + O0);
        own[tid] = rz[row0b + tid];
        own[128 + tid] = wrz[row0b + tid];
        own[256 + tid] = SW ? ks[row0b + tid] : 1.f;
        if (tid < 64) reinterpret_cast<float*>(lds + O1)[tid] = 1.f;
    }
    const float rzp_inter = rz[row0w + l31];
    const float rzp_intra = wrz[row0w + l31];
    const float kp = SW ? ks[row0w + l31] : 1.f;
    __syncthreads();        // (before any LDS-DMA is in flight: this barrier may drain VMEM)
    wait_loads_visible();

    // transpose-read roles: in a 16-lane group lane 4j+c addresses row j, 8-byte piece c
    const int grp = lane >> 4, i16 = lane & 15, jrow = i16 >> 2, piece = i16 & 3, dsub = grp & 1;
    // W image of one tile inside a slot (tile b at +8192).  Direct: group pi at pi*2048, lane-linear fragment (k-step th at +1024*th).
    // Mirrored: 16-byte chunk (th_s, hf, rho) of the stored tile [row rho = a column q of ours, chunk = 8 of OUR rows] at slot
    // 16*(rho>>2) + (rho&3) + 4*(hf ^ ((rho>>2)&1)) + 8*th_s; the reader (lane (half, g1 = dsub, jj = jrow, c = piece)) addresses
    // th*1024 + u*512 + [half*256 + (jj + 4((c&1) ^ half) + 8 g1)*16 + 8(c>>1)].  (The XOR spreads the eight chunks a ds_write_b128
    // lane group stores over the eight 16-byte bank groups; within a transpose read `half` is constant, so its lanes still cover one
    // 256-byte line exactly once.)
    const int wr_dir = wave * 2048 + lane * 16;                                                                   // + 1024*th
    const int wr_mir = wave * 2048 + 16 * (16 * (l31 >> 2) + (l31 & 3) + 4 * (half ^ ((l31 >> 2) & 1)));          // + 128*th
    const int rd_dir = lane * 16;                                                                                 // + pi*2048 + th*1024
    const int rd_mir = half * 256 + (jrow + 4 * ((piece & 1) ^ half) + 8 * dsub) * 16 + 8 * (piece >> 1);         // + pi*2048 + th*1024 (+512)

    // fragment (dt = 4 DI part + DI wave + di, k-step ks) of tile u at u * QT * RBG + (2 dt + ks) * 1024 + 16 lane
    const RawRsrc rs_xf = make_raw_rsrc(xf, (unsigned)((size_t)col_segs * 2 * g.bpad * RBG));
    const unsigned xfv0 = (unsigned)((4 * DI * part + DI * wave) * 2048 + lane * 16), xfv1 = xfv0 + 4096u;
    u32x4 BS[2][DI][2];          // [tile of the pair][fragment][k-step]
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
    const BufRsrc rs_e = make_rsrc(stash, stash_bytes);
    unsigned char* ebuf = lds + E0 + wave * 4096;      // + stage * ESTG (+ 2048: tile b)
    unsigned char* sbuf = lds + S0 + wave * 256;       // + stage * SSTG: omega/Z (or w omega/Z) of the pair's 64 columns
    unsigned char* kbuf = lds + K0 + wave * 256;       // SW: k of the pair's columns

    f32x16 acc[4][DI];
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
#pragma unroll
        for (int di = 0; di < DI; ++di)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pi][di][r] = 0.f;

    const int t0 = blockIdx.y * tiles_per_slice;
    int t_end = t0 + tiles_per_slice;
    if (t_end > NT) t_end = NT;
    // stash tile of this wave for column tile u: (r32, u) where the forward evaluated it for these rows (u >= rb0: index Cd + u),
    // (u, r32) left of that (mirrored)
    // (RECT: the rectangular stash [row group][usable tile]; TR: tile (this rank's row group u, item = partner segment * per_rank + r32))
    const unsigned Cd = RECT ? (unsigned)r32 * (unsigned)NT
                             : TPR * (unsigned)(r32 / TPR) * ((unsigned)NT - (TPR / 2) * (unsigned)(r32 / TPR) + (TPR / 2)) +
                               (unsigned)(r32 % TPR) * ((unsigned)NT - TPR * (unsigned)(r32 / TPR)) - TPR * (unsigned)(r32 / TPR);
    const unsigned tr_stride = (unsigned)(g.col_ranks * per_rank), tr_base = (unsigned)(g.skip_rank * per_rank + r32);
    // byte offsets of the two stash tiles of the pair that starts at tile u (branch-free: M and D pairs select)
    struct EOff { unsigned a, b; };
    auto eoff_of = [&](int u) {
        if constexpr (RECT) { EOff e; e.a = (Cd + (unsigned)u) << 11; e.b = e.a + 2048u; return e; }
        if constexpr (TR) { EOff e; e.a = ((unsigned)u * tr_stride + tr_base) << 11; e.b = e.a + (tr_stride << 11); return e; }
        const unsigned R8 = (unsigned)u & ~(unsigned)(TPR - 1), wp = (unsigned)u & (unsigned)(TPR - 1);
        const unsigned bb = (unsigned)NT - R8;
        const unsigned mir = R8 * ((unsigned)NT + (TPR / 2) - R8 / 2) + wp * bb + ((unsigned)r32 - R8);
        const unsigned dir = Cd + (unsigned)u;
        const bool d = u >= rb0;
        EOff e;
        e.a = (d ? dir : mir) << 11;
        e.b = (d ? dir + 1u : mir + bb) << 11;
        return e;
    };
    // piece k of the pair starting at tile u: k < 4 the saved exponentials (tile k >> 1, k-step k & 1) into ring stage offset `so`; k = 4 / 5 the pair's 64 column statistics / negative scales into the statistics rings (so >> 4)
    auto issue_e = [&](auto kc_, const EOff& e, const Cur& cu, unsigned so) {
        constexpr int k = decltype(kc_)::value;
        if (CROSSCLR_PABL & 1) return;
        if constexpr (k < 4) {
            const unsigned off = ((CROSSCLR_PABL & 2048) ? 0x1FF800u : 0xFFFFFFFFu) & (k < 2 ? e.a : e.b);
            lds_dma16_buf(rs_e, (unsigned)(lane * 16 + 1024 * (k & 1)), off, ebuf + so + 2048 * (k >> 1) + 1024 * (k & 1));
            if (CROSSCLR_PABL & 32768) {
                unsigned off2 = off;
                if (CROSSCLR_PABL & 65536) { off2 = off + (stash_bytes >> 1) & ~2047u; if (off2 >= stash_bytes) off2 -= (stash_bytes >> 1) & ~2047u; }
                lds_dma16_buf(rs_e, (unsigned)(lane * 16 + 1024 * (k & 1)), off2, ebuf + so + 2048 * (k >> 1) + 1024 * (k & 1));
            }
        } else if constexpr (k == 4) {
            lds_dma4_buf(cu.same ? rs_wrz : rs_rz, (unsigned)(lane * 4), cu.mt * (unsigned)(QT * 4), sbuf + (so >> 4));
        } else if constexpr (SW) {
            lds_dma4_buf(rs_k, (unsigned)(lane * 4), cu.mt * (unsigned)(QT * 4), kbuf + (so >> 4));
        }
    };
    auto load_xf = [&](auto setc, auto dic, auto ksc, unsigned so) {
        constexpr int S = decltype(setc)::value, di = decltype(dic)::value, kk = decltype(ksc)::value;
        if (CROSSCLR_PABL & 2) return;
        if ((CROSSCLR_PABL & 16384) && (di & 1)) return;
        constexpr int off = (2 * di + kk) * 1024;
        BS[S][di][kk] = buf_load_b128_async<(off & 4095)>(rs_xf, off >= 4096 ? xfv1 : xfv0, (CROSSCLR_PABL & 4096) ? so % (unsigned)(8 * QT * RBG) : so);
    };

    struct Pair { s16x4 lo, hi; };
    struct Bits8 { bf16_t e[8]; };
    // one tile as this wave weighs it: saved exponentials, row statistic(s) of the lane, column statistics of its 16 columns
    struct Staged { u32x4 e[2]; u32x4 cs[4]; u32x4 kc[4]; unsigned rs, kr; };
    // MIR: the stored tile is E^T -- the lane's row is column q = l31 of the tile, its 16 columns are rows of this wave's group.
    // which = 0 / 1: tile a / b of the pair in ring stage offset `so`; `same`: the pair's columns are of the rows' modality
    auto read_staged = [&](auto mir, auto partc, int which, unsigned so, bool same, Staged& st) {
        constexpr bool MIR = decltype(mir)::value;
        constexpr int P = decltype(partc)::value;
        if constexpr (P == 0) {
            const auto ed = lds_addr(ebuf + so + 2048 * which + 16 * lane);
            st.e[0] = lds_read_b128_async<0>(ed);
            st.e[1] = lds_read_b128_async<1024>(ed);
            // SW: W = E (rs k_q + cs k_r) for EVERY tile -- one multiply and one FMA per element, no select: a tile of the other modality reads
            // its column scales and the row's own scale as 1.0 (the ones at O1), which makes the sum rs + cs exactly
            if (MIR) {
                st.rs = lds_read_b32_async<0>(lds_addr(sbuf + (so >> 4) + 128 * which + 4 * l31));
                if (SW) st.kr = lds_read_b32_async<0>(same ? lds_addr(kbuf + (so >> 4) + 128 * which + 4 * l31) : lds_addr(lds + O1 + 4 * l31));
            } else if (SW) {
                st.kr = __builtin_bit_cast(unsigned, same ? kp : 1.f);
            }
        } else if constexpr (P == 1) {
            // quad (th, r4): columns 16th + 8r4 + 4half ..+3 -- of the tile (direct) or of this wave's own row group (mirrored)
            const auto sa = MIR ? lds_addr(lds + O0 + (same ? 512 : 0) + 128 * wave + 16 * half) : lds_addr(sbuf + (so >> 4) + 128 * which + 16 * half);
            st.cs[0] = lds_read_b128_async<0>(sa);
            st.cs[1] = lds_read_b128_async<32>(sa);
            st.cs[2] = lds_read_b128_async<64>(sa);
            st.cs[3] = lds_read_b128_async<96>(sa);
        } else if constexpr (SW) {
            const auto ka = !same ? lds_addr(lds + O1 + 16 * half)
                                  : (MIR ? lds_addr(lds + O0 + 1024 + 128 * wave + 16 * half) : lds_addr(kbuf + (so >> 4) + 128 * which + 16 * half));
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
    // W for ONE column (k-step th, register quad r4, element j) of the staged tile, packed to bf16 in place (the arithmetic and its
    // order are those of crossclr_kernels_dsl.h: the two kernels produce the same bits).  One element = 3 VALU: a chore per MFMA slot.
    auto weigh1 = [&](auto mir, bool same_mod, const Staged& st, Bits8 (&pk)[2], int th, int r4, int j) {
        constexpr bool MIR = decltype(mir)::value;
        const float rs = MIR ? __builtin_bit_cast(float, st.rs) : (same_mod ? rzp_intra : rzp_inter);
        const float kr = SW ? __builtin_bit_cast(float, st.kr) : 1.f;
        const f32x4 cs = __builtin_bit_cast(f32x4, st.cs[2 * th + r4]);
        f32x4 kq = {1.f, 1.f, 1.f, 1.f};
        if (SW) kq = __builtin_bit_cast(f32x4, st.kc[2 * th + r4]);
        const Bits8 ev = __builtin_bit_cast(Bits8, st.e[th]);
        const float v = bf16_bits_to_f32(ev.e[4 * r4 + j]);
        const float zz = SW ? __builtin_fmaf(rs, kq[j], cs[j] * kr) : (rs + cs[j]);     // (crossclr_kernels_dsl.h: the same expression, the same bits)
        pk[th].e[4 * r4 + j] = (CROSSCLR_PABL & 4) ? ev.e[4 * r4 + j] : f32_to_bf16_bits(v * zz);
    };
    // (model bit17: the same element once more from an opaque copy of the exponential, into a second packed tile)
    auto weigh1_again = [&](auto mir, bool same_mod, const Staged& st, Bits8 (&pk2)[2], int th, int r4, int j) {
        constexpr bool MIR = decltype(mir)::value;
        const float rs = MIR ? __builtin_bit_cast(float, st.rs) : (same_mod ? rzp_intra : rzp_inter);
        const f32x4 cs = __builtin_bit_cast(f32x4, st.cs[2 * th + r4]);
        const Bits8 ev = __builtin_bit_cast(Bits8, st.e[th]);
        float v = bf16_bits_to_f32(ev.e[4 * r4 + j]);
        asm volatile("" : "+v"(v));
        pk2[th].e[4 * r4 + j] = f32_to_bf16_bits(v * (rs + cs[j]));
    };
    // wofs: byte offset of the W slot being WRITTEN (0 / WPAIR); which: tile a / b of the pair
    auto write_w = [&](auto mir, unsigned wofs, int which, const Bits8 (&pk)[2]) {
        constexpr bool MIR = decltype(mir)::value;
        if (CROSSCLR_PABL & 8) return;
        unsigned char* wb = lds + W0 + wofs + 8192 * which;
        if (MIR) {
            *reinterpret_cast<u32x4*>(wb + wr_mir) = __builtin_bit_cast(u32x4, pk[0]);
            *reinterpret_cast<u32x4*>(wb + wr_mir + 128) = __builtin_bit_cast(u32x4, pk[1]);
        } else {
            *reinterpret_cast<u32x4*>(wb + wr_dir) = __builtin_bit_cast(u32x4, pk[0]);
            *reinterpret_cast<u32x4*>(wb + wr_dir + 1024) = __builtin_bit_cast(u32x4, pk[1]);
        }
    };

    bf16x8 A1c[4];     // k-step 1 of the previous pair's second tile: read before the barrier, multiplied after it
    {
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) A1c[pi] = z;
    }

    if (t0 < t_end) {
        // tiles [t0, tm) are mirrored, [tm, t_end) direct (all even); a rectangular block's tiles are all direct, its transpose's all mirrored
        const int tm = RECT ? t0 : (TR ? t_end : (t_end < rb0 ? t_end : (rb0 > t0 ? rb0 : t0)));
        int t = t0;                                                       // first tile of the current pair
        // ---- prologue: the saved exponentials / statistics of pairs 0, 1, 2 (stages 0, 1, 2), then the fragments of tile t0 (set 0) and
        // ONE full wait -- the fragment loads and their wait sit in one basic block behind every branch of the address arithmetic
        // (tools/asm_audit.py: no asm-loaded register in flight at a control-flow point); pair 0 is weighed at once
        static_for<3>([&](auto kc_) {
            constexpr int k = decltype(kc_)::value;
            const EOff e = eoff_of(t + 2 * k);
            const Cur cu = cur_of(t + 2 * k);
            static_for<NEO>([&](auto jc) { issue_e(jc, e, cu, (unsigned)(k * ESTG)); });
        });
        {
            unsigned so = cur_of(t).mt * (unsigned)(QT * RBG);
            pin_s(so);
            static_for<2 * DI>([&](auto jc) { constexpr int j = decltype(jc)::value; load_xf(IdxC<0>{}, IdxC<j / 2>{}, IdxC<j % 2>{}, so); });
        }
        if (!(CROSSCLR_PABL & 3)) wait_dma();
#pragma unroll
        for (int di = 0; di < DI; ++di) { after_wait(BS[0][di][0]); after_wait(BS[0][di][1]); }
        {
            const bool same = cur_of(t).same;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                Bits8 pk[2];
                Staged st;
                if (t < tm) {
                    static_for<3>([&](auto pc) { read_staged(IdxC<true>{}, pc, which, 0u, same, st); });
                    wait_lgkm_all();
                    staged_landed(IdxC<true>{}, st);
#pragma unroll
                    for (int q = 0; q < 16; ++q) weigh1(IdxC<true>{}, same, st, pk, q >> 3, (q >> 2) & 1, q & 3);
                    write_w(IdxC<true>{}, 0u, which, pk);
                } else {
                    static_for<3>([&](auto pc) { read_staged(IdxC<false>{}, pc, which, 0u, same, st); });
                    wait_lgkm_all();
                    staged_landed(IdxC<false>{}, st);
#pragma unroll
                    for (int q = 0; q < 16; ++q) weigh1(IdxC<false>{}, same, st, pk, q >> 3, (q >> 2) & 1, q & 3);
                    write_w(IdxC<false>{}, 0u, which, pk);
                }
            }
        }
        barrier_keep_dma();
        timing_mark(1);

        // loop-carried scalars of iteration j (pair at tile t): all bumped by constants in ONE slot
        unsigned xo_b = (cur_of(t).mt + 1u) * (unsigned)(QT * RBG);   // fragment offset of tile b; local block: tile a' = xo_b + QT * RBG
        const unsigned xo_total = (unsigned)NT * (unsigned)(QT * RBG);
        if (CROSSCLR_PABL & 8192) { xo_b += (unsigned)(2 * ((blockIdx.x * 7 + blockIdx.y * 3) % (NT / 2))) * (unsigned)(QT * RBG); if (xo_b >= xo_total) xo_b -= xo_total; }
        // RECT: the cursors of the pairs at t + 2 (fragments of tile a', weights), t + 4, t + 6 (statistics DMA) travel with the loop -- one
        // reciprocal division per iteration (the pair at t + 8)
        Cur c2 = cur_of(t + 2), c4 = cur_of(t + 4), c6 = cur_of(t + 6);
        unsigned e_rd = (unsigned)ESTG;                               // ring stage offset of pair j+1 (read); pair j+3 goes to e_rd ^ 2 ESTG
        unsigned wofs = 0;                                            // W slot of the current pair (read); the next pair is written to wofs ^ WPAIR
        pin_s(xo_b); pin_s(e_rd); pin_s(wofs);

        // One iteration = 4 H MFMA slots { MFMA ; a share of the chores }, pinned by sched_fence() on both sides of the chores.
        auto body = [&](auto mc_, auto mn_) {
            constexpr bool MC = decltype(mc_)::value, MN = decltype(mn_)::value;
            u32x4 Ad[2][4];            // direct A fragments: [buffer][row group]   (buffer 0: quarters 1 and 3, buffer 1: quarter 2 and the carry)
            Pair Am[2][4];             // mirrored
            Staged st;
            Bits8 pk[2];
            Bits8 pk2[2];
            const Cur cn = RECT ? c2 : cur_of(t + 2);               // the pair being weighed
            const Cur cd = RECT ? c6 : cur_of(t + 6);               // the pair whose exponentials / statistics this iteration requests
            const bool same_n = cn.same;
            const unsigned wa_rd = wofs, wa_wr = wofs ^ (unsigned)WPAIR;
            unsigned xo_a2 = RECT ? cn.mt * (unsigned)(QT * RBG) : xo_b + (unsigned)(QT * RBG);
            if ((CROSSCLR_PABL & 8192) && xo_a2 >= xo_total) xo_a2 -= xo_total;
            const unsigned e_wr = e_rd ^ (unsigned)(2 * ESTG);
            const EOff enext = eoff_of(t + 6);
            const auto wa = lds_addr(lds + W0 + wa_rd + (MC ? rd_mir : rd_dir));
            // A fragments (row group pi, k-step th) of tile `which` of the current pair into buffer BUF
            auto read_a = [&](auto bufc, auto pic, auto thc, auto whichc) {
                constexpr int BUF = decltype(bufc)::value, pi = decltype(pic)::value, th = decltype(thc)::value, wh = decltype(whichc)::value;
                if (CROSSCLR_PABL & 64) return;
                if ((CROSSCLR_PABL & 262144) && MC && th == 1) return;
#pragma unroll
                for (int rep = 0; rep < ((CROSSCLR_PABL & 131072) ? 2 : 1); ++rep) {
                    if (MC) {
                        Am[BUF][pi].lo = lds_read_tr16_b64_async<wh * 8192 + pi * 2048 + th * 1024>(wa);
                        Am[BUF][pi].hi = lds_read_tr16_b64_async<wh * 8192 + pi * 2048 + th * 1024 + 512>(wa);
                    } else {
                        Ad[BUF][pi] = lds_read_b128_async<wh * 8192 + pi * 2048 + th * 1024>(wa);
                    }
                }
            };
            auto a_landed = [&](auto bufc) {
                constexpr int BUF = decltype(bufc)::value;
#pragma unroll
                for (int p = 0; p < 4; ++p) { if (MC) { after_wait(Am[BUF][p].lo); after_wait(Am[BUF][p].hi); } else after_wait(Ad[BUF][p]); }
            };
            auto a_frag = [&](auto bufc, int pi) {
                constexpr int BUF = decltype(bufc)::value;
                return MC ? __builtin_bit_cast(bf16x8, Am[BUF][pi]) : __builtin_bit_cast(bf16x8, Ad[BUF][pi]);
            };
            // chore placement inside a quarter: item i of n over the slots [lo, hi)
            static_for<4 * H>([&](auto gc) {
                constexpr int gs = decltype(gc)::value, q = gs / H, s = gs % H;
                constexpr int pi = s % 4, di = s / 4;                 // fragment-major: fragment di is dead after its 4 MFMAs
                // ---- the MFMA of this slot
                if constexpr (q == 0) mfma_acc(acc[pi][di], A1c[pi], __builtin_bit_cast(bf16x8, BS[1][di][1]));
                else if constexpr (q == 1) mfma_acc(acc[pi][di], a_frag(IdxC<0>{}, pi), __builtin_bit_cast(bf16x8, BS[0][di][0]));
                else if constexpr (q == 2) mfma_acc(acc[pi][di], a_frag(IdxC<1>{}, pi), __builtin_bit_cast(bf16x8, BS[0][di][1]));
                else mfma_acc(acc[pi][di], a_frag(IdxC<0>{}, pi), __builtin_bit_cast(bf16x8, BS[1][di][0]));
                sched_fence();
                // ---- the pair being weighed: tile a' over Q0 + Q1, tile b' over Q2 + Q3 (h2 = slot inside that half of the iteration).
                // Staged reads in the first slots (the exponentials landed an iteration ago), ONE wait three slots later (everything it
                // covers is ~100 cycles old), then the tile's 16 elements -- 3 VALU each -- one or two per slot up to the write.
                {
                    constexpr int h2 = gs % (2 * H), wh = gs / (2 * H);
                    constexpr int NS1 = 2 + (SW ? 1 : 0);
                    // CROSSCLR_PSPREAD 0 (default): reads in the first quarter of the half, their wait = that quarter's closing wait, the 16
                    // elements over the second quarter (measured: 0.234 ms at B = 8192, D = 512).  1: reads in slots 0.., an extra wait three
                    // slots later, elements over both quarters (fewer VALU per slot, but 0.246 ms: the extra wait costs more than the
                    // evener stream buys -- a quarter with 4.4 fillers per MFMA is still inside the 5 that hide behind an MFMA).
                    constexpr bool SPREAD = CROSSCLR_PSPREAD != 0;
                    constexpr int RS0 = SPREAD ? 0 : 1;
                    constexpr int WS = SPREAD ? (NS1 + 1 < 2 * H - 3 ? NS1 + 1 : (2 * H - 3 > 0 ? 2 * H - 3 : 0)) : H - 1;   // slot of the wait
                    constexpr int WR = 2 * H - 2;                                                            // slot of the W write
                    constexpr int LW = WR - WS - 1 > 0 ? WR - WS - 1 : 1;                                    // slots WS+1 .. WR-1 carry the elements
                    if constexpr (SPREAD && h2 == WS) { if (!(CROSSCLR_PABL & 1024)) wait_lgkm_all(); staged_landed(IdxC<MN>{}, st); }
                    static_for<NS1>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        if constexpr ((RS0 + i < WS ? RS0 + i : WS - 1) == h2 && h2 < WS + (SPREAD ? 0 : 1)) {
                            read_staged(IdxC<MN>{}, ic, wh, e_rd, same_n, st);
                            if (CROSSCLR_PABL & 131072) read_staged(IdxC<MN>{}, ic, wh, e_rd, same_n, st);
                        }
                    });
                    static_for<16>([&](auto qc) {
                        constexpr int k = decltype(qc)::value;
                        constexpr int at = WS + 1 + (k * LW) / 16 < WR ? WS + 1 + (k * LW) / 16 : WR;
                        if constexpr (at == h2) {
                            weigh1(IdxC<MN>{}, same_n, st, pk, k >> 3, (k >> 2) & 1, k & 3);
                            if (CROSSCLR_PABL & 131072) weigh1_again(IdxC<MN>{}, same_n, st, pk2, k >> 3, (k >> 2) & 1, k & 3);
                        }
                    });
                    if constexpr (h2 == WR) {
                        write_w(IdxC<MN>{}, wa_wr, wh, pk);
                        if (CROSSCLR_PABL & 131072) write_w(IdxC<MN>{}, wa_wr, wh, pk2);
                    }
                }
                // ---- A reads of the NEXT quarter: 4 items from slot 1 over the first 3/4 of the quarter (complete well ahead of the
                // boundary wait; behind the staged wait in program order: it must not cover a read issued in its own slot)
                constexpr int LA = (3 * H) / 4 > 0 ? (3 * H) / 4 : 1;
                static_for<4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr ((1 + (i * LA) / 4 < H ? 1 + (i * LA) / 4 : H - 1) == s) {
                        if constexpr (q == 0) read_a(IdxC<0>{}, ic, IdxC<0>{}, IdxC<0>{});          // tile a, k-step 0 -> Q1
                        else if constexpr (q == 1) read_a(IdxC<1>{}, ic, IdxC<1>{}, IdxC<0>{});     // tile a, k-step 1 -> Q2
                        else if constexpr (q == 2) read_a(IdxC<0>{}, ic, IdxC<0>{}, IdxC<1>{});     // tile b, k-step 0 -> Q3
                        else read_a(IdxC<1>{}, ic, IdxC<1>{}, IdxC<1>{});                           // tile b, k-step 1 -> carried to the next Q0
                    }
                });
                // ---- fragment loads: each right behind the last MFMA that read its predecessor
                static_for<DI>([&](auto dc) {
                    constexpr int d = decltype(dc)::value;
                    if constexpr (gs == d) load_xf(IdxC<1>{}, dc, IdxC<0>{}, xo_b);                  // tile b, k-step 0 (dead since the last Q3)
                    if constexpr (gs == 4 * (d + 1)) load_xf(IdxC<1>{}, dc, IdxC<1>{}, xo_b);        // tile b, k-step 1 (behind Q0's use)
                    if constexpr (gs == H + 4 * (d + 1)) load_xf(IdxC<0>{}, dc, IdxC<0>{}, xo_a2);   // tile a', k-step 0 (behind Q1's use)
                    if constexpr (gs == 2 * H + 4 * (d + 1)) load_xf(IdxC<0>{}, dc, IdxC<1>{}, xo_a2);   // tile a', k-step 1 (behind Q2's use)
                });
                // ---- saved exponentials / statistics of pair j+3: the LAST VMEM operations of the iteration
                if constexpr (q == 3) {
                    static_for<NEO>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        constexpr int L3 = H - 2 > 1 ? H - 2 : 1;
                        if constexpr (1 + (k * L3) / NEO == s || (1 + (k * L3) / NEO > H - 1 && s == H - 1)) issue_e(kc, enext, cd, e_wr);
                    });
                }
                // ---- quarter boundaries
                if constexpr (s == H - 1) {
                    if constexpr (q < 3) {
                        if (!(CROSSCLR_PABL & 1024)) wait_lgkm_all();
                        if constexpr (q == 0) { a_landed(IdxC<0>{}); if constexpr (CROSSCLR_PSPREAD == 0) staged_landed(IdxC<MN>{}, st); }
                        else if constexpr (q == 1) a_landed(IdxC<1>{});
                        else {
                            a_landed(IdxC<0>{});
                            if constexpr (CROSSCLR_PSPREAD == 0) staged_landed(IdxC<MN>{}, st);
                            // tile b's k-step-0 fragments: 3 DI - 1 VMEM operations were issued behind the last of them
                            if (!(CROSSCLR_PABL & (3 | 256))) wait_dma_keep<((CROSSCLR_PABL & 16384) ? (3 * DI) / 2 : 3 * DI - 1)>();
#pragma unroll
                            for (int d = 0; d < DI; ++d) after_wait(BS[1][d][0]);
                        }
                    }
                }
                sched_fence();
            });
            // every LDS operation of the iteration is complete: W(j+1) written, tile b's k-step-1 fragments read ...
            if (!(CROSSCLR_PABL & 1024)) wait_lgkm_all();
            a_landed(IdxC<1>{});
#pragma unroll
            for (int p = 0; p < 4; ++p) A1c[p] = a_frag(IdxC<1>{}, p);
            // ... and every fragment load has landed (only the NEO operations issued behind the last of them stay in flight)
            if (!(CROSSCLR_PABL & (3 | 512))) wait_dma_keep<NEO>();
#pragma unroll
            for (int d = 0; d < DI; ++d) { after_wait(BS[0][d][0]); after_wait(BS[0][d][1]); after_wait(BS[1][d][1]); }
            if constexpr (RECT) {       // (behind the closing waits: scalar arithmetic only, no asm load in flight)
                xo_b = (c2.mt + 1u) * (unsigned)(QT * RBG);
                c2 = c4; c4 = c6; c6 = cur_of(t + 8);
            } else {
                xo_b += (unsigned)(2 * QT * RBG);
                if ((CROSSCLR_PABL & 8192) && xo_b >= xo_total) xo_b -= xo_total;
            }
            t += 2;
            e_rd = (e_rd + (unsigned)ESTG) & (unsigned)(NSE * ESTG - 1);
            wofs ^= (unsigned)WPAIR;
            if (!(CROSSCLR_PABL & 32)) barrier_keep_dma();
#ifndef CROSSCLR_EMU
            if (CROSSCLR_PABL & 524288) { for (int i = 0; i < wave; ++i) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); } }
            if (CROSSCLR_PABL & 1048576) { for (int i = 0; i < wave; ++i) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(1); } }
#endif
        };
        if constexpr (TR) {            // every tile mirrored (the last iteration weighs a pair past the end: never consumed)
            while (t < t_end) body(IdxC<true>{}, IdxC<true>{});
        } else {
            if constexpr (!RECT) {     // (a rectangular block's tiles are all direct)
                while (t + 2 < tm) body(IdxC<true>{}, IdxC<true>{});
                if (t < tm) body(IdxC<true>{}, IdxC<false>{});
            }
            while (t < t_end) body(IdxC<false>{}, IdxC<false>{});
        }
        // k-step 1 of the last tile
        static_for<H>([&](auto sc) { constexpr int s = decltype(sc)::value; mfma_acc(acc[s % 4][s / 4], A1c[s % 4], __builtin_bit_cast(bf16x8, BS[1][s / 4][1])); });
        wait_dma();   // the prefetches past the end must not outlive the block's LDS
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

}  // namespace crossclr
