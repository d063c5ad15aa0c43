// crossclr_kernels_sym.h -- the software-pipelined forward (Dpad <= 1024): symmetric local block, rectangular remote blocks, pairs.
//
// The persistent forward (reference trainer/loss.py:83-100, 59-60); workspace layouts: FwdWork / fwd_finish_kernel:
//   KIND 1  symmetric: rows and columns are the same operand; upper triangle of the stacked 2b x 2b matrix of exponentials,
//           the mirrored half recovered from column sums                                  (single device / local block)
//   KIND 2  rectangular: this rank's rows against other ranks' columns (segments in memory order, one rank skipped)
//   KIND 3  pairs: rectangular over ranks col_rank0, col_rank0+1, ... (mod col_wrap) of the whole gathered operand, and EVERY
//           tile also yields its column sums over this rank's rows (what the column ranks would otherwise compute)
// built like this:
//   * 4 waves x 64 rows per 256-row block, ONE wave per SIMD (512 registers): both 32-row halves' fragments stay resident
//     (2 x Dpad/4 registers), so one ds_read_b128 of the column tile feeds TWO MFMAs (half the LDS reads of a 32-row wave);
//   * the epilogue of tile t-1 (scale, exp2, row sums, bf16 pack + stash stores, 64-row column sums) is cut into chores that
//     are pinned between the 64 MFMAs of tile t (sched_fence per k-step): with one wave per SIMD nothing else would hide them;
//   * every LDS read is asm with hand-counted lgkmcnt, the barrier does not drain VMEM, the column tiles arrive by
//     buffer-addressed LDS-DMA (NST-deep ring) -- hipcc puts no wait of its own into the MFMA stream;
//   * tiles that need a mask (the 256x256 diagonal blocks, ragged columns, padding rows) and the sample-weight variant take the
//     plain, un-overlapped epilogue: 8 of ~65 tiles per row block.
// ST: the bf16 exponentials of every evaluated tile are saved for fast_bwd_dsl_kernel -- KIND 1: the triangular layout of
// stash_tile_index; KIND 2/3: rectangular, tile (32-row group r32, item j) at (r32 * NT + j) * 2 KiB.
#pragma once

namespace crossclr {

// NH = 32-row halves per wave: 2 for Dpad <= 512 (256-row blocks); 1 for 512 < Dpad <= 1024 (the fragments of 64 rows would
// not fit the register file: 128-row blocks, one MFMA per LDS read, a 2-deep tile ring of 64-KiB tiles).
template <int DK, int KIND, bool SW, bool ST, int NH = 2>
__global__ void __launch_bounds__(256, 1) fast_fwd_pipe_kernel(const bf16_t* x, const bf16_t* xc, Geo g, FwdWork wk, float* part,
                                                               float* colpart, int* header, const float* ks, const float* kc,
                                                               unsigned char* stash, FwdPerm perm) {
    static_assert(KIND >= 1 && KIND <= 3, "1 symmetric, 2 rectangular, 3 pairs");
    constexpr int RB = DK * 32;            // bytes per operand row
    constexpr int QT = 32;
    constexpr int TILE = QT * RB;
    constexpr int TPR = 4 * NH;            // 32-row groups per row block (the workspace / stash layout)
    constexpr int RW = 32 * NH;            // rows per wave
    constexpr int RBLK = 4 * RW;           // rows per block
    constexpr int NE = 16 * NH;            // exponentials per lane and tile
    constexpr int NST = (4 * TILE + 4096 <= 160 * 1024) ? 4 : ((3 * TILE + 4096 <= 160 * 1024) ? 3 : 2);
    constexpr int NXO = DK / 4;            // DMA pieces per wave and tile
    constexpr int PF = 4;                  // A-fragment reads in flight ahead of their MFMA pair
    constexpr int CS0 = NST * TILE;        // two column-sum slots [4 waves][32] floats
    constexpr int KQ0 = CS0 + 2 * 4 * QT * 4;
    static_assert(NH == 1 || NH == 2, "one or two 32-row halves per wave");
    static_assert(DK % 8 == 0 && DK >= 8 && DK <= 32 * (3 - NH), "Dpad in {128, 256, 384, 512}; NH = 1: up to 1024");
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[KQ0 + (SW ? NST * 256 : 0)];   // k_q ring: 64 lanes x 4 B per stage (32 used)

    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    timing_mark(0);
    if (blockIdx.x == 0 && tid == 0) { header[0] = wk.kind; header[1] = TPR; header[2] = wk.NT; header[3] = wk.per; }
    const int NT = wk.NT;                      // KIND 1: column tiles of the operand; KIND 2/3: usable column tiles
    const int per_rank = 2 * g.bpad / QT, per_mod = g.bpad / QT;
    const int skip_seg = (KIND == 2 && g.skip_rank >= g.col_rank0 && g.skip_rank < g.col_rank0 + g.col_ranks) ? g.skip_rank - g.col_rank0 : -1;

    // the work range of this block (XCD-aware placement: fwd_make_perm; everything below is indexed by the RANGE, not by blockIdx)
    const int vb = uniform(blockIdx.x < 256 ? (int)perm.v[blockIdx.x < 256 ? blockIdx.x : 0] : (int)blockIdx.x);
    int w = fwd_block_begin(wk, vb);                        // (ranges of equal cost, not of equal length: FwdWork)
    const int w_end = fwd_block_begin(wk, vb + 1);
    if (w >= w_end) return;                                 // (a trailing range that holds no item's first unit)
    // item = (row block rb, index j inside its tile list); mt = the tile's index inside the column operand (DMA / statistics
    // address), seg / in_seg = its rank segment and position inside the segment (modality, ragged test) -- tracked
    // incrementally: no division per tile
    struct Cursor { int rb, j, mt, seg, in_seg; };
    auto seg_start = [&](Cursor& c) {          // c.seg is set: first tile of that segment
        c.in_seg = 0;
        // col_wrap = 0: the operand holds exactly the launch's segments, in order; col_wrap = W: it is the whole gathered array
        // and segment i is rank (col_rank0 + i) mod W
        if (KIND != 1) {
            int r = c.seg;
            if (g.col_wrap > 0) { r += g.col_rank0; if (r >= g.col_wrap) r -= g.col_wrap; }
            c.mt = r * per_rank;
        }
    };
    auto row_start = [&](Cursor& c) {          // c.rb is set: first item of that row block
        c.j = 0;
        if (KIND == 1) { c.mt = TPR * c.rb; c.seg = 0; c.in_seg = c.mt; }
        else { c.seg = (skip_seg == 0) ? 1 : 0; seg_start(c); }
    };
    auto advance = [&](Cursor& c) {
        ++c.j; ++c.mt; ++c.in_seg;
        if (c.j == (KIND == 1 ? NT - TPR * c.rb : NT)) { ++c.rb; row_start(c); return; }
        if (KIND != 1 && c.in_seg == per_rank) { ++c.seg; if (c.seg == skip_seg) ++c.seg; seg_start(c); }
    };
    Cursor cq[NST];
    {
        int rb = 0;
        while (fwd_prefix(wk, rb + 1) <= w) ++rb;
        cq[0].rb = rb;
        row_start(cq[0]);
        const int j0 = w - fwd_prefix(wk, rb);
        if (KIND == 1) { cq[0].j = j0; cq[0].mt += j0; cq[0].in_seg = cq[0].mt; }
        else {
            int su = j0 / per_rank;                        // usable segments before the item (once per block)
            cq[0].seg = su + ((skip_seg >= 0 && su >= skip_seg) ? 1 : 0);
            seg_start(cq[0]);
            cq[0].j = j0; cq[0].in_seg = j0 - su * per_rank; cq[0].mt += cq[0].in_seg;
        }
    }
#pragma unroll
    for (int k = 1; k < NST; ++k) { cq[k] = cq[k - 1]; advance(cq[k]); }

    // column-tile DMA: piece k of this wave fills LDS bytes [(wave + 4k) KiB, +1 KiB) of the stage
    const int col_segs = KIND == 1 ? 1 : (g.col_wrap > 0 ? g.col_wrap : g.col_ranks);     // rank segments the column operand holds
    const BufRsrc rs_x = make_rsrc(xc, (unsigned)((size_t)col_segs * 2 * g.bpad * RB));
    const BufRsrc rs_k = make_rsrc(SW ? (const void*)kc : (const void*)xc, (unsigned)((size_t)col_segs * 2 * g.bpad * 4));
    unsigned voffx[NXO];
#pragma unroll
    for (int k = 0; k < NXO; ++k) {
        const int L = (wave + 4 * k) * 1024 + lane * 16;
        const int row = L / RB, slot = (L - row * RB) >> 4;
        voffx[k] = (unsigned)(row * RB + (swz_slot(slot, row) << 4));
    }
    // (tiles past the end of the work list are clamped to the last column tile: a harmless re-fetch into a free stage that
    // keeps the VMEM count per iteration constant and the k-step chain free of branches)
#ifndef CROSSCLR_YABL
#define CROSSCLR_YABL 0   // timing ablations of this kernel (WRONG results): bit0 every block streams the same 64 column tiles (L2-resident),
                          // bit1 no stash stores, bit2 no column-sum butterfly, bit3 plain epilogue for every tile (no overlap), bit4 no DMA,
                          // bit5 no epilogue at all, bit6 no barrier, bit7 every second LDS read of the column tile skipped
#endif
    const int mt_last = col_segs * per_rank - 1;
    auto tile_of = [&](const Cursor& c) { return (CROSSCLR_YABL & 1) ? (c.mt & 63) : (c.mt < mt_last ? (c.mt < 0 ? 0 : c.mt) : mt_last); };
    auto issue_piece = [&](const Cursor& c, int stage, int k) {
        if (CROSSCLR_YABL & 16) return;
        lds_dma16_buf(rs_x, voffx[k], (unsigned)tile_of(c) * (unsigned)TILE, lds + stage * TILE + (wave + 4 * k) * 1024);
    };
    auto issue_stat = [&](const Cursor& c, int stage) {     // SW: the tile's 32 k_q (every wave issues it: equal VMEM counts)
        if (SW) lds_dma4_buf(rs_k, (unsigned)((lane & 31) * 4), (unsigned)(tile_of(c) * QT * 4), lds + KQ0 + stage * 256);
    };
    constexpr int NOPS = NXO + (SW ? 1 : 0);
#pragma unroll
    for (int k = 0; k < NST - 1; ++k) {
#pragma unroll
        for (int p = 0; p < NXO; ++p) issue_piece(cq[k], k, p);
        issue_stat(cq[k], k);
    }

    int off8[8];  // byte offset of logical chunk (2j + half) of this lane's tile row
#pragma unroll
    for (int j = 0; j < 8; ++j) off8[j] = l31 * RB + ((((2 * j + half) ^ sigma16(l31)) & 15) << 4);

    float* cs = reinterpret_cast<float*>(lds + CS0);
    // ---- per-row-block state ----
    int my_rb = -1, row0w = 0, rmod = 0;
    float rowacc[NH], kp[NH];
    size_t st0[NH];                        // ST: stash index of tile j = 0 of the current row block, per 32-row half
#pragma unroll
    for (int s = 0; s < NH; ++s) { rowacc[s] = 0.f; kp[s] = 1.f; st0[s] = 0; }
    bf16x8 pf[NH][DK];
    auto store_rows = [&]() {
#pragma unroll
        for (int s = 0; s < NH; ++s) {
            const float v = rowacc[s] + wave_xor_f32(rowacc[s], 32);
            if (half == 0) part[(size_t)(vb - fwd_first_block(wk, my_rb)) * 2 * g.bpad + row0w + 32 * s + l31] = v;
        }
    };
    // ---- the tile whose epilogue is still owed ----
    struct Prev { bool valid, fast; int j, cmod, in_mod0, stage, crank; };   // (the owed tile always belongs to the current row block)
    Prev prev = {false, false, 0, 0, 0, 0, 0};
    f32x16 pacc[NH];
    // ---- column sums waiting for the next barrier ----
    bool pending = false;
    int ptile = 0, pbuf = 0, prb = 0;
    auto flush = [&]() {
        if (tid < QT) {
            const float* c = cs + pbuf * (4 * QT);
            colpart[(size_t)prb * NT * QT + QT * ptile + tid] = (c[tid] + c[QT + tid]) + (c[2 * QT + tid] + c[3 * QT + tid]);
        }
    };
    struct Bits8 { bf16_t v[8]; };
    auto stash_store = [&](int j, int s, const float (&e)[16]) {     // (my_rb is the row block of the tile being finished)
        const BufRsrc rs_st = make_rsrc(stash + (st0[s] + (size_t)j) * 2048, 2048u);
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            Bits8 pk;
#pragma unroll
            for (int q = 0; q < 8; ++q) pk.v[q] = f32_to_bf16_bits(e[8 * th + q]);
            buf_store16(rs_st, (unsigned)(lane * 16 + 1024 * th), 0u, __builtin_bit_cast(u32x4, pk));
        }
    };
    // position of item j's 32 column sums inside a colpart row: KIND 1 the tile's place in the operand, KIND 3 in the pair range
    auto colpos = [&](int j) { return KIND == 1 ? TPR * my_rb + j : j; };
    auto wants_colsum = [&](int j) { return KIND == 3 || (KIND == 1 && j >= TPR); };   // KIND 1: strictly right of the diagonal block
    auto colsum_publish = [&](const float (&es)[16], int j) {
        const float colsum = halving_sum16(es, l31);
        pbuf ^= 1;
        if (l31 < 16) cs[pbuf * (4 * QT) + wave * QT + frag_row(halving_elem16(l31), half)] = colsum;
        pending = true;
        ptile = colpos(j);
        prb = my_rb;
    };
    // general (masked / weighted) epilogue of one tile, not overlapped with anything
    auto epilogue_plain = [&](f32x16 (&acc)[NH], const Prev& pv) {
        if (CROSSCLR_YABL & 32) { rowacc[0] += acc[0][0] + acc[NH - 1][1]; return; }
        const bool same_mod = pv.cmod == rmod;
        const float c2s = same_mod ? g.c_intra : g.c_inter;
        const bool upper = wants_colsum(pv.j);
        const float ninf = -__builtin_inff();
        float es[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) es[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NH; ++s) {
            const int r_in_mod = row0w + 32 * s - rmod * g.bpad + l31;
            float xx[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xx[r] = acc[s][r] * c2s - g.m2;
            if (pv.in_mod0 + QT > g.b) {                    // ragged tile: columns beyond the valid batch
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (pv.in_mod0 + frag_row(r, half) >= g.b) xx[r] = ninf;
            }
            // the tile that holds this half's self pairs: KIND 1 t == r32; a rectangular launch that includes the rows' own rank
            const bool diag_here = KIND == 1 ? (pv.j == NH * wave + s)
                                             : (pv.crank == g.row_rank && same_mod && pv.in_mod0 == row0w + 32 * s - rmod * g.bpad);
            if (diag_here) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (frag_row(r, half) == l31) xx[r] = ninf;
            }
            if (upper && r_in_mod >= g.b) {                  // padding ROWS must not reach the column sums
#pragma unroll
                for (int r = 0; r < 16; ++r) xx[r] = ninf;
            }
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = fast_exp2(xx[r]);
            if (ST) stash_store(pv.j, s, e);
            if (SW && same_mod) {
                const float* kq = reinterpret_cast<const float*>(lds + KQ0 + pv.stage * 256);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 k4 = *reinterpret_cast<const f32x4*>(kq + 8 * r4 + 4 * half);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        rowacc[s] += e[4 * r4 + q] * k4[q];
                        es[4 * r4 + q] += e[4 * r4 + q] * kp[s];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { rowacc[s] += e[r]; es[r] += e[r]; }
            }
        }
        if (upper) colsum_publish(es, pv.j);
    };

    int stage = 0;
    timing_mark(1);
    while (w < w_end) {
        if (cq[0].rb != my_rb) {   // (re)load this wave's 64 rows as MFMA B fragments; first settle what is owed to the old rows
            if (prev.valid) {   // (SW tiles never stay owed: see below)
                // the owed tile may publish column sums while an earlier publication still waits for its barrier: flush that first
                __syncthreads();
                if (pending) { flush(); pending = false; }
                epilogue_plain(pacc, prev);
                prev.valid = false;
            }
            if (my_rb >= 0) store_rows();
            my_rb = cq[0].rb;
#pragma unroll
            for (int s = 0; s < NH; ++s) rowacc[s] = 0.f;
            row0w = my_rb * RBLK + RW * wave;
            rmod = row0w / g.bpad;
#pragma unroll
            for (int s = 0; s < NH; ++s) {
                if (SW) kp[s] = ks[row0w + 32 * s + l31];
                if (ST) st0[s] = KIND == 1 ? stash_tile_index(TPR, NT, TPR * my_rb + NH * wave + s, TPR * my_rb)
                                           : (size_t)(TPR * my_rb + NH * wave + s) * (size_t)NT;
                const bf16_t* src = x + (size_t)(row0w + 32 * s + l31) * (DK * 16) + 8 * half;
#pragma unroll
                for (int k = 0; k < DK; ++k) pf[s][k] = *reinterpret_cast<const bf16x8*>(src + 16 * k);
            }
            wait_loads_visible();   // here, once per row block -- not as vmcnt countdowns inside every tile's MFMA stream
        }
        wait_dma_keep<(NST - 2) * NOPS>();   // tile w has landed (the NST-2 tiles issued after it may still be in flight) ...
        if (!(CROSSCLR_YABL & 64)) barrier_keep_dma();   // ... everywhere; and every wave is done with tile w-1's stage
        if (pending) { flush(); pending = false; }
        const int rstage = (stage + NST - 1) % NST;
        const auto xa = lds_addr(lds + stage * TILE);
        decltype(lds_addr(lds)) abase[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) abase[j] = xa + off8[j];

        f32x16 acc[NH];
#pragma unroll
        for (int s = 0; s < NH; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
        u32x4 ring[PF];
        auto fetch = [&](auto ic) {
            constexpr int k = decltype(ic)::value;
            if ((CROSSCLR_YABL & 128) && (k & 1)) { ring[k % PF] = ring[(k + PF - 1) % PF]; return; }   // (ablation: half the LDS reads)
            ring[k % PF] = lds_read_b128_async<(k >> 3) * 256>(abase[k & 7]);
        };
        // one k-step = two half-slots, each opened by an MFMA (NH == 2: one per row half; NH == 1: the second is empty): wait for
        // the A fragment; MFMA; the read PF k-steps ahead + chore A; MFMA; this step's DMA piece + chore B.  One wave per SIMD
        // issues in order, so whatever follows an MFMA runs in its shadow -- and an MFMA directly behind another MFMA waits for
        // the pipe with nothing issued meanwhile: the chores are dealt evenly over the 2 DK half-slots.
        auto kstep = [&](auto ic, auto&& chore_a, auto&& chore_b) {
            constexpr int k = decltype(ic)::value;
            constexpr int later = (DK - 1 - k) < (PF - 1) ? (DK - 1 - k) : (PF - 1);
            wait_lgkm<later>(ring[k % PF]);
            const bf16x8 a = __builtin_bit_cast(bf16x8, ring[k % PF]);
            acc[0] = mfma_32x32x16_bf16(a, pf[0][k], acc[0]);
            if constexpr (k + PF < DK) fetch(IdxC<k + PF>{});
            chore_a();
            if constexpr (NH == 2) {
                sched_fence();
                acc[1] = mfma_32x32x16_bf16(a, pf[NH - 1][k], acc[NH - 1]);
            }
            if constexpr (k < NXO) issue_piece(cq[NST - 1], rstage, k);
            if constexpr (k == NXO) issue_stat(cq[NST - 1], rstage);
            chore_b();
            sched_fence();
        };
        // NOTE: the reads that prime the ring must sit in the SAME basic block as the k-step chain: an asm load's registers
        // count as written when the asm statement ends, so a register copy at a control-flow merge would read them early.

        if (prev.valid && prev.fast) {
            // ---- pipelined: the owed tile is unmasked and unweighted (and, KIND 1, strictly right of the diagonal block) ----
            const float c2s = (prev.cmod == rmod) ? g.c_intra : g.c_inter;
            float e[NH][16], es[16], k8[8], k4[4], k2[2];
            // chore plan over the H = 2 DK half-slots:
            //   [0, H/2): scale + exp2 + row sum of the 16 NH elements
            //   [H/2, 3H/4): es = e0 + e1 (16), bf16 pack + stash stores (2 NH fragments)
            //   [3H/4, H): recursive-halving butterfly (8 + 4 + 2 + 1 + final) and the LDS slot write (KIND 1 / 3)
            constexpr int H = 2 * DK, H1 = H / 2, H2 = 3 * H / 4;
            auto chore = [&](auto hc) {
                constexpr int h = decltype(hc)::value;
                if (CROSSCLR_YABL & 32) { if (h == 0) rowacc[0] += pacc[0][0] + pacc[NH - 1][1]; return; }   // (keeps the MFMAs alive)
                if constexpr (h < H1) {
#pragma unroll
                    for (int idx = (NE * h) / H1; idx < (NE * (h + 1)) / H1; ++idx) {
                        const int s = idx >> 4, r = idx & 15;
                        const float v = fast_exp2(pacc[s][r] * c2s - g.m2);
                        e[s][r] = v;
                        rowacc[s] += v;
                    }
                } else if constexpr (h < H2) {
                    constexpr int n = H2 - H1, i = h - H1;         // n half-slots: 16 sums and 2 NH fragments
                    if (KIND != 2) {
#pragma unroll
                        for (int r = (16 * i) / n; r < (16 * (i + 1)) / n; ++r) es[r] = NH == 2 ? e[0][r] + e[NH - 1][r] : e[0][r];
                    }
                    if (ST && !(CROSSCLR_YABL & 2)) {
#pragma unroll
                        for (int f = (2 * NH * i) / n; f < (2 * NH * (i + 1)) / n; ++f) {
                            const int s = f >> 1, th = f & 1;
                            const BufRsrc rs_st = make_rsrc(stash + (st0[s] + (size_t)prev.j) * 2048, 2048u);
                            Bits8 pk;
#pragma unroll
                            for (int q = 0; q < 8; ++q) pk.v[q] = f32_to_bf16_bits(e[s][8 * th + q]);
                            buf_store16(rs_st, (unsigned)(lane * 16 + 1024 * th), 0u, __builtin_bit_cast(u32x4, pk));
                        }
                    }
                } else if constexpr (KIND != 2) {
                    constexpr int n = H - H2, i = h - H2;          // n >= 4 half-slots for the butterfly
                    constexpr int lo = (16 * i) / n, hi = (16 * (i + 1)) / n;   // work units 0..7: k8, 8..11: k4, 12..13: k2, 14: k1, 15: publish
#pragma unroll
                    for (int u = lo; u < hi; ++u) {
                        if (CROSSCLR_YABL & 4) { if (u == 15) { k2[0] = es[l31 & 15]; } else continue; }
                        if (u < 8) {
                            const bool up = (l31 >> 3) & 1;
                            k8[u] = (up ? es[8 + u] : es[u]) + lane_xor<15>(up ? es[u] : es[8 + u]);
                        } else if (u < 12) {
                            const int q = u - 8;
                            const bool up = (l31 >> 2) & 1;
                            k4[q] = (up ? k8[4 + q] : k8[q]) + lane_xor<7>(up ? k8[q] : k8[4 + q]);
                        } else if (u < 14) {
                            const int q = u - 12;
                            const bool up = (l31 >> 1) & 1;
                            k2[q] = (up ? k4[2 + q] : k4[q]) + lane_xor<2>(up ? k4[q] : k4[2 + q]);
                        } else if (u == 14) {
                            const bool up = l31 & 1;
                            const float k1 = (up ? k2[1] : k2[0]) + lane_xor<1>(up ? k2[0] : k2[1]);
                            k2[0] = k1 + lane_xor<16>(k1);
                        } else {
                            pbuf ^= 1;
                            if (l31 < 16) cs[pbuf * (4 * QT) + wave * QT + frag_row(halving_elem16(l31), half)] = k2[0];
                            pending = true;
                            ptile = colpos(prev.j);
                            prb = my_rb;
                        }
                    }
                }
            };
            static_for<PF>([&](auto ic) { fetch(ic); });
            static_for<DK>([&](auto ic) {
                constexpr int k = decltype(ic)::value;
                kstep(ic, [&]() { chore(IdxC<2 * k>{}); }, [&]() { chore(IdxC<2 * k + 1>{}); });
            });
        } else {
            if (prev.valid) epilogue_plain(pacc, prev);
            static_for<PF>([&](auto ic) { fetch(ic); });
            static_for<DK>([&](auto ic) { kstep(ic, []() {}, []() {}); });
        }
        // this tile's epilogue is owed to the next iteration
        {
            const int cmod = cq[0].in_seg >= per_mod ? 1 : 0;
            const int in_mod0 = (cq[0].in_seg - cmod * per_mod) * QT;
            const bool ragged = in_mod0 + QT > g.b;
            const bool padrows = (row0w - rmod * g.bpad) + RW > g.b;
            const bool colsum = wants_colsum(cq[0].j);
            int crank = g.row_rank;
            if (KIND != 1) { crank = g.col_rank0 + cq[0].seg; if (g.col_wrap > 0 && crank >= g.col_wrap) crank -= g.col_wrap; }
            const int r0 = row0w - rmod * g.bpad;       // the wave's first row inside its modality
            const bool selfpairs = KIND != 1 && crank == g.row_rank && cmod == rmod && (in_mod0 == r0 || (NH == 2 && in_mod0 == r0 + 32));
            prev.valid = true;
            // fast = nothing to mask: KIND 1 tiles of the diagonal block (j < TPR) hold the self pairs; padding rows only matter
            // where column sums are formed
            prev.fast = !SW && !ragged && !(colsum && padrows) && !(KIND == 1 && !colsum) && !selfpairs && !(CROSSCLR_YABL & 8);
            prev.crank = crank;
            prev.j = cq[0].j;
            prev.cmod = cmod;
            prev.in_mod0 = in_mod0;
            prev.stage = stage;
#pragma unroll
            for (int s = 0; s < NH; ++s) pacc[s] = acc[s];
        }
        if (SW) {   // weighted tiles read their k_q from the tile's ring stage: finish them before the stage is refilled
            epilogue_plain(pacc, prev);
            prev.valid = false;
        }
        stage = (stage + 1) % NST;
        ++w;
#pragma unroll
        for (int k = 0; k < NST - 1; ++k) cq[k] = cq[k + 1];
        advance(cq[NST - 1]);
    }
    timing_mark(2);
    wait_dma();      // (the clamped re-fetches past the end of the work list)
    __syncthreads();
    if (pending) { flush(); pending = false; }
    if (prev.valid) epilogue_plain(pacc, prev);
    __syncthreads();
    if (pending) flush();
    if (my_rb >= 0) store_rows();
    timing_mark(3);
}

}  // namespace crossclr
