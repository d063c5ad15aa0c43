// crossclr_kernels_symp.h -- the symmetric forward of the local block (Dpad <= 1024, whole 128-row batches, no sample weights), rebuilt
// around ONE unbroken MFMA stream per wave.
//
// Same math, same work list, same workspace / stash layout and the same summation order as fast_fwd_pipe_kernel<DK, 1, false, ST>
// (crossclr_kernels_sym.h; reference trainer/loss.py:83-100, 59-60): the two kernels produce the same bits
// (tests/test_gpu_fwd_pair.py).  What changed is everything BETWEEN the 64 MFMAs of consecutive tiles.  Round 4's kernel left the
// matrix pipe idle for ~300 instructions per tile (profiles/r05_fwd_seam.txt: 32 v_accvgpr_read copies of the finished tile, the 32 row-sum
// adds hipcc had sunk below the loop, a 5-field cursor queue, the barrier, the flush, the descriptor of the stash tile, and the
// full LDS latency of the first fragment reads): MFMA-busy 0.51.  Here:
//
//   two accumulator sets   tile i accumulates into set i & 1 (VGPRs, not AGPRs: the epilogue reads them in place -- no copies) while the
//                          epilogue of tile i - 1 works on the other set in the MFMA shadow; the loop body is unrolled by two so that the
//                          sets are static.  All 256 B-fragment registers of the wave's 64 rows live in AGPRs.
//   barrier mid-tile       barrier B_i sits between k-steps DK/2 - 1 and DK/2 of tile i and proves "tile i + 1 has landed everywhere, tile
//                          i - 1 is read by everybody": the first fragment reads of tile i + 1 are issued in the LAST k-steps of tile i
//                          and are complete when its last MFMA issues -- the seam between two tiles is the loop branch and ~12 scalar
//                          instructions.  The DMA of tile i + 3 goes out behind B_i into the stage tile i - 1 left (4-stage ring).
//   counted everything     every LDS read is asm with a counted lgkmcnt (one wait per two k-steps); all reads of a tile are issued by k-step
//                          DK - 5, so the closing lgkmcnt(0) in front of the back edge costs nothing; s_waitcnt vmcnt(NXO) in front of the
//                          barrier counts the LOADS younger than tile i + 1's pieces only (stores may retire out of order with loads).
//   scalars                the stash is ONE buffer descriptor with 32-bit scalar offsets bumped by 2 KiB per tile (the host takes this kernel
//                          for stashes below 4 GiB), the column sums leave through one descriptor as well; the work cursor is (row block,
//                          tile, tiles left in the segment) and the DMA cursor (tile, tiles left in its row block).
//   flush without branches the 32 column sums a tile publishes are added up by every wave (two ds_read2_b32) and stored by ONE of them: the
//                          other waves' store carries an out-of-range offset and is dropped by the descriptor's range check -- no divergent
//                          branch inside the MFMA stream (asm-loaded registers must not be live across control flow: tools/asm_audit.py).
//
// Tiles that need a mask (the TPR tiles of a row block's own diagonal block) run the bare MFMA stream and a plain epilogue behind it; the last
// tile of a segment (a maximal run of one row block's tiles inside the thread block's range) is finished the same way.  Everything else --
// ragged batches, padding rows, sample weights, rectangular / pair launches -- stays with fast_fwd_pipe_kernel.
#pragma once

namespace crossclr {

#ifndef CROSSCLR_ZABL
#define CROSSCLR_ZABL 0   // timing ablations (WRONG results): bit0 no overlapped epilogue chores, bit1 no stash stores, bit2 no DMA,
                          // bit3 no barrier, bit4 no MFMA, bit5 no fragment reads after the first tile, bit6 no butterfly
#endif

#ifndef CROSSCLR_ZPF
#define CROSSCLR_ZPF 4    // A fragments in flight ahead of their MFMA pair (k-steps); the first ZPF fragments of a tile are read during the previous tile
#endif
#ifndef CROSSCLR_ZW
#define CROSSCLR_ZW 2     // k-steps per counted LDS wait (2: one wait covers the fragments of k-steps k and k + 1; 1: a wait per k-step)
#endif
#ifndef CROSSCLR_ZKB
#define CROSSCLR_ZKB -1   // the k-step in front of which the tile's barrier sits (even; -1: DK / 2).  Measured (profiles/r05c_ab_fwdp.txt): the barrier
                          // at k-step 2 with the DMA pieces spread over every second k-step behind it is not faster (0.130 vs 0.127 ms without save)
#endif
#ifndef CROSSCLR_ZLAG
#define CROSSCLR_ZLAG 1   // column-sum butterfly: the DPP add of unit u - 1 behind the selects of unit u (no s_nop between a select and its DPP reader)
#endif

#ifndef CROSSCLR_EMU
// accumulators in VGPRs (the epilogue reads them with plain VALU), B fragments in AGPRs (256 of them: the whole accumulation-register half)
__device__ __forceinline__ void mfma_first_va(f32x16& acc, bf16x8 a, bf16x8 b) {
    if (CROSSCLR_ZABL & 16) { asm volatile("" : "=v"(acc) : "v"(a), "a"(b)); return; }
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_va(f32x16& acc, bf16x8 a, bf16x8 b) {
    if (CROSSCLR_ZABL & 16) { asm volatile("" : "+v"(acc) : "v"(a), "a"(b)); return; }
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
// a finished accumulator may be read by VALU 19 wait states after the MFMA that wrote it (16-pass XDL write -> VALU read); the asm MFMAs are
// invisible to hipcc's hazard recognizer, so the plain epilogues (which follow a tile's last MFMA directly) open with this
__device__ __forceinline__ void mfma_results_visible() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
__device__ __forceinline__ void barrier_only() { asm volatile("s_barrier" ::: "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm_n() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// two dwords per lane from LDS, `OFF0` and `OFF1` in units of 4 bytes from the lane's address
template <int OFF0, int OFF1> __device__ __forceinline__ u32x2 lds_read2_b32_async(unsigned addr) {
    u32x2 r;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(addr), "n"(OFF0), "n"(OFF1));
    return r;
}
#else
__device__ __forceinline__ void mfma_first_va(f32x16& acc, bf16x8 a, bf16x8 b) {
    f32x16 z;
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    acc = mfma_32x32x16_bf16(a, b, z);
}
__device__ __forceinline__ void mfma_va(f32x16& acc, bf16x8 a, bf16x8 b) { acc = mfma_32x32x16_bf16(a, b, acc); }
__device__ __forceinline__ void mfma_results_visible() {}
__device__ __forceinline__ void barrier_only() { __syncthreads(); }
template <int N> __device__ __forceinline__ void wait_lgkm_n() {}
template <int OFF0, int OFF1> __device__ __forceinline__ u32x2 lds_read2_b32_async(unsigned long addr) {
    const unsigned* p = reinterpret_cast<const unsigned*>(addr);
    u32x2 r;
    r[0] = p[OFF0];
    r[1] = p[OFF1];
    return r;
}
#endif

// ---- the fragment-read schedule of one tile (compile-time) ----
// Read q = 0 .. DK-1 of a step: q < DK - PF fetches the A fragment of k-step q + PF of THIS tile, the last PF ones the fragments of k-steps
// 0 .. PF-1 of the NEXT tile.  One read per k-step up to k-step DK - 9, two per k-step in k-steps DK - 8 .. DK - 5, none in the last four.
struct FwdReadPlan {
    static constexpr int PF = CROSSCLR_ZPF;
    static constexpr int kb(int DK) { return CROSSCLR_ZKB < 0 ? DK / 2 : CROSSCLR_ZKB; }
    static constexpr int kstep_of(int DK, int q) { return q < DK - 8 ? q : (DK - 8) + (q - (DK - 8)) / 2; }
    static constexpr int reads_before(int DK, int k) {        // reads issued in k-steps < k
        return k <= DK - 8 ? k : ((DK - 8) + 2 * (k - (DK - 8)) < DK ? (DK - 8) + 2 * (k - (DK - 8)) : DK);
    }
    // LDS operations issued before the wait at the head of k-step k (the flush's two reads go out at the head of k-step KB, behind its wait)
    static constexpr int ops_before(int DK, int k) { return reads_before(DK, k) + (k > kb(DK) ? 2 : 0); }
    static constexpr int seq_of(int DK, int q) { return q + (kstep_of(DK, q) >= kb(DK) ? 2 : 0); }
    // younger operations that may stay in flight when fragment k' (k' >= PF) of this tile is needed at the head of k-step k <= k'
    static constexpr int keep_for(int DK, int k, int kfrag) { return ops_before(DK, k) - seq_of(DK, kfrag - PF) - 1; }
    // the flush's second read is consumed in k-step KB + 2, behind that k-step's wait
    static constexpr int flush_seq(int DK) { return reads_before(DK, kb(DK)) + 1; }
};

// DK = k-steps of 16 embedding columns per ring STAGE, KS = stages per column tile (Dpad = 16 DK KS), NH = 32-row halves per wave:
//   Dpad <= 512:        <DK = Dpad / 16, ST, 2, 1>   256-row blocks, one MFMA pair per LDS read, a tile = one stage
//   512 < Dpad <= 1024: <DK = Dpad / 32, ST, 1, 2>   128-row blocks (the fragments of 64 rows x 1024 columns would not fit the register file); a
//                       64-KiB tile would leave room for two ring stages only, so the tile travels as TWO stages of half its embedding columns:
//                       the ring, the barrier, the DMA and the fragment reads run per stage exactly as below, the accumulators run through
//                       both stages and the epilogue of the previous tile is dealt over the 2 DK MFMA slots of both.
// KIND (the work lists of crossclr_device.h / crossclr_kernels_sym.h):
//   1  symmetric local block (rows = columns = x): upper triangle, the row block's own TPR diagonal tiles masked, column sums for the rest
//   2  rectangular: this rank's rows against OTHER ranks' columns (xc = their segments; the host keeps launches that would contain the
//      rows' own rank with fast_fwd_pipe_kernel) -- no masks at all, no column sums; ST: the rectangular stash of crossclr_forward_rect_save
//   3  pairs: rectangular, and every tile also yields its column sums over this rank's rows (what the partner rank is owed)
// SW (per-sample weights): k >= 0 multiplies a sample's exponential wherever it is an INTRA-modal negative column -- the row sums
//   take e k_q, the column sums (the mirrored tile's row sums) e k_p.  The tile's 32 k_q are fetched by four asm buffer loads per lane behind the
//   tile's barrier (in front of the step's DMA pieces, waited for at the end of the step: s_waitcnt vmcnt(NXO)) and travel to the tile's epilogue
//   in registers; inter-modal tiles multiply by 1.0 (exact), so both kinds of tile run the same branch-free chores.  Instantiated for NH = 1 (wide
//   operands: BASELINE config 5's D = 1024) -- beside the accumulators of both 32-row halves the 16 scales spill; those launches keep fast_fwd_pipe_kernel.
template <int DK, bool ST, int NH = 2, int KS = 1, int KIND = 1, bool SW = false>
__global__ void __launch_bounds__(256, 1) fast_fwd_pair_kernel(const bf16_t* x, const bf16_t* xc, Geo g, FwdWork wk, float* part, float* colpart,
                                                               int* header, unsigned char* stash, unsigned stash_bytes, FwdPerm perm,
                                                               const float* ks, const float* kc) {
    static_assert(KIND >= 1 && KIND <= 3, "1 symmetric, 2 rectangular, 3 pairs");
    constexpr bool CSUM = KIND != 2;       // column sums are formed (KIND 1: right of the diagonal block)
    constexpr int RB = DK * 32;            // bytes per row of a ring stage
    constexpr int RBG = KS * RB;           // bytes per operand row in memory
    constexpr int QT = 32;
    constexpr int TILE = QT * RB;          // bytes of a ring stage
    constexpr int TILEG = QT * RBG;        // bytes of a column tile in memory
    constexpr int TPR = 4 * NH;            // 32-row groups per row block
    constexpr int RW = 32 * NH;            // rows per wave
    constexpr int RBLK = 4 * RW;
    constexpr int NST = 4;
    constexpr int NXO = DK / 4;            // DMA pieces per wave and stage
    constexpr int PF = FwdReadPlan::PF;
    constexpr int SPS = DK * NH;           // MFMA slots of a step (= a stage)
    constexpr int H = SPS * KS, H1 = H / 2, H2 = 3 * H / 4;     // MFMA slots of a tile: exp + row sums | sums of halves, pack, stash | butterfly
    static_assert((NH == 2 && KS == 1) || (NH == 1 && KS == 2), "256-row blocks with whole tiles, or 128-row blocks with the tile in two stages");
    constexpr int KB = FwdReadPlan::kb(DK);      // the k-step the barrier sits in front of
    // DMA pieces behind the barrier: every second k-step from KB + 1 on where that fits, else one per k-step from KB on
    constexpr int DSTRIDE = (KB + 1 + 2 * (NXO - 1) < DK) ? 2 : 1, DK0 = DSTRIDE == 2 ? KB + 1 : KB;
    // SW: the k-step at whose head the tile's column scales are waited for -- behind the last DMA piece, as late as the blend's VALU still hides
    // (DK: behind the k-step chain, where the short tiles of Dpad = 128 issue their last piece in the last k-step)
    constexpr int KLAST = DK0 + DSTRIDE * (NXO - 1), KWAIT = KLAST + 1 > DK - 3 ? KLAST + 1 : DK - 3;
    static_assert(KB % 2 == 0 && KB + 2 < DK && DK0 + DSTRIDE * (NXO - 1) < DK && PF % 2 == 0 && PF >= 4 && PF <= DK - 2, "schedule constants");
    constexpr int CS0 = NST * TILE;        // two column-sum slots [4 waves][32] floats, then a dump slot for the lanes that publish nothing
    static_assert(DK % 8 == 0 && DK >= 8 && DK <= 32, "Dpad in {128, 256, 384, 512} (KS = 1) / {768, 1024} (KS = 2)");
    static_assert(NST * TILE + 2 * 4 * QT * 4 + 256 <= 160 * 1024, "LDS budget");
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[CS0 + 2 * 4 * QT * 4 + 256];

    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    timing_mark(0);
    if (blockIdx.x == 0 && tid == 0) { header[0] = wk.kind; header[1] = TPR; header[2] = wk.NT; header[3] = wk.per; }
    const int NT = wk.NT;
    const int per_mod = g.bpad / QT, per_rank = 2 * per_mod;
    // rectangular launches: usable segment u of the column operand -> the rank segment it lives in (crossclr_kernels_sym.h: skipped rank,
    // col_wrap = W: the operand is the whole gathered array and segment i is rank (col_rank0 + i) mod W)
    const int skip_seg = (KIND == 2 && g.skip_rank >= g.col_rank0 && g.skip_rank < g.col_rank0 + g.col_ranks) ? g.skip_rank - g.col_rank0 : -1;
    const int col_segs = KIND == 1 ? 1 : (g.col_wrap > 0 ? g.col_wrap : g.col_ranks);
    auto seg_mem = [&](int u) {            // memory segment of usable segment u
        int r = u + ((skip_seg >= 0 && u >= skip_seg) ? 1 : 0);
        if (g.col_wrap > 0) { r += g.col_rank0; if (r >= g.col_wrap) r -= g.col_wrap; }
        return r;
    };

    const int vb = uniform(blockIdx.x < 256 ? (int)perm.v[blockIdx.x < 256 ? blockIdx.x : 0] : (int)blockIdx.x);
    int w = fwd_block_begin(wk, vb);
    const int w_end = fwd_block_begin(wk, vb + 1);
    if (w >= w_end) return;

    // ---- descriptors and per-lane offsets ----
    const BufRsrc rs_x = make_rsrc(KIND == 1 ? x : xc, (unsigned)((size_t)col_segs * 2 * g.bpad * RBG));
    const BufRsrc rs_st = make_rsrc(stash, stash_bytes);
    const BufRsrc rs_cp = make_rsrc(colpart, (unsigned)((size_t)wk.NB * NT * QT * 4));
    const RawRsrc rs_k = make_raw_rsrc(SW ? (const void*)kc : (const void*)x, (unsigned)((size_t)col_segs * 2 * g.bpad * 4));
    const unsigned kq_voff = (unsigned)(16 * half);      // register quad r4 of a lane = columns 8 r4 + 4 half .. + 3 of the tile
    unsigned voffx[NXO];
#pragma unroll
    for (int k = 0; k < NXO; ++k) {
        const int L = (wave + 4 * k) * 1024 + lane * 16;
        const int row = L / RB, slot = (L - row * RB) >> 4;
        voffx[k] = (unsigned)(row * RBG + (swz_slot(slot, row) << 4));
    }
    // LDS address of logical chunk (2j + half) of this lane's tile row IN THE CURRENT STAGE: carried from step to step (bumped by the ring
    // distance: 8 VALU per step, the same as rebuilding them from per-lane offsets, and 8 registers fewer)
    decltype(lds_addr(lds)) abase[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) abase[j] = lds_addr(lds) + (unsigned)(l31 * RB + ((((2 * j + half) ^ sigma16(l31)) & 15) << 4));
    float* cs = reinterpret_cast<float*>(lds + CS0);
    // publish: lanes l31 < 16 write their column sum to cs[buf][wave][frag_row(elem, half)], the others into the dump slot
    const unsigned pub_addr = (unsigned)(l31 < 16 ? (wave * QT + frag_row(halving_elem16(l31), half)) * 4 : 2 * 4 * QT * 4 + lane * 4);
    const unsigned pub_flip = l31 < 16 ? (unsigned)(4 * QT * 4) : 0u;      // the second buffer, for publishing lanes only
    const unsigned st_voff = (unsigned)(lane * 16);

    // ---- DMA cursor: item w + 3 ahead of the compute cursor after the prologue ----
    // KIND 1: d_mt = the column tile itself (row block d_rb owns tiles TPR d_rb .. NT - 1).  Rectangular: d_u = usable segment, d_mt = tile
    // inside the segment (every row block owns all NT usable tiles); d_left = tiles left in the DMA cursor's row block.
    int d_rb, d_mt, d_left, d_u = 0, d_kh = 0;    // (d_kh: which stage of the tile goes out next, KS = 2)
    int rb, j;                                    // compute cursor: row block, tile inside its list
    {
        int r0 = 0;
        if (KIND == 1) { while (fwd_prefix(wk, r0 + 1) <= w) ++r0; } else r0 = w / NT;
        const int j0 = w - fwd_prefix(wk, r0);
        d_rb = rb = r0;
        j = j0;
        if (KIND == 1) { d_mt = TPR * r0 + j0; d_left = NT - d_mt; }
        else { d_u = j0 / per_rank; d_mt = j0 - d_u * per_rank; d_left = NT - j0; }
    }
    unsigned dstage = 0;                          // LDS byte offset of the stage the next DMA fills
    auto dma_tile = [&]() {
        const int mt = KIND == 1 ? (d_mt < NT - 1 ? d_mt : NT - 1) : seg_mem(d_u) * per_rank + d_mt;
        return (unsigned)mt * (unsigned)TILEG + (unsigned)(d_kh * RB);
    };
    auto ring_next = [&](unsigned o) {            // next stage of the ring (a power of two of bytes except at DK = 24)
        if constexpr ((NST * TILE & (NST * TILE - 1)) == 0) return (o + (unsigned)TILE) & (unsigned)(NST * TILE - 1);
        else { const unsigned nx = o + (unsigned)TILE; return nx == (unsigned)(NST * TILE) ? 0u : nx; }
    };
    auto dma_advance = [&]() {
        dstage = ring_next(dstage);
        if (KS == 2) { d_kh ^= 1; if (d_kh) return; }      // (the tile's second stage follows)
        ++d_mt;
        if (KIND != 1 && d_mt == per_rank) { d_mt = 0; ++d_u; }
        if (--d_left == 0) {
            ++d_rb;
            if (KIND == 1) {
                d_mt = TPR * d_rb;
                d_left = NT - d_mt;
                if (d_left <= 0) d_left = 1 << 30;   // past the last row block: clamped re-fetches of the last tile, never consumed
            } else {
                d_u = 0; d_mt = 0; d_left = NT;      // (past the last row block: re-fetches of valid tiles, never consumed)
            }
        }
    };
    auto issue_piece = [&](int k, unsigned tile_off, unsigned stage_off) {
        if (CROSSCLR_ZABL & 4) return;
        lds_dma16_buf(rs_x, voffx[k], tile_off, lds + stage_off + (wave + 4 * k) * 1024);
    };
    // prologue: the first three stages of the range (KS = 1: the tiles of items w, w + 1, w + 2)
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) {
        const unsigned to = dma_tile();
#pragma unroll
        for (int k = 0; k < NXO; ++k) issue_piece(k, to, dstage);
        dma_advance();
    }

    // ---- per-segment state ----
    float rowacc[NH], kp[NH];
    u32x4 kq[4], kq_next[4];                      // SW: the 16 column scales of this lane for the tile behind the cursor (set by that tile's
                                                  // last step; KS = 2: loaded in its first step, parked in kq_next while the owed tile still needs kq)
    bf16x8 pf[NH][DK * KS];
    int row0w = 0, rmod = 0;
    unsigned st_soff[NH];                         // stash byte offset of the CURRENT tile's records, per 32-row half
    // ---- published column sums waiting for their flush ----
    int pend = 0;                                 // 1: cs[pbuf] holds the sums of colpart row pend_soff
    unsigned pbuf = 0, pend_soff = 0;
    unsigned cstage = 0;                          // LDS byte offset of the current tile's stage
    // first PF fragments of the CURRENT tile (read during the previous step / the prologue)
    u32x4 nx[PF];

    struct Bits8 { bf16_t v[8]; };
    auto store_rows = [&]() {
#pragma unroll
        for (int s = 0; s < NH; ++s) {
            const float v = rowacc[s] + wave_xor_f32(rowacc[s], 32);
            if (half == 0) part[(size_t)(vb - fwd_first_block(wk, rb)) * 2 * g.bpad + row0w + 32 * s + l31] = v;
        }
    };
    // the flush of a pending publication OUTSIDE a step (segment end, kernel end): every wave adds, wave 0 stores
    auto flush_now = [&]() {
        if (pend && tid < QT) {
            const float* c = cs + pbuf * (4 * QT);
            buf_store4(rs_cp, (unsigned)(tid * 4), pend_soff, (c[tid] + c[QT + tid]) + (c[2 * QT + tid] + c[3 * QT + tid]));
        }
        pend = 0;
    };
    auto publish = [&](float colsum, unsigned soff) {
        pbuf ^= 1u;
        *reinterpret_cast<float*>(lds + CS0 + pub_addr + (pbuf ? pub_flip : 0u)) = colsum;
        pend = 1;
        pend_soff = soff;
    };
    auto stash_store = [&](const f32x16& e, int s, unsigned soff) {
        if (!ST || (CROSSCLR_ZABL & 2)) return;
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            Bits8 pk;
#pragma unroll
            for (int q = 0; q < 8; ++q) pk.v[q] = f32_to_bf16_bits(e[8 * th + q]);
            buf_store16(rs_st, st_voff + 1024u * th, soff, __builtin_bit_cast(u32x4, pk));
        }
    };
    // The tile BEHIND the compute cursor (index j - 1 of the row block's list -- the one whose epilogue is owed): modality of its columns and
    // the byte offset of its 32 column sums in `colpart`.  c_in = position of tile j inside its rank segment (rectangular launches).
    int c_in = 0;
    struct TileInfo { int cmod; unsigned cs_off; };
    auto owed_tile = [&]() {
        TileInfo t;
        if (KIND == 1) {
            const int mt = TPR * rb + j - 1;
            t.cmod = mt >= per_mod ? 1 : 0;
            t.cs_off = (unsigned)((rb * NT + mt) * (QT * 4));
        } else {
            const int in = c_in == 0 ? per_rank - 1 : c_in - 1;
            t.cmod = in >= per_mod ? 1 : 0;
            t.cs_off = (unsigned)((rb * NT + j - 1) * (QT * 4));
        }
        return t;
    };
    int c_u = 0;                                  // rectangular launches: usable segment of tile j
    auto cur_tile_mt = [&]() { return KIND == 1 ? TPR * rb + j : seg_mem(c_u) * per_rank + c_in; };     // memory tile (statistics index / 32) of tile j
    // plain (not overlapped) epilogue of the tile in `acc`: jt = j - 1 = its index in the row block's list
    auto epilogue_plain = [&](f32x16 (&acc)[NH], int jt) __attribute__((always_inline)) {
        mfma_results_visible();
        const TileInfo ti = owed_tile();
        const float c2s = (ti.cmod == rmod) ? g.c_intra : g.c_inter;
        const bool upper = KIND == 1 ? jt >= TPR : KIND == 3;
        const float ninf = -__builtin_inff();
        float es[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) es[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NH; ++s) {
            float xx[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xx[r] = acc[s][r] * c2s - g.m2;
            if (KIND == 1 && jt == NH * wave + s) {             // the tile that holds this half's self pairs
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (frag_row(r, half) == l31) xx[r] = ninf;
            }
            f32x16 e;
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = fast_exp2(xx[r]);
            stash_store(e, s, st_soff[s] + 2048u * (unsigned)jt);
            if (SW) {      // (kq holds 1.0 for an inter-modal tile: exact)
                const float kpe = ti.cmod == rmod ? kp[s] : 1.f;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 k4 = __builtin_bit_cast(f32x4, kq[r4]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { rowacc[s] += e[4 * r4 + q] * k4[q]; es[4 * r4 + q] += e[4 * r4 + q] * kpe; }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) { rowacc[s] += e[r]; es[r] += e[r]; }
            }
        }
        if (upper) publish(halving_sum16(es, l31), ti.cs_off);
    };

    // ---- one stage: NH DK MFMAs into accC (stage KH of tile jt); MODE 1: the epilogue of the previous tile (accP, index jt - 1) in their shadow ----
    auto step = [&](auto modec, auto khc, f32x16 (&accC)[NH], f32x16 (&accP)[NH], int jt) __attribute__((always_inline)) {
        constexpr int MODE = decltype(modec)::value, KH = decltype(khc)::value;
        constexpr bool EPI = MODE == 1 && !(CROSSCLR_ZABL & 1);
        const unsigned nstage = ring_next(cstage);
        const int ring_step = (int)nstage - (int)cstage;      // + TILE, or back to the first stage
        decltype(lds_addr(lds)) anext[PF];
#pragma unroll
        for (int q = 0; q < PF; ++q) anext[q] = abase[q] + ring_step;
        u32x4 fr[DK];                                  // A fragments of k-steps PF .. DK-1 (indices < PF unused)
        u32x4 nn[PF];                                  // first fragments of the next tile
#pragma unroll
        for (int q = 0; q < PF; ++q) fr[q] = nx[q];
        // the owed tile (MODE 1)
        const TileInfo ti = owed_tile();             // (jt == j here: the tile behind the cursor is the owed one)
        const float c2s = (ti.cmod == rmod) ? g.c_intra : g.c_inter;
        const unsigned so0 = st_soff[0] + 2048u * (unsigned)(jt - 1), so1 = st_soff[NH - 1] + 2048u * (unsigned)(jt - 1);
        float es[16], k8[8], k4[4], k2[2], sa[15], sb[15];
        // SW: the owed tile's scales -- its columns' (set by its own step) and its rows' -- or 1.0 where they do not apply (inter-modal tile)
        const bool weigh = SW && ti.cmod == rmod;
        float kpe[NH];
        const bool weigh_cur = SW && ((KIND == 1 ? (TPR * rb + jt >= per_mod ? 1 : 0) : (c_in >= per_mod ? 1 : 0)) == rmod);      // this tile: intra-modal?
        u32x4 kqn[4];                                  // SW: this tile's column scales, loaded behind the barrier
        const unsigned kq_soff = (unsigned)cur_tile_mt() * (unsigned)(QT * 4);
        if (SW) {
#pragma unroll
            for (int s = 0; s < NH; ++s) kpe[s] = weigh ? kp[s] : 1.f;
        }
        // the pending publication (flushed behind this step's barrier)
        const auto fa = lds_addr(lds + CS0 + (pbuf ? 4 * QT * 4 : 0) + l31 * 4);
        const unsigned f_voff = (pend && wave == (jt & 3) && half == 0) ? (unsigned)(l31 * 4) : 0xFFFFFF00u;
        const unsigned f_soff = pend_soff;
        u32x2 f01, f23;
        const unsigned d_to = dma_tile(), d_so = dstage;

        auto issue_read = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if ((CROSSCLR_ZABL & 32)) {
                if constexpr (q < DK - PF) fr[q + PF] = fr[(q + PF) % PF]; else nn[q - (DK - PF)] = fr[q - (DK - PF)];
                return;
            }
            if constexpr (q < DK - PF) {
                constexpr int kf = q + PF;
                fr[kf] = lds_read_b128_async<(kf >> 3) * 256>(abase[kf & 7]);
            } else {
                constexpr int kn = q - (DK - PF);
                nn[kn] = lds_read_b128_async<0>(anext[kn]);
            }
        };
        auto chore = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            if constexpr (!EPI) return;
            if constexpr (h < H1) {
                // exp of element idx, then the row-sum add of element idx - 1 (a transcendental's result costs a wait state when it is used
                // by the very next instruction; the adds keep their order r = 0 .. 15 per half: the same bits as the plain epilogue)
#pragma unroll
                for (int idx = (16 * NH * h) / H1; idx < (16 * NH * (h + 1)) / H1; ++idx) {
                    const int s = idx >> 4, r = idx & 15;
                    accP[s][r] = fast_exp2(accP[s][r] * c2s - g.m2);
                    if (idx > 0) {
                        const int sp = (idx - 1) >> 4, rp = (idx - 1) & 15;
                        if (SW) rowacc[sp] += accP[sp][rp] * __builtin_bit_cast(f32x4, kq[rp >> 2])[rp & 3];
                        else rowacc[sp] += accP[sp][rp];
                        pin_v(rowacc[sp]);
                    }
                }
            } else if constexpr (h < H2) {
                constexpr int n = H2 - H1, i = h - H1;
                if constexpr (h == H1) {      // (KS = 2: a new invocation of this lambda: the scale is read again)
                    if (SW) rowacc[NH - 1] += accP[NH - 1][15] * __builtin_bit_cast(f32x4, kq[3])[3];
                    else rowacc[NH - 1] += accP[NH - 1][15];
                    pin_v(rowacc[NH - 1]);
                }
                if (CSUM) {
#pragma unroll
                    for (int r = (16 * i) / n; r < (16 * (i + 1)) / n; ++r) {
                        if (SW) { es[r] = accP[0][r] * kpe[0]; if (NH == 2) es[r] += accP[NH - 1][r] * kpe[NH - 1]; }
                        else es[r] = NH == 2 ? accP[0][r] + accP[NH - 1][r] : accP[0][r];
                    }
                }
                if (ST && !(CROSSCLR_ZABL & 2)) {
#pragma unroll
                    for (int f = (2 * NH * i) / n; f < (2 * NH * (i + 1)) / n; ++f) {
                        const int s = f >> 1, th = f & 1;
                        Bits8 pk;
#pragma unroll
                        for (int q = 0; q < 8; ++q) pk.v[q] = f32_to_bf16_bits(accP[s][8 * th + q]);
                        buf_store16(rs_st, st_voff + 1024u * th, s ? so1 : so0, __builtin_bit_cast(u32x4, pk));
                    }
                }
            } else if constexpr (!CSUM) {
                // (rectangular launch without column sums: the last quarter of the tile carries no chores)
            } else if constexpr (CROSSCLR_ZLAG == 0) {
                constexpr int n = H - H2, i = h - H2;
                constexpr int lo = (16 * i) / n, hi = (16 * (i + 1)) / n;     // units 0..7: k8, 8..11: k4, 12..13: k2, 14: k1, 15: (publish: behind the loop)
#pragma unroll
                for (int u = lo; u < hi; ++u) {
                    if (CROSSCLR_ZABL & 64) { if (u == 14) k2[0] = es[l31 & 15]; continue; }
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
                    }
                }
            } else {
                // the same 15 exchanges (halving_sum16), each cut in two: unit u SELECTS its two operands, the DPP add that consumes them
                // runs with unit u + 1 -- one MFMA slot later, so no wait states between a v_cndmask and the DPP instruction reading it
                constexpr int n = H - H2, i = h - H2;
                constexpr int lo = (16 * i) / n, hi = (16 * (i + 1)) / n;
#pragma unroll
                for (int u = lo; u < hi; ++u) {
                    if (CROSSCLR_ZABL & 64) { if (u == 15) k2[0] = es[l31 & 15]; continue; }
                    // combine(u - 1)
                    if (u >= 1 && u <= 8) k8[u - 1] = sa[u - 1] + lane_xor<15>(sb[u - 1]);
                    else if (u >= 9 && u <= 12) k4[u - 9] = sa[u - 1] + lane_xor<7>(sb[u - 1]);
                    else if (u >= 13 && u <= 14) k2[u - 13] = sa[u - 1] + lane_xor<2>(sb[u - 1]);
                    else if (u == 15) { const float k1 = sa[14] + lane_xor<1>(sb[14]); k2[0] = k1 + lane_xor<16>(k1); }
                    // select(u)
                    if (u < 8) {
                        const bool up = (l31 >> 3) & 1;
                        sa[u] = up ? es[8 + u] : es[u]; sb[u] = up ? es[u] : es[8 + u];
                    } else if (u < 12) {
                        const int q = u - 8;
                        const bool up = (l31 >> 2) & 1;
                        sa[u] = up ? k8[4 + q] : k8[q]; sb[u] = up ? k8[q] : k8[4 + q];
                    } else if (u < 14) {
                        const int q = u - 12;
                        const bool up = (l31 >> 1) & 1;
                        sa[u] = up ? k4[2 + q] : k4[q]; sb[u] = up ? k4[q] : k4[2 + q];
                    } else if (u == 14) {
                        const bool up = l31 & 1;
                        sa[u] = up ? k2[1] : k2[0]; sb[u] = up ? k2[0] : k2[1];
                    }
                    if (u < 15) { pin_v(sa[u]); pin_v(sb[u]); }
                }
            }
        };

        // this tile's column scales have landed (they went out at the barrier, in front of the DMA pieces: only those may still be in flight); an
        // inter-modal tile's become 1.0 -- its epilogue then multiplies exactly like an intra-modal one's
        auto scales_landed = [&]() {
            wait_dma_keep<NXO>();
#pragma unroll
            for (int q = 0; q < 4; ++q) after_wait(kqn[q]);
            const u32x4 ones = {0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};
#pragma unroll
            for (int q = 0; q < 4; ++q) kqn[q] = weigh_cur ? kqn[q] : ones;
        };
        static_for<DK>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            // ---- head: the fragments of k-steps k and k + 1 are complete (one counted wait per two k-steps); at k-step KB + 2 the flush's reads too
            if constexpr ((k >= PF && (k % CROSSCLR_ZW == 0)) || k == KB + 2) {
                constexpr bool frag = k >= PF && (k % CROSSCLR_ZW == 0);
                constexpr int kl = (CROSSCLR_ZW == 2 && k + 1 < DK) ? k + 1 : k;
                constexpr int keep_frag = frag ? FwdReadPlan::keep_for(DK, k, kl) : 63;
                constexpr int keep_flush = k == KB + 2 ? FwdReadPlan::ops_before(DK, k) - FwdReadPlan::flush_seq(DK) - 1 : 63;
                constexpr int keep = keep_flush < keep_frag ? keep_flush : keep_frag;
                static_assert(keep >= 0 && keep < 16, "a wait cannot ask for an operation that has not been issued (lgkmcnt is 4 bits wide)");
                wait_lgkm_n<keep>();
                if constexpr (frag) {
                    after_wait(fr[k]);
                    if constexpr (CROSSCLR_ZW == 2 && k + 1 < DK) after_wait(fr[k + 1]);
                }
                if constexpr (k == KB + 2) { after_wait(f01); after_wait(f23); }
            }
            if constexpr (k == KB) {
                // tile i + 1 has landed (only the NXO pieces of tile i + 2 may still be in flight) -- everywhere; everybody is done with tile i - 1
                if (!(CROSSCLR_ZABL & 4)) wait_dma_keep<NXO>();
                if (!(CROSSCLR_ZABL & 8)) barrier_only();
                f01 = lds_read2_b32_async<0, QT>(fa);
                f23 = lds_read2_b32_async<2 * QT, 3 * QT>(fa);
                if constexpr (SW && KH == 0) {       // this tile's column scales: IN FRONT of the step's DMA pieces (the closing vmcnt(NXO) covers them)
                    kqn[0] = buf_load_b128_async<0>(rs_k, kq_voff, kq_soff);
                    kqn[1] = buf_load_b128_async<32>(rs_k, kq_voff, kq_soff);
                    kqn[2] = buf_load_b128_async<64>(rs_k, kq_voff, kq_soff);
                    kqn[3] = buf_load_b128_async<96>(rs_k, kq_voff, kq_soff);
                }
            }
            const bf16x8 a = __builtin_bit_cast(bf16x8, fr[k]);
            constexpr int kp = KH * DK + k;          // the k-step inside the row fragments
            if constexpr (kp == 0) mfma_first_va(accC[0], a, pf[0][kp]); else mfma_va(accC[0], a, pf[0][kp]);
            // ---- this k-step's fragment reads
            static_for<DK>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (FwdReadPlan::kstep_of(DK, q) == k) issue_read(qc);
            });
            if constexpr (k == KB + 2) {
                // (whole-vector casts: clang mis-reads __builtin_bit_cast(float, v[1]) of an asm-written ext_vector element as v[0])
                const f32x2 c01 = __builtin_bit_cast(f32x2, f01), c23 = __builtin_bit_cast(f32x2, f23);
                buf_store4(rs_cp, f_voff, f_soff, (c01[0] + c01[1]) + (c23[0] + c23[1]));
            }
            if constexpr (SW && KH == 0 && k == KWAIT) scales_landed();
            chore(IdxC<KH * SPS + NH * k>{});
            sched_fence();
            if constexpr (NH == 2) { if constexpr (kp == 0) mfma_first_va(accC[NH - 1], a, pf[NH - 1][kp]); else mfma_va(accC[NH - 1], a, pf[NH - 1][kp]); }
            if constexpr (k >= DK0 && (k - DK0) % DSTRIDE == 0 && (k - DK0) / DSTRIDE < NXO) issue_piece((k - DK0) / DSTRIDE, d_to, d_so);
            if constexpr (NH == 2) chore(IdxC<KH * SPS + NH * k + 1>{});
            sched_fence();
        });
        // every read of the step is complete (all were issued >= 4 k-steps ago): the next tile's first fragments may cross the back edge
        wait_lgkm_n<0>();
#pragma unroll
        for (int q = 0; q < PF; ++q) after_wait(nn[q]);
#pragma unroll
        for (int q = 0; q < PF; ++q) nx[q] = nn[q];
        if constexpr (SW && KH == 0 && KWAIT >= DK) scales_landed();
        if constexpr (SW && KH == KS - 1) {          // from here on the tile is "the tile behind the cursor"
#pragma unroll
            for (int q = 0; q < 4; ++q) kq[q] = KS == 1 ? kqn[q] : kq_next[q];
        }
        if constexpr (SW && KS == 2 && KH == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) kq_next[q] = kqn[q];
        }
        pend = 0;
        if constexpr (EPI && CSUM && KH == KS - 1) {
            if (!(CROSSCLR_ZABL & 64)) publish(k2[0], ti.cs_off);
        } else if constexpr (!EPI && MODE == 1 && KH == KS - 1) {
            rowacc[0] += accP[0][0] + accP[NH - 1][1];      // (ablation: keeps the previous tile's MFMAs alive)
        }
        dma_advance();
        cstage = nstage;
#pragma unroll
        for (int q = 0; q < 8; ++q) abase[q] += ring_step;
    };

    // a tile = KS stages
    auto tile = [&](auto modec, f32x16 (&accC)[NH], f32x16 (&accP)[NH], int jt) __attribute__((always_inline)) {
        step(modec, IdxC<0>{}, accC, accP, jt);
        if constexpr (KS == 2) step(modec, IdxC<1>{}, accC, accP, jt);
    };
    f32x16 accA[NH], accB[NH];
    timing_mark(1);
    bool primed = false;
    while (w < w_end) {
        // ---- a segment: the tiles j .. j + n - 1 of row block rb ----
        int n = (KIND == 1 ? NT - TPR * rb : NT) - j;
        if (n > w_end - w) n = w_end - w;
        w += n;
        row0w = rb * RBLK + RW * wave;
        rmod = uniform(row0w / g.bpad);
        c_u = KIND == 1 ? 0 : j / per_rank;        // (j > 0 only in a range's first segment: one division per thread block)
        c_in = KIND == 1 ? 0 : j - c_u * per_rank;
#pragma unroll
        for (int s = 0; s < NH; ++s) {
            rowacc[s] = 0.f;
            kp[s] = SW ? ks[row0w + 32 * s + l31] : 1.f;
            st_soff[s] = !ST ? 0u : (KIND == 1 ? (unsigned)(stash_tile_index(TPR, NT, TPR * rb + NH * wave + s, TPR * rb) * 2048)
                                               : (unsigned)(TPR * rb + NH * wave + s) * (unsigned)NT * 2048u);
            const bf16_t* src = x + (size_t)(row0w + 32 * s + l31) * (DK * KS * 16) + 8 * half;      // (the ROW operand)
#pragma unroll
            for (int k = 0; k < DK * KS; ++k) pf[s][k] = *reinterpret_cast<const bf16x8*>(src + 16 * k);
        }
        wait_loads_visible();
        if (!primed) {
            // the first tile of the range has landed everywhere; its first fragments
            primed = true;
            wait_dma();
            barrier_only();
            static_for<PF>([&](auto qc) { constexpr int q = decltype(qc)::value; nx[q] = lds_read_b128_async<(q >> 3) * 256>(abase[q & 7]); });
            wait_lgkm_n<0>();
#pragma unroll
            for (int q = 0; q < PF; ++q) after_wait(nx[q]);
        }
        // phase 1: masked tiles one by one, up to and including the first tile that can stay owed (set A)
        bool owedA = false;
        auto next_tile = [&]() { ++j; --n; if (KIND != 1 && ++c_in == per_rank) { c_in = 0; ++c_u; } };
        while (n > 0) {
            tile(IdxC<0>{}, accA, accB, j);
            next_tile();
            if (KIND == 1 && j - 1 < TPR) epilogue_plain(accA, j - 1);
            else { owedA = true; break; }
        }
        // phase 2: pairs -- (B while A's epilogue runs), (A while B's epilogue runs)
        bool owedB = false;
        while (n > 0) {
            tile(IdxC<1>{}, accB, accA, j);
            next_tile();
            if (n == 0) { owedA = false; owedB = true; break; }
            tile(IdxC<1>{}, accA, accB, j);
            next_tile();
        }
        // the segment's last tile: an earlier publication may still wait for its barrier -- flush it, then finish the tile in the open
        if (owedA || owedB) {
            wait_lgkm_n<0>();
            barrier_only();
            flush_now();
            if (owedA) epilogue_plain(accA, j - 1); else epilogue_plain(accB, j - 1);
        }
        store_rows();
        ++rb;
        j = 0;
    }
    timing_mark(2);
    wait_dma();      // (the clamped re-fetches past the end of the work list)
    __syncthreads();
    flush_now();
    timing_mark(3);
}

}  // namespace crossclr
