// crossclr_kernels_project.h -- the step BEFORE the loss (reference README.md:24-38: the criterion is fed
// "features: [bsz, f_dim]" that a projection head produced): linear projection + bias + row L2-normalisation (trainer/loss.py:79-80)
// + packing into the kernels' operand, in ONE launch -- the separate normalize pass (crossclr_normalize: reads 2 B D fp32, writes
// the packed operand) and the HBM round trip of the projected features disappear.
//
//   project_pack_kernel   Y_m = X_m W_m^T + b_m for both modalities m of the same 64 rows (bf16 MFMA, fp32 accumulate: 64 rows x Dpad
//                         columns x 2 modalities = 256 accumulator registers per wave), then -- the rows are complete inside the
//                         block -- ||y||, y / max(||y||, 1e-12), the positive-pair cosine v^_i . t^_i in fp32 (it must NOT come from the
//                         bf16-rounded operand: its error is multiplied by 1/tau), bf16 pack through LDS, 16-byte coalesced stores.
//                         HBM-bound (reads 2 b Din inputs once, writes the packed operand once); the weights stay in L2.
//   project_backward_prep_kernel   the normalise-backward in front of the projection's own backward: from the gradient w.r.t. the unit
//                         rows (crossclr_backward_finish_p, prenormalized = 1) g_y = (G - y^ (y^ . G)) / ||y||; the two GEMMs that follow
//                         (dW = g_y^T X, dX = g_y W) are plain library GEMMs on the caller's side.
#pragma once

namespace crossclr {

// 16 consecutive elements of an input row, fetched as RAW registers (16-byte loads where the row allows it) and converted to bf16 only when
// they are committed to LDS -- so that the loads of K chunk c + 1 stay in flight behind the MFMAs of chunk c (a converting load would be
// waited for on the spot: the first version of this kernel exposed one HBM latency per chunk and read bf16 inputs two bytes at a time).
template <typename TIN> struct Raw16 {          // generic / fallback: converted on the spot (double inputs)
    float e[16];
    __device__ __forceinline__ void load(const TIN* row, int k, int Din, bool valid) {
#pragma unroll
        for (int j = 0; j < 16; ++j) e[j] = (valid && k + j < Din) ? (float)in_load(row, k + j) : 0.f;
    }
    __device__ __forceinline__ float get(int j) const { return e[j]; }
};
template <> struct Raw16<float> {
    f32x4 v[4];
    __device__ __forceinline__ void load(const float* row, int k, int Din, bool valid) {
        if (valid && k + 16 <= Din && (reinterpret_cast<uintptr_t>(row + k) & 15) == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(row + k + 4 * q);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j >> 2][j & 3] = (valid && k + j < Din) ? row[k + j] : 0.f;
        }
    }
    __device__ __forceinline__ float get(int j) const { return v[j >> 2][j & 3]; }
};
template <int F16> struct Raw16Half {           // bf16 / fp16 inputs: two 16-byte loads
    u32x4 v[2];
    __device__ __forceinline__ void load(const unsigned short* row, int k, int Din, bool valid) {
        if (valid && k + 16 <= Din && (reinterpret_cast<uintptr_t>(row + k) & 15) == 0) {
            v[0] = *reinterpret_cast<const u32x4*>(row + k);
            v[1] = *reinterpret_cast<const u32x4*>(row + k + 8);
        } else {
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const unsigned lo = (valid && k + j < Din) ? row[k + j] : 0u, hi = (valid && k + j + 1 < Din) ? row[k + j + 1] : 0u;
                v[j >> 3][(j >> 1) & 3] = lo | (hi << 16);
            }
        }
    }
    __device__ __forceinline__ unsigned short bits(int j) const { return (unsigned short)(v[j >> 3][(j >> 1) & 3] >> (16 * (j & 1))); }
};
template <> struct Raw16<in_bf16> : Raw16Half<0> {
    __device__ __forceinline__ void load(const in_bf16* row, int k, int Din, bool valid) { Raw16Half<0>::load(reinterpret_cast<const unsigned short*>(row), k, Din, valid); }
    __device__ __forceinline__ float get(int j) const { return bf16_bits_to_f32(bits(j)); }
};
template <> struct Raw16<in_f16> : Raw16Half<1> {
    __device__ __forceinline__ void load(const in_f16* row, int k, int Din, bool valid) { Raw16Half<1>::load(reinterpret_cast<const unsigned short*>(row), k, Din, valid); }
    __device__ __forceinline__ float get(int j) const { in_f16 h; h.bits = bits(j); return (float)in_load(&h, 0); }
};

// grid = bpad / (32 RF); 4 waves: wave w owns output columns [Dpad/4 * w, Dpad/4 * (w+1)) of all 32 RF rows and both modalities.
// DKP = Dpad / 128 = 32-wide column fragments per wave (1..4: Dpad = 128 .. 512 with RF = 2 row fragments = 64 rows per block;
// 6 / 8: Dpad = 768 / 1024 with RF = 1 = 32 rows per block -- either way 2 x RF x DKP <= 16 accumulator tuples = 256 registers).
// WF: the weights arrive FRAGMENT-MAJOR -- [Dpad / 32][ldw / 16][64 lanes][8 bf16], lane (l31, half) of record (d32, ks) holding
// W[32 d32 + l31][16 ks + 8 half .. + 7] (rows beyond D zero) -- so that a wave's B fragment is ONE coalesced 1-KiB load instead of 64
// scattered 16-byte pieces of 64 different weight rows (the row-major form costs the kernel 3x: the texture addresser, not the MFMA pipe).
template <typename TIN, int DKP, int RF = 2, bool WF = false>
__global__ void __launch_bounds__(256, 1) project_pack_kernel(const TIN* xv, const TIN* xt, long ldv, long ldt, int Din_v, int Din_t,
                                                              const bf16_t* wv, const bf16_t* wt, int ldw_v, int ldw_t,
                                                              const float* bias_v, const float* bias_t, Geo g,
                                                              bf16_t* X, float* inv_norm, float* diag_cos) {
    constexpr int CF = DKP;                 // column fragments per wave
    constexpr int DP = DKP * 128;           // Dpad
    constexpr int KC = 64;                  // K chunk: one 128-byte K-tile row of bf16
    constexpr int ATILE = 64 * 128;         // 64 rows x 128 bytes
    constexpr int A0 = 0;                   // [2 buffers][2 modalities][ATILE]
    constexpr int R0 = 4 * ATILE;           // reduction scratch: [3 quantities][4 waves][64 rows] floats
    constexpr int LDS_MAIN = R0 + 3 * 4 * 64 * 4;
    constexpr int ROWS = 32 * RF;           // rows per block
    constexpr int OUT_BYTES = 2 * ROWS * DP * 2;   // the packed rows of both modalities, staged for coalesced stores
    static_assert(2 * RF * DKP <= 16, "accumulators: 2 modalities x RF x DKP tuples of 16 registers");
    static_assert(OUT_BYTES <= 160 * 1024, "LDS budget of the output staging");
    constexpr int LDS_BYTES = LDS_MAIN > OUT_BYTES ? LDS_MAIN : OUT_BYTES;
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int row0 = blockIdx.x * ROWS;

    f32x16 acc[2][RF][CF];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int rf = 0; rf < RF; ++rf)
#pragma unroll
            for (int cf = 0; cf < CF; ++cf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][rf][cf][r] = 0.f;

    // staging role: row srow, 16 consecutive k from 16 * spair
    const int srow = tid >> 2, spair = tid & 3;
    Raw16<TIN> raw[2];
    const bool srow_valid = row0 + srow < g.b && srow < ROWS;
    auto fetch = [&](int kc) {          // raw loads only: nothing here waits for them
        raw[0].load(xv + (size_t)(row0 + srow) * ldv, kc + 16 * spair, Din_v, srow_valid);
        raw[1].load(xt + (size_t)(row0 + srow) * ldt, kc + 16 * spair, Din_t, srow_valid);
    };
    auto commit = [&](int buf) {        // convert, write the two 16-byte K-tile chunks of this thread's row
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            struct { bf16_t e[8]; } pk[2];
#pragma unroll
            for (int j = 0; j < 16; ++j) pk[j >> 3].e[j & 7] = f32_to_bf16_bits(raw[m].get(j));
            unsigned char* tile = lds + A0 + (buf * 2 + m) * ATILE;
            *reinterpret_cast<u32x4*>(tile + ktile_off(srow, 2 * spair)) = __builtin_bit_cast(u32x4, pk[0]);
            *reinterpret_cast<u32x4*>(tile + ktile_off(srow, 2 * spair + 1)) = __builtin_bit_cast(u32x4, pk[1]);
        }
    };
    const int nchunks = ((Din_v > Din_t ? Din_v : Din_t) + KC - 1) / KC;     // (the shorter modality multiplies zeros in its last chunks)
    fetch(0);
    commit(0);
    __syncthreads();
    // The weight fragments of k-step g (16 input columns; global index g = 4 c + ks) sit in slot g % 4 of a rotating register set and are
    // requested THREE k-steps ahead of their MFMAs (with the loads right in front of their use every k-step exposed an L2 latency: 78 -> 61 us).
    // Four k-steps per chunk = four slots, so a slot is a compile-time index.  (The same depth for the INPUT chunks -- raw registers of four
    // chunks rotating -- measured slower, 61 -> 66 us: the inputs were not what the kernel waited for.)
    constexpr int NSLOT = (2 * RF * CF <= 8) ? 4 : 2;      // (256-accumulator instantiations keep two slots: one k-step of look-ahead)
    bf16x8 wb[NSLOT][2][CF];
    auto load_w = [&](auto sc, int gk) {                   // slot sc <- k-step gk
        constexpr int S = decltype(sc)::value;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const bf16_t* w = m == 0 ? wv : wt;
            const int ldw = m == 0 ? ldw_v : ldw_t;
#pragma unroll
            for (int cf = 0; cf < CF; ++cf) {
                const int d = CF * 32 * wave + 32 * cf + l31;
                const int k = 16 * gk + 8 * half;          // (the weights are zero-padded to a multiple of 64 columns)
                if constexpr (WF) {
                    if (k < ldw) wb[S][m][cf] = *reinterpret_cast<const bf16x8*>(w + (((size_t)(CF * wave + cf) * (ldw / 16) + gk) * 64 + lane) * 8);
                    else wb[S][m][cf] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
                } else if (d < g.D && k < ldw) wb[S][m][cf] = *reinterpret_cast<const bf16x8*>(w + (size_t)d * ldw + k);
                else wb[S][m][cf] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
            }
        }
    };
    static_for<NSLOT - 1>([&](auto sc) { load_w(sc, decltype(sc)::value); });
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) fetch((c + 1) * KC);     // in flight behind this chunk's MFMAs
        static_for<4>([&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            load_w(IdxC<(ks + NSLOT - 1) % NSLOT>{}, 4 * c + ks + NSLOT - 1);      // NSLOT - 1 k-steps ahead (past the end: zeros, never used)
            bf16x8 a[2][RF];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const unsigned char* tile = lds + A0 + (buf * 2 + m) * ATILE;
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) a[m][rf] = Operand<bf16_t>::load(tile, 32 * rf + l31, ks, half);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                    for (int cf = 0; cf < CF; ++cf) acc[m][rf][cf] = mfma_32x32x16_bf16(a[m][rf], wb[ks % NSLOT][m][cf], acc[m][rf][cf]);
        });
        if (c + 1 < nchunks) commit(buf ^ 1);          // (the other buffer: its readers finished before the last barrier)
        __syncthreads();
    }
    // ---- bias, row statistics: lane (l31, half) holds column d = CF*32*wave + 32 cf + l31 of rows 32 rf + frag_row(r, half) ----
    float* red = reinterpret_cast<float*>(lds + R0);
#pragma unroll
    for (int rf = 0; rf < RF; ++rf) {
        float ssv[16], sst[16], dot[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ssv[r] = sst[r] = dot[r] = 0.f;
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) {
            const int d = CF * 32 * wave + 32 * cf + l31;
            const float bv = (bias_v && d < g.D) ? bias_v[d] : 0.f, bt = (bias_t && d < g.D) ? bias_t[d] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float yv = acc[0][rf][cf][r] + bv, yt = acc[1][rf][cf][r] + bt;
                acc[0][rf][cf][r] = yv;
                acc[1][rf][cf][r] = yt;
                ssv[r] += yv * yv; sst[r] += yt * yt; dot[r] += yv * yt;
            }
        }
        const float sv = halving_sum16(ssv, l31), st = halving_sum16(sst, l31), sd = halving_sum16(dot, l31);
        if (l31 < 16) {
            const int row = 32 * rf + frag_row(halving_elem16(l31), half);
            red[(0 * 4 + wave) * 64 + row] = sv;
            red[(1 * 4 + wave) * 64 + row] = st;
            red[(2 * 4 + wave) * 64 + row] = sd;
        }
    }
    __syncthreads();
    float iv[RF][16], it[RF][16];
#pragma unroll
    for (int rf = 0; rf < RF; ++rf)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * rf + frag_row(r, half);
            const float sv = (red[(0 * 4 + 0) * 64 + row] + red[(0 * 4 + 1) * 64 + row]) + (red[(0 * 4 + 2) * 64 + row] + red[(0 * 4 + 3) * 64 + row]);
            const float st = (red[(1 * 4 + 0) * 64 + row] + red[(1 * 4 + 1) * 64 + row]) + (red[(1 * 4 + 2) * 64 + row] + red[(1 * 4 + 3) * 64 + row]);
            const float nv = sqrtf(sv), nt = sqrtf(st);
            const bool valid = row0 + row < g.b;
            iv[rf][r] = valid ? 1.f / (nv > 1e-12f ? nv : 1e-12f) : 0.f;     // x / max(||x||, eps), eps = 1e-12 (F.normalize default)
            it[rf][r] = valid ? 1.f / (nt > 1e-12f ? nt : 1e-12f) : 0.f;
        }
    if (wave == 0 && l31 == 0) {      // lanes 0 and 32 of wave 0 cover the block's rows between them (16 rows per fragment and half)
#pragma unroll
        for (int rf = 0; rf < RF; ++rf)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * rf + frag_row(r, half);
                const float sd = (red[(2 * 4 + 0) * 64 + row] + red[(2 * 4 + 1) * 64 + row]) + (red[(2 * 4 + 2) * 64 + row] + red[(2 * 4 + 3) * 64 + row]);
                inv_norm[row0 + row] = iv[rf][r];
                inv_norm[g.bpad + row0 + row] = it[rf][r];
                diag_cos[row0 + row] = sd * iv[rf][r] * it[rf][r];
            }
    }
    __syncthreads();     // (the staging area below overlaps the reduction scratch and the A tiles)
    // ---- unit rows -> bf16 -> LDS [modality][row][Dpad] -> 16-byte coalesced stores ----
    bf16_t* outl = reinterpret_cast<bf16_t*>(lds);
#pragma unroll
    for (int rf = 0; rf < RF; ++rf)
#pragma unroll
        for (int cf = 0; cf < CF; ++cf) {
            const int d = CF * 32 * wave + 32 * cf + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * rf + frag_row(r, half);
                outl[(0 * ROWS + row) * DP + d] = f32_to_bf16_bits(d < g.D ? acc[0][rf][cf][r] * iv[rf][r] : 0.f);
                outl[(1 * ROWS + row) * DP + d] = f32_to_bf16_bits(d < g.D ? acc[1][rf][cf][r] * it[rf][r] : 0.f);
            }
        }
    __syncthreads();
    constexpr int PIECES = 2 * ROWS * DP * 2 / 16;   // 16-byte pieces of the staged rows
    for (int i = tid; i < PIECES; i += 256) {
        const int m = i / (PIECES / 2), rem = i - m * (PIECES / 2);
        const int row = rem / (DP / 8), pc = rem - row * (DP / 8);
        const u32x4 v = *reinterpret_cast<const u32x4*>(lds + (size_t)i * 16);
        *reinterpret_cast<u32x4*>(X + ((size_t)m * g.bpad + row0 + row) * g.Dpad + 8 * pc) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// The projection's weight gradient on the chip's terms (round 4): dW_m[D][Din_m] = sum_i g_y,m[i][:]^T x_m[i][:]  (+ db_m = column sums of
// g_y,m) for both modalities in ONE launch.  Both operands carry the contraction index (the batch row) as their ROW index, i.e. neither is in
// MFMA operand order: 32-row chunks of both are staged row-major in LDS (pitch 288 bytes: the four rows of a transposing read fall into
// disjoint bank groups) and every A / B fragment is two ds_read_b64_tr_b16.  The library GEMM this replaces (hipBLASLt's choice for a
// [512 x 8192] x [8192 x 1024] product: 128 tiles of 64 x 64, no split-K, 59 us) leaves half the chip idle; here the batch is SPLIT over
// blockIdx.z so that ~256+ thread blocks exist, each split writes its own fp32 partial (no atomics: deterministic) and
// project_dw_reduce_kernel adds them in split order.
//   grid = (2 * nsplit, Dp / 128, Dinp / 128); block = 4 waves as 2 x 2, wave tile 64 x 64 (4 accumulator tuples).
//   gy: bf16 [b][ldgy]; x: TIN [b][ldx]; partial: [2][nsplit][Dp][Dinp] floats; dbpart: [2][nsplit][Dp] floats (written by the blocks
//   of the first Din tile).
// ---------------------------------------------------------------------------------------------
constexpr int kDwPitch = 288;                    // bytes per staged row: 128 bf16 + 32 bytes of padding
constexpr int kDwTile = 32 * kDwPitch;           // one operand chunk: 32 batch rows x 128 columns
template <typename TIN>
__global__ void __launch_bounds__(256) project_dw_kernel(const in_bf16* gyv, const in_bf16* gyt, long ldgy, const TIN* xv, const TIN* xt, long ldxv,
                                                         long ldxt, int b, int D, int Din_v, int Din_t, int nsplit, float* partial, int Dp,
                                                         int Dinp, float* dbpart) {
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[4 * kDwTile];      // [2 buffers][A | B]   (36 KiB; the db reduction reuses it)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wr = wave & 1, wc = wave >> 1;
    // blockIdx.x = (modality, split): block b runs on XCD b % 8, so the thread blocks of one XCD share ONE split's batch rows -- the
    // (Dp / 128) x (Dinp / 128) tiles of that split re-read the same 32-row chunks of g_y and x at about the same time, out of that XCD's L2
    // (with the tile index fastest, every re-read went to the Infinity Cache: 400 MB per launch, bandwidth-bound at 67 us)
    const int m = blockIdx.x / nsplit, split = blockIdx.x - m * nsplit;
    const int Din = m == 0 ? Din_v : Din_t;
    const int d0 = blockIdx.y * 128, n0 = blockIdx.z * 128;
    if (n0 >= Din) return;                                        // (the modality with the narrower input has fewer column tiles)
    const in_bf16* gy = m == 0 ? gyv : gyt;
    const TIN* x = m == 0 ? xv : xt;
    const long ldx = m == 0 ? ldxv : ldxt;
    const int rps = ((b + nsplit - 1) / nsplit + 31) / 32 * 32;   // batch rows per split (whole chunks)
    const int r0 = split * rps, r1 = r0 + rps < b ? r0 + rps : b;
    const bool want_db = blockIdx.z == 0;
    constexpr bool F16IN = __is_same(TIN, in_f16);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float dbacc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) dbacc[j] = 0.f;

    // staging role: chunk row srow (0..31), 16 consecutive columns from 16 * spiece.  The raw registers of FOUR chunks rotate: the loads of
    // chunk c + 3 are issued in iteration c and committed to LDS at the end of iteration c + 2 -- three iterations to come from HBM (with one
    // chunk of look-ahead every iteration exposed a full memory latency: 8 MFMAs per wave against ~1 us)
    const int srow = tid >> 3, spiece = tid & 7;
    constexpr int NR = 4;
    Raw16<in_bf16> ra[NR];
    Raw16<TIN> rb[NR];
    auto fetch = [&](auto kc, int r) {
        constexpr int k = decltype(kc)::value;
        const bool valid = r + srow < r1;
        ra[k].load(gy + (size_t)(r + srow) * ldgy, d0 + 16 * spiece, D, valid);
        rb[k].load(x + (size_t)(r + srow) * ldx, n0 + 16 * spiece, Din, valid);
    };
    auto commit = [&](auto kc, int buf) {
        constexpr int k = decltype(kc)::value;
        unsigned char* ta = lds + (2 * buf) * kDwTile + srow * kDwPitch + 32 * spiece;
        unsigned char* tb = lds + (2 * buf + 1) * kDwTile + srow * kDwPitch + 32 * spiece;
        *reinterpret_cast<u32x4*>(ta) = ra[k].v[0];            // g_y is bf16 already: the raw registers are the LDS image
        *reinterpret_cast<u32x4*>(ta + 16) = ra[k].v[1];
        if constexpr (sizeof(TIN) == 2 && !F16IN) {            // bf16 inputs: likewise
            *reinterpret_cast<u32x4*>(tb) = rb[k].v[0];
            *reinterpret_cast<u32x4*>(tb + 16) = rb[k].v[1];
        } else {
            struct { bf16_t e[8]; } pb[2];
#pragma unroll
            for (int j = 0; j < 16; ++j) pb[j >> 3].e[j & 7] = f32_to_bf16_bits(rb[k].get(j));
            *reinterpret_cast<u32x4*>(tb) = __builtin_bit_cast(u32x4, pb[0]);
            *reinterpret_cast<u32x4*>(tb + 16) = __builtin_bit_cast(u32x4, pb[1]);
        }
        if (want_db) {                                          // (block-uniform: one block in Dinp / 128 sums the columns)
#pragma unroll
            for (int j = 0; j < 16; ++j) dbacc[j] += ra[k].get(j);
        }
    };
    // transposing read of one fragment: in a 16-lane group lane 4 j + c addresses chunk row k0 + j, 8-byte piece c of the 16 columns the
    // group covers; lane i then receives the 4 rows k0 .. k0 + 3 of column i (hip_emu.h / MI355X ISA: ds_read_b64_tr_b16)
    const int grp = lane >> 4, i16 = lane & 15;
    const int tr_off = (8 * (grp >> 1) + (i16 >> 2)) * kDwPitch + (16 * (grp & 1) + 4 * (i16 & 3)) * 2;
    auto frag = [&](const unsigned char* tile, int col0, int ks) {
        struct { s16x4 lo, hi; } p;
        const unsigned char* a = tile + tr_off + 16 * ks * kDwPitch + col0 * 2;
        p.lo = lds_read_tr16_b64(a);
        p.hi = lds_read_tr16_b64(a + 4 * kDwPitch);
        return __builtin_bit_cast(bf16x8, p);
    };

    if (r0 < r1) {
        fetch(IdxC<0>{}, r0);
        fetch(IdxC<1>{}, r0 + 32);          // (rows past r1 load nothing: zeros)
        fetch(IdxC<2>{}, r0 + 64);
        commit(IdxC<0>{}, 0);
        __syncthreads();
        for (int rr = r0; rr < r1; rr += 32 * NR) {
            static_for<NR>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const int r = rr + 32 * k;
                if (r < r1) {                                     // (block-uniform)
                    constexpr int buf = k & 1;
                    fetch(IdxC<(k + 3) % NR>{}, r + 96);          // three chunks ahead
                    const unsigned char* ta = lds + (2 * buf) * kDwTile;
                    const unsigned char* tb = lds + (2 * buf + 1) * kDwTile;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        bf16x8 a[2], bq[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) { a[i] = frag(ta, 64 * wr + 32 * i, ks); bq[i] = frag(tb, 64 * wc + 32 * i, ks); }
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x16_bf16(a[i], bq[j], acc[i][j]);
                    }
                    if (r + 32 < r1) commit(IdxC<(k + 1) % NR>{}, buf ^ 1);
                    __syncthreads();
                }
            });
        }
    }
    // C fragment (i, j): lane (l31, half) holds dW[d0 + 64 wr + 32 i + frag_row(r, half)][n0 + 64 wc + 32 j + l31]
    float* out = partial + ((size_t)(m * nsplit + split) * Dp + d0) * Dinp + n0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(size_t)(64 * wr + 32 * i + frag_row(r, half)) * Dinp + 64 * wc + 32 * j + l31] = acc[i][j][r];
    if (want_db) {      // column sums of this block's g_y rows: thread (srow, spiece) holds 16 columns of every 32nd row
        float* red = reinterpret_cast<float*>(lds);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) red[srow * 128 + 16 * spiece + j] = dbacc[j];
        __syncthreads();
        if (tid < 128) {
            float sum = 0.f;
            for (int q = 0; q < 32; ++q) sum += red[q * 128 + tid];
            dbpart[(size_t)(m * nsplit + split) * Dp + d0 + tid] = sum;
        }
    }
}
// dW_m[d][n] = sum_split partial[m][split][d][n] (split order: deterministic), db_m[d] likewise.  grid = (ceil(Dinp / 256), D, 2).
__global__ void __launch_bounds__(256) project_dw_reduce_kernel(const float* partial, const float* dbpart, int nsplit, int D, int Dp, int Dinp,
                                                                int Din_v, int Din_t, float* dwv, float* dwt, long lddwv, long lddwt,
                                                                float* dbv, float* dbt) {
    const int m = blockIdx.z, d = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    const int Din = m == 0 ? Din_v : Din_t;
    if (n < Din) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += partial[((size_t)(m * nsplit + k) * Dp + d) * Dinp + n];
        (m == 0 ? dwv : dwt)[(size_t)d * (m == 0 ? lddwv : lddwt) + n] = s;
    }
    float* db = m == 0 ? dbv : dbt;
    if (db && blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += dbpart[(size_t)(m * nsplit + k) * Dp + d];
        db[d] = s;
    }
}

// one wave per row index i (both modalities): g_y = inv_norm (G - y^ (y^ . G)), written as fp32 [b, D] per modality
__global__ void __launch_bounds__(256) project_backward_prep_kernel(const float* gv, const float* gt, long ldgv, long ldgt, Geo g,
                                                                    const bf16_t* X, const float* inv_norm, float* ov, float* ot,
                                                                    long ldo) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= g.b) return;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const float* G = m == 0 ? gv + (size_t)i * ldgv : gt + (size_t)i * ldgt;
        const bf16_t* y = X + ((size_t)m * g.bpad + i) * g.Dpad;
        float* o = (m == 0 ? ov : ot) + (size_t)i * ldo;
        double dot = 0.0;
        for (int d = lane; d < g.D; d += 64) dot += (double)bf16_bits_to_f32(y[d]) * (double)G[d];
        dot = wave_sum_f64(dot);
        const float inv = inv_norm[m * g.bpad + i], dt = (float)dot;
        for (int d = lane; d < g.D; d += 64) o[d] = inv * (G[d] - bf16_bits_to_f32(y[d]) * dt);
    }
}

}  // namespace crossclr
