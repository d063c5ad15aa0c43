// crossclr_kernels_generic.h -- the tiled kernels that work for every shape and both operand
// types (exact-fp32 MFMA and bf16 MFMA): normalisation, forward denominators, forward finish,
// backward, backward finish.  The register-resident bf16 fast path lives in
// crossclr_kernels_fast.h; this file is what it falls back to and what compute_mode="fp32" runs.
#pragma once
#include "crossclr_device.h"

namespace crossclr {

// ---------------------------------------------------------------------------------------------
// input element access (any float dtype of the reference's inputs)
// ---------------------------------------------------------------------------------------------
struct in_f16 { unsigned short bits; };
struct in_bf16 { unsigned short bits; };
__device__ __forceinline__ double in_load(const float* p, size_t i) { return (double)p[i]; }
__device__ __forceinline__ double in_load(const double* p, size_t i) { return p[i]; }
__device__ __forceinline__ double in_load(const in_bf16* p, size_t i) { return (double)bf16_bits_to_f32(p[i].bits); }
__device__ __forceinline__ double in_load(const in_f16* p, size_t i) {
    return (double)(float)__builtin_bit_cast(_Float16, p[i].bits);
}
__device__ __forceinline__ void in_store(float* p, size_t i, double v) { p[i] = (float)v; }
__device__ __forceinline__ void in_store(double* p, size_t i, double v) { p[i] = v; }
__device__ __forceinline__ void in_store(in_bf16* p, size_t i, double v) { p[i].bits = f32_to_bf16_bits((float)v); }
__device__ __forceinline__ void in_store(in_f16* p, size_t i, double v) {
    p[i].bits = __builtin_bit_cast(unsigned short, (_Float16)(float)v);
}
__device__ __forceinline__ void op_store(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void op_store(bf16_t* p, size_t i, float v) { p[i] = f32_to_bf16_bits(v); }

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += wave_xor_f64(v, m);
    return v;
}

// ---------------------------------------------------------------------------------------------
// row access helpers for the memory-bound row kernels: a lane owns 4 consecutive elements of every
// 256-element stretch of its row (16-byte accesses when the row is fp32 and 16-byte aligned),
// and keeps them in registers so the row is read from HBM exactly once.
// ---------------------------------------------------------------------------------------------
constexpr int kRowCache = 4;  // stretches of 256 elements cached per lane (D <= 1024)

template <typename TIN>
__device__ __forceinline__ void row_load4(const TIN* row, int d, int D, double (&out)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = (d + j < D) ? in_load(row, d + j) : 0.0;
}
template <>
__device__ __forceinline__ void row_load4<float>(const float* row, int d, int D, double (&out)[4]) {
    if (d + 3 < D && ((reinterpret_cast<uintptr_t>(row + d) & 15) == 0)) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + d);
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = (double)v[j];
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = (d + j < D) ? (double)row[d + j] : 0.0;
    }
}
template <typename TIN>
__device__ __forceinline__ void row_store4(TIN* row, int d, int D, const double (&v)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (d + j < D) in_store(row, d + j, v[j]);
}
template <>
__device__ __forceinline__ void row_store4<float>(float* row, int d, int D, const double (&v)[4]) {
    if (d + 3 < D && ((reinterpret_cast<uintptr_t>(row + d) & 15) == 0)) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (float)v[j];
        *reinterpret_cast<f32x4*>(row + d) = o;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (d + j < D) row[d + j] = (float)v[j];
    }
}
// packed-operand rows are padded to a multiple of 64 elements and 16-byte aligned: whole 4-element stores
__device__ __forceinline__ void op_store4(float* row, int d, const float (&v)[4]) {
    f32x4 o = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(row + d) = o;
}
__device__ __forceinline__ void op_store4(bf16_t* row, int d, const float (&v)[4]) {
    u32x2 o;
    o[0] = (uint32_t)f32_to_bf16_bits(v[0]) | ((uint32_t)f32_to_bf16_bits(v[1]) << 16);
    o[1] = (uint32_t)f32_to_bf16_bits(v[2]) | ((uint32_t)f32_to_bf16_bits(v[3]) << 16);
    *reinterpret_cast<u32x2*>(row + d) = o;
}

// ---------------------------------------------------------------------------------------------
// K0: row L2-normalisation of both modalities  (reference trainer/loss.py:79-80)
// one wavefront per row index i: video row i and text row i together, so the diagonal cosine
// vhat_i . that_i (the positive-pair logit of loss.py:83 before /tau) comes out in fp32 for free.
// HBM-bound: reads 2*B*D inputs once, writes the packed operand once.
// ---------------------------------------------------------------------------------------------
// NORM = false ("pack"): the rows are already unit vectors (the producer's projection head normalised them): they are only
// cast / laid out into the packed operand, inv_norm = 1, and the positive-pair cosine is still formed in fp32 here.
template <typename TIN, typename T, bool NORM = true>
__global__ void __launch_bounds__(256) normalize_kernel(const TIN* video, const TIN* text, long ldv, long ldt,
                                                        Geo g, T* X, float* inv_norm, float* diag_cos, int* zero_word) {
    // (crossclr_step_forward: the step's first kernel clears the ticket its finish kernel's last block is found with)
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= g.bpad) return;
    T* xv = X + (size_t)i * g.Dpad;
    T* xt = X + ((size_t)g.bpad + i) * g.Dpad;
    const float zero4[4] = {0.f, 0.f, 0.f, 0.f};
    if (i >= g.b) {  // padding rows: zeros (their columns are masked, their rows ignored)
        for (int d = 4 * lane; d < g.Dpad; d += 256) { op_store4(xv, d, zero4); op_store4(xt, d, zero4); }
        if (lane == 0) { inv_norm[i] = 0.f; inv_norm[g.bpad + i] = 0.f; diag_cos[i] = 0.f; }
        return;
    }
    const TIN* pv = video + (size_t)i * ldv;
    const TIN* pt = text + (size_t)i * ldt;
    const bool cached = g.D <= 256 * kRowCache;
    double cv[kRowCache][4], ct[kRowCache][4];
    double ssv = 0, sst = 0, dot = 0;
    if (cached) {
#pragma unroll
        for (int k = 0; k < kRowCache; ++k) {
            const int d = 4 * lane + 256 * k;
            if (d < g.D) {
                row_load4(pv, d, g.D, cv[k]);
                row_load4(pt, d, g.D, ct[k]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { ssv += cv[k][j] * cv[k][j]; sst += ct[k][j] * ct[k][j]; dot += cv[k][j] * ct[k][j]; }
            }
        }
    } else {
        for (int d = lane; d < g.D; d += 64) {
            double a = in_load(pv, d), c = in_load(pt, d);
            ssv += a * a; sst += c * c; dot += a * c;
        }
    }
    ssv = wave_sum_f64(ssv); sst = wave_sum_f64(sst); dot = wave_sum_f64(dot);
    // x / max(||x||, eps), eps = 1e-12 (F.normalize default)
    double nv = sqrt(ssv), nt = sqrt(sst);
    double iv = 1.0 / (nv > 1e-12 ? nv : 1e-12), it = 1.0 / (nt > 1e-12 ? nt : 1e-12);
    if (!NORM) iv = it = 1.0;
    if (cached) {
#pragma unroll
        for (int k = 0; k < kRowCache; ++k) {
            const int d = 4 * lane + 256 * k;
            if (d < g.Dpad) {
                float a[4] = {0.f, 0.f, 0.f, 0.f}, c[4] = {0.f, 0.f, 0.f, 0.f};
                if (d < g.D) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[j] = (float)(cv[k][j] * iv); c[j] = (float)(ct[k][j] * it); }
                }
                op_store4(xv, d, a); op_store4(xt, d, c);
            }
        }
        for (int d = 4 * lane + 256 * kRowCache; d < g.Dpad; d += 256) { op_store4(xv, d, zero4); op_store4(xt, d, zero4); }
    } else {
        for (int d = lane; d < g.Dpad; d += 64) {
            float a = 0.f, c = 0.f;
            if (d < g.D) { a = (float)(in_load(pv, d) * iv); c = (float)(in_load(pt, d) * it); }
            op_store(xv, d, a); op_store(xt, d, c);
        }
    }
    if (lane == 0) {
        inv_norm[i] = (float)iv; inv_norm[g.bpad + i] = (float)it;
        diag_cos[i] = (float)(dot * iv * it);
    }
}

// K0x: the same normalisation (term by term) that ALSO leaves the bf16 unit rows in the fragment-major layout the XF backward loads
// straight into MFMA B fragments (crossclr_kernels_dsl.h):
//   XF[u = stacked row / 32][dt = column / 32][ks][lane = 32 h + n][e = 0..7] = xhat[32 u + 16 ks + 8 (e >> 2) + 4 h + (e & 3)][32 dt + n]
// (1 KiB per (u, dt, ks); the row order inside a lane is the k <-> q permutation of the saved exponentials' A fragments).
// A block = 16 row pairs (one k-step of a tile, two per wave, 8 waves): the 32 unit rows pass through LDS once and leave as 16-byte
// chunks -- 8 rows x 1 column -- in runs of 512 bytes.  Dpad <= 256 KC (KC = 2: Dpad <= 512, KC = 4: the wide operands up to 1024);
// +16 MiB written at B = 8192, D = 512.
template <typename TIN, bool NORM, int KC>
__global__ void __launch_bounds__(512) normalize_xf_kernel(const TIN* video, const TIN* text, long ldv, long ldt, Geo g, bf16_t* X,
                                                           unsigned char* XF, float* inv_norm, float* diag_cos, int* zero_word) {
    CROSSCLR_SHARED __attribute__((aligned(16))) bf16_t sh[2][16][256 * KC];
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = blockIdx.x * 16;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {     // (unrolled: the second pair's loads are in flight while the first pair is reduced)
        const int r = 2 * wave + rr, i = i0 + r;     // (bpad is a multiple of 128: every i < bpad)
        bf16_t* xv = X + (size_t)i * g.Dpad;
        bf16_t* xt = X + ((size_t)g.bpad + i) * g.Dpad;
        double cv[KC][4], ct[KC][4];
        double ssv = 0, sst = 0, dot = 0;
        const bool valid = i < g.b;
        const TIN* pv = video + (size_t)(valid ? i : 0) * ldv;
        const TIN* pt = text + (size_t)(valid ? i : 0) * ldt;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int d = 4 * lane + 256 * k;
#pragma unroll
            for (int j = 0; j < 4; ++j) { cv[k][j] = 0.0; ct[k][j] = 0.0; }
            if (valid && d < g.D) {
                row_load4(pv, d, g.D, cv[k]);
                row_load4(pt, d, g.D, ct[k]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { ssv += cv[k][j] * cv[k][j]; sst += ct[k][j] * ct[k][j]; dot += cv[k][j] * ct[k][j]; }
            }
        }
        ssv = wave_sum_f64(ssv); sst = wave_sum_f64(sst); dot = wave_sum_f64(dot);
        double nv = sqrt(ssv), nt = sqrt(sst);
        double iv = 1.0 / (nv > 1e-12 ? nv : 1e-12), it = 1.0 / (nt > 1e-12 ? nt : 1e-12);
        if (!NORM) iv = it = 1.0;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int d = 4 * lane + 256 * k;
            if (d < g.Dpad) {
                float a[4] = {0.f, 0.f, 0.f, 0.f}, c[4] = {0.f, 0.f, 0.f, 0.f};
                if (valid && d < g.D) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[j] = (float)(cv[k][j] * iv); c[j] = (float)(ct[k][j] * it); }
                }
                op_store4(xv, d, a); op_store4(xt, d, c);
                op_store4(&sh[0][r][0], d, a); op_store4(&sh[1][r][0], d, c);
            }
        }
        if (lane == 0) {
            inv_norm[i] = valid ? (float)iv : 0.f; inv_norm[g.bpad + i] = valid ? (float)it : 0.f;
            diag_cos[i] = valid ? (float)(dot * iv * it) : 0.f;
        }
    }
    __syncthreads();
    const int nfr = g.Dpad / 32, ks = (i0 >> 4) & 1;
    for (int c = threadIdx.x; c < 4 * g.Dpad; c += 512) {
        const int m = c / (2 * g.Dpad), rest = c - m * 2 * g.Dpad, h = rest / g.Dpad, d = rest - h * g.Dpad;
        struct { bf16_t e[8]; } v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v.e[e] = sh[m][8 * (e >> 2) + 4 * h + (e & 3)][d];
        const size_t u = ((size_t)m * g.bpad + i0) >> 5;
        *reinterpret_cast<u32x4*>(XF + ((u * nfr + (d >> 5)) * 2 + ks) * 1024 + (32 * h + (d & 31)) * 16) = __builtin_bit_cast(u32x4, v);
    }
}

// K0y: the fragment-major copy of operands that are ALREADY packed (row-major bf16 [rows][Dpad], rows a multiple of 32): what a rank does
// with the slices it RECEIVED from other ranks, so that the saved backwards of remote blocks can load their column tiles as MFMA B
// fragments too (a gathered operand travels once, row-major; the second layout is made where it is needed).  A block = one tile of 32
// rows: through LDS once, out as 16-byte chunks (same map as normalize_xf_kernel).  HBM-bound: reads and writes the operand once.
__global__ void __launch_bounds__(256) xf_from_packed_kernel(const bf16_t* X, unsigned char* XF, int Dpad) {
    CROSSCLR_SHARED __attribute__((aligned(16))) bf16_t sh[32][1024 + 8];     // (+8: rows 16 bytes apart in the bank map)
    const size_t u = blockIdx.x;
    const int col0 = blockIdx.y * 1024;             // wide plans (Dpad > 1024): 1024 columns of the tile per block (grid.y = ceil(Dpad / 1024))
    const int wcols = Dpad - col0 < 1024 ? Dpad - col0 : 1024;
    const bf16_t* src = X + u * 32 * (size_t)Dpad + col0;
    const int chunks = wcols / 8;                   // 16-byte pieces per row
    for (int c = threadIdx.x; c < 32 * chunks; c += 256) {
        const int r = c / chunks, k = c - r * chunks;
        *reinterpret_cast<u32x4*>(&sh[r][8 * k]) = *reinterpret_cast<const u32x4*>(src + (size_t)r * Dpad + 8 * k);
    }
    __syncthreads();
    const int nfr = Dpad / 32;
    for (int c = threadIdx.x; c < 4 * wcols; c += 256) {     // (ks, h, column d) -> one 16-byte chunk of the copy
        const int ks = c / (2 * wcols), rest = c - ks * 2 * wcols, h = rest / wcols, d = rest - h * wcols;
        struct { bf16_t e[8]; } v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v.e[e] = sh[16 * ks + 8 * (e >> 2) + 4 * h + (e & 3)][d];
        const int dg = col0 + d;
        *reinterpret_cast<u32x4*>(XF + ((u * nfr + (dg >> 5)) * 2 + ks) * 1024 + (32 * h + (dg & 31)) * 16) = __builtin_bit_cast(u32x4, v);
    }
}

// ---------------------------------------------------------------------------------------------
// K1: forward denominators, generic tiled version.
//   block = 256 threads (4 waves as 2x2), tile = 128 rows x 128 columns, K-chunks of 128 bytes.
//   MFMA operands are SWAPPED (A = column rows, B = row rows) so that a lane owns one output ROW
//   p = lane&31 of each 32x32 fragment and 16 of its columns: the running row sum is one
//   register per fragment row-block, no cross-lane traffic until the very end.
//   grid = (2*bpad/128 row blocks, nsplit column splits); each (row block, split) writes its own
//   slot -> no atomics, deterministic.
// ---------------------------------------------------------------------------------------------
//   SW (sample weights): the exponential of an INTRA-modal column q is multiplied by kcols[q] (include/crossclr.h).
//   MODE 0: sums of exp2(x - g.m2) (one shift for everything).  Small temperatures (Geo::row_shift) take two passes:
//   MODE 1: the slot receives the row MAXIMUM of the scaled logits x over the slot's unmasked columns (columns with k_q = 0 do
//           not count: they are not in the soft-max), MODE 2: sums of exp2(x - shift[p]) with the per-row shift of pass 1.
//   ST (save for backward; MODE 0, rows and columns the same local operand): every 32x32 fragment of exponentials (masked
//   entries as 0, WITHOUT the k_q factor) is also written to `stash` in the registers' own layout -- fragment (p32, q32) of the
//   stacked [2 bpad] x [2 bpad] matrix at ((p32 * (2 bpad / 32)) + q32) * 4 KiB, inside it [r4][lane][4 floats]: lane
//   (p = lane & 31, half) holds E[32 p32 + p][32 q32 + 8 r4 + 4 half + j] -- what bwd_saved32_kernel loads back, 16 bytes per
//   lane, fully coalesced, no transposition anywhere (the generic forward evaluates both triangles).
//   Score statistics of the inter-modal block (the max-margin ranking loss of trainer/loss.py:29-41
//   and retrieval ranks), same tiling, no soft-max: the cosines S[p][q] of row p against the OTHER modality's columns,
//   MODE 4: the slot receives S[p][partner(p)] (the positive pair's cosine, from the very MFMA sequence MODE 3 compares with:
//           S > S_pp is then exact, a tie with oneself impossible),
//   MODE 3: hinge sums  sum_q max(0, g.m2 + S[p][q] - shift[p])  over q != partner into `part`, and the number of active
//           terms (g.m2 + S - shift[p] > 0) into `stash` (same slot layout, as floats); shift = the MODE 4 result, g.m2 = margin.
//   SYM (MODE 0, rows and columns the same local operand, nothing saved): the stacked matrix of exponentials is symmetric, so row
//   block I evaluates only the column tiles t >= I (split y takes t = I + y, I + y + nsplit, ...); every tile right of the
//   diagonal one also leaves its 128 column sums over the block's rows in colpart[I][128 t ..] -- the row sums of the
//   mirrored tile that is never evaluated -- and, with ST, stores each of its fragments a second time, transposed, where the
//   mirrored tile's fragment belongs (16 scalar stores per lane and fragment: the stash stays the full matrix the backward reads).  The launch announces itself in `header` as kind 4 (dense slots +
//   column sums of the row blocks above); fwd_finish_kernel adds them up in a fixed order.  4.06 B^2 D instead of 8 B^2 D executed.
template <typename T, bool SW, int MODE, bool ST = false, bool SYM = false>
__global__ void __launch_bounds__(256, 2) fwd_sums_kernel(const T* rows, const T* cols, Geo g, int tiles_per_split,
                                                       float* part, const float* kcols, const float* shift, float* stash, int* header,
                                                       float* colpart) {
    static_assert(!ST || MODE == 0 || MODE == 2, "exponentials are saved by the passes that form sums");
    static_assert(!(ST && sizeof(T) == 2) || MODE == 0 || (MODE == 2 && !SYM),
                  "bf16 records: the single-pass forward (symmetric: upper triangle; rectangular: every tile) or the full second pass of the two-pass soft-max");
    static_assert(!SYM || MODE <= 3, "symmetric evaluation: the soft-max passes; MODE 3: one pass for both directions");
    // MODE 3 + SYM (score statistics in ONE pass): only the rows of modality 0 are walked (grid.x = bpad / 128), against the column
    // tiles of modality 1; every tile also yields, per column q, the hinge sum and the active count over the block's rows against
    // the COLUMN's own positive-pair score shift[q] -- the statistics of the stacked rows of modality 1 -- into
    // colpart[rb][q] (hinges) and colpart[bpad / 128 + rb][q] (counts).  The score matrix is evaluated once, not once per direction.
    constexpr bool PAIRED = SYM && MODE != 3;      // the paired-row-block walk of the symmetric soft-max passes
    // MODE 2 (per-row shifts): U[p][q] = exp2(x - shift[p]) is NOT symmetric.  The mirrored tile's sums, and the backward
    // (weight U[p][q] rz_p + U[q][p] rz_q), need exp2(x - shift[q]) as well: a second exponential per element; ST keeps both
    // matrices -- U, and Ut[p][q] = U[q][p] behind it at stash + (2 bpad)^2 -- in the layout of the single-pass stash.
    typedef Operand<T> Op;
    // MODE 2 over OTHER ranks' columns (rectangular launch, not SYM: `colpart` is free): the columns' shifts come from the gathered
    // [world][2 bpad] array passed in its place, indexed like the gathered statistics; the rows' shifts stay `shift`
    const float* shift_q = (!SYM && MODE == 2 && colpart != nullptr) ? colpart : shift;
    // MODE 3 + SYM (`header` is unused there: no launch header): when not NULL it is the byte mask crossclr_score_rows_save asks for --
    // per (video row, text column) the number of active hinges of the pair, 0 / 1 / 2 (0 for the positive pair and for padding), what
    // maxmargin_saved_kernel multiplies the other modality's rows with instead of evaluating the scores again
    unsigned char* hinge_mask = (MODE == 3 && SYM) ? reinterpret_cast<unsigned char*>(header) : nullptr;
    (void)hinge_mask;
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[2 * 128 * 128 + 2 * 128 * 4];
    unsigned char* tileP = lds;                 // row operand chunk   [128][128 B]
    unsigned char* tileQ = lds + 128 * 128;     // column operand chunk [128][128 B]
    float* red = reinterpret_cast<float*>(lds + 2 * 128 * 128);  // [2][128]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wr = wave & 1, wc = wave >> 1;
    const size_t pitch = (size_t)g.Dpad * sizeof(T);
    const int nchunks = g.Dpad / Op::kChunkElems;

    const int ntiles = g.col_ranks * 2 * g.bpad / 128;
    if (PAIRED && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { header[0] = 4; header[1] = 4; header[2] = 0; header[3] = 0; }
    // SYM: row block I has ntiles - I tiles; it is paired with row block J = ntiles - 1 - I (I + 1 tiles) so that every
    // blockIdx.x owns ntiles + 1 tiles, cut evenly over the column splits: equal work for every thread block.  A block
    // therefore walks up to two segments (rows of I, then rows of J) and writes slot blockIdx.y of both row blocks.
    const int sym_total = ntiles + 1;
    const int sym_k0 = PAIRED ? (int)((long)blockIdx.y * sym_total / (long)gridDim.y) : 0;
    const int sym_k1 = PAIRED ? (int)((long)(blockIdx.y + 1) * sym_total / (long)gridDim.y) : 0;
    for (int seg = 0; seg < (PAIRED ? 2 : 1); ++seg) {
    const int rbk = !PAIRED ? (int)blockIdx.x : (seg == 0 ? (int)blockIdx.x : ntiles - 1 - (int)blockIdx.x);   // row block of this segment
    const int seg_first = seg == 0 ? 0 : ntiles - (int)blockIdx.x;                                          // its place in the block's tile list
    const int row0 = rbk * 128;                  // first row of the segment inside [2][bpad]
    const int rmod = row0 / g.bpad;
    const int r_in_mod0 = row0 - rmod * g.bpad;
    const unsigned char* rbase = reinterpret_cast<const unsigned char*>(rows) + (size_t)row0 * pitch;

    int t_begin = PAIRED ? rbk + (sym_k0 > seg_first ? sym_k0 - seg_first : 0)
                         : (SYM ? ntiles / 2 : 0) + (int)blockIdx.y * tiles_per_split;      // (MODE 3 + SYM: the tiles of modality 1)
    int t_end = PAIRED ? rbk + (sym_k1 - seg_first) : t_begin + tiles_per_split;
    if (t_end > ntiles) t_end = ntiles;
    const int t_step = 1;
    float kp[2] = {1.f, 1.f};       // SYM + SW: k of this lane's rows (they are the mirrored tile's intra-modal negative columns)
    if (PAIRED && SW) { kp[0] = kcols[row0 + 64 * wr + l31]; kp[1] = kcols[row0 + 64 * wr + 32 + l31]; }

    const float kNone = -3.0e38f;   // "no unmasked column yet" (MODE 1)
    float rowacc[2] = {MODE == 1 ? kNone : 0.f, MODE == 1 ? kNone : 0.f};
    float myshift[2] = {0.f, 0.f};
    if (MODE == 2 || MODE == 3) { myshift[0] = shift[row0 + 64 * wr + l31]; myshift[1] = shift[row0 + 64 * wr + 32 + l31]; }
    float rowcnt[2] = {0.f, 0.f};   // MODE 3: active hinge terms
    KTileStage<128, 256> sp, sq, sp2, sq2;

    for (int t = t_begin; t < t_end; t += t_step) {
        const ColTile ct = col_tile(g, t, 128);
        if (ct.rank == g.skip_rank) continue;
        if (MODE >= 3 && ct.mod == rmod) continue;                                                  // inter-modal block only
        if (MODE == 4 && !(ct.rank == g.row_rank && ct.in_mod0 == r_in_mod0)) continue;             // the partners' tile only
        const unsigned char* cbase = reinterpret_cast<const unsigned char*>(cols) + ct.row0 * pitch;
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

        // Two chunks in flight: while chunk kc is multiplied, the loads of chunks kc + 1 and kc + 2 are outstanding (two
        // register sets, alternating).  One chunk of look-ahead left a block waiting on HBM/L2 every chunk: 16 MFMAs of
        // work per wave cover a quarter of a round trip.
        auto chunk = [&](KTileStage<128, 256>& fp, KTileStage<128, 256>& fq, int kc) {
            fp.commit(tileP, tid);
            fq.commit(tileQ, tid);
            __syncthreads();
            if (kc + 2 < nchunks) {
                fp.fetch(rbase, pitch, (kc + 2) * 128, tid);
                fq.fetch(cbase, pitch, (kc + 2) * 128, tid);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                typename Op::frag a[2], bfr[2];
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    a[x] = Op::load(tileQ, 64 * wc + 32 * x + l31, s, half);
                    bfr[x] = Op::load(tileP, 64 * wr + 32 * x + l31, s, half);
                }
#pragma unroll
                for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                    for (int pi = 0; pi < 2; ++pi) acc[qi][pi] = Op::mma(a[qi], bfr[pi], acc[qi][pi]);
            }
            __syncthreads();
        };
        sp.fetch(rbase, pitch, 0, tid);
        sq.fetch(cbase, pitch, 0, tid);
        if (nchunks > 1) {
            sp2.fetch(rbase, pitch, 128, tid);
            sq2.fetch(cbase, pitch, 128, tid);
        }
        for (int kc = 0; kc < nchunks; kc += 2) {
            chunk(sp, sq, kc);
            if (kc + 1 < nchunks) chunk(sp2, sq2, kc + 1);
        }
        // epilogue: e = exp2(c*g - m2), masked, summed into the lane's row
        const bool same_mod = (ct.mod == rmod);
        const float c2 = same_mod ? g.c_intra : g.c_inter;
        const bool diag_tile = (MODE >= 3 ? !same_mod : same_mod) && ct.rank == g.row_rank && ct.in_mod0 == r_in_mod0;
        const bool ragged = ct.in_mod0 + 128 > g.b;
        const bool mirror = SYM && (MODE == 3 || t > rbk);     // this tile also stands for its never-evaluated mirror image
        float es[2][16], ec[2][16];             // column side: sums / maxima / hinge sums; MODE 3: active counts
        if (SYM) {
#pragma unroll
            for (int qi = 0; qi < 2; ++qi)
#pragma unroll
                for (int r = 0; r < 16; ++r) { es[qi][r] = MODE == 1 ? -3.0e38f : 0.f; ec[qi][r] = 0.f; }
        }
#pragma unroll
        for (int qi = 0; qi < 2; ++qi)
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                const int p_t = 64 * wr + 32 * pi + l31;
                const bool pad_row = r_in_mod0 + p_t >= g.b;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 kq = {1.f, 1.f, 1.f, 1.f};
                    if (SW && same_mod) kq = *reinterpret_cast<const f32x4*>(kcols + ct.stat0 + 64 * wc + 32 * qi + 8 * r4 + 4 * half);
                    f32x4 ev = {0.f, 0.f, 0.f, 0.f}, etv = {0.f, 0.f, 0.f, 0.f}, shq = {0.f, 0.f, 0.f, 0.f};
                    constexpr bool kNeedColShift = (MODE == 2 && (SYM || (ST && sizeof(T) == 4))) || (MODE == 3 && SYM);   // (bf16 records: U only)
                    // bf16 records of a rectangular second pass (other ranks' columns: shift_q is its own array): Ut as well, at run time
                    const bool ut_records = ST && sizeof(T) == 2 && MODE == 2 && !SYM && colpart != nullptr;   // (not `shift_q != shift`: rank 0's rows may BE the head of the gathered array)
                    if (kNeedColShift || ut_records)
                        shq = *reinterpret_cast<const f32x4*>(shift_q + ct.stat0 + 64 * wc + 32 * qi + 8 * r4 + 4 * half);
                    unsigned hinge_bytes = 0;      // (MODE 3 + SYM: active hinges of the four pairs, one byte each)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * r4 + j;
                        const int q_t = 64 * wc + 32 * qi + frag_row(r, half);
                        const bool masked = (ragged && ct.in_mod0 + q_t >= g.b) || (diag_tile && q_t == p_t);
                        if (MODE == 1) {
                            const float x = acc[qi][pi][r] * c2;
                            if (!masked && !(SW && kq[j] == 0.f)) rowacc[pi] = fmaxf(rowacc[pi], x);
                            // mirrored tile: row p is a column of row q's soft-max (not when its k is 0 inside a modality)
                            if (SYM && !masked && !pad_row && !(SW && same_mod && kp[pi] == 0.f)) es[qi][r] = fmaxf(es[qi][r], x);
                        } else if (MODE == 3) {
                            const float h = g.m2 + acc[qi][pi][r] - myshift[pi];
                            const bool ha = !masked && h > 0.f;
                            if (ha) { rowacc[pi] += h; rowcnt[pi] += 1.f; }
                            if (SYM) {   // the same score seen from column q: against ITS positive pair
                                const float hc = g.m2 + acc[qi][pi][r] - shq[j];
                                const bool hca = !masked && !pad_row && hc > 0.f;
                                if (hca) { es[qi][r] += hc; ec[qi][r] += 1.f; }
                                // the pair's weight in the backward (crossclr_maxmargin_backward_saved): its number of active hinges, 0 .. 2
                                hinge_bytes |= (unsigned)((ha && !pad_row ? 1 : 0) + (hca ? 1 : 0)) << (8 * j);
                            }
                        } else if (MODE == 4) {
                            if (diag_tile && q_t == p_t) rowacc[pi] += acc[qi][pi][r];
                        } else {
                            const float x2 = acc[qi][pi][r] * c2;
                            float e = fast_exp2(x2 - (MODE == 2 ? myshift[pi] : g.m2));
                            if (masked) e = 0.f;
                            ev[j] = e;
                            float et = e;            // relative to the COLUMN's shift (single pass: one shift, symmetric matrix)
                            if (MODE == 2 && (kNeedColShift || ut_records)) { et = fast_exp2(x2 - shq[j]); if (masked) et = 0.f; }
                            etv[j] = et;
                            if (SYM) es[qi][r] += pad_row ? 0.f : ((SW && same_mod) ? et * kp[pi] : et);
                            if (SW) e *= kq[j];
                            rowacc[pi] += e;
                        }
                    }
                    if (MODE == 3 && SYM && hinge_mask)      // mask[video row][text column], bpad x bpad bytes: four columns of this lane's row
                        *reinterpret_cast<unsigned*>(hinge_mask + (size_t)(r_in_mod0 + p_t) * g.bpad + ct.in_mod0 + 64 * wc + 32 * qi + 8 * r4 + 4 * half) = hinge_bytes;
                    (void)hinge_bytes;
                    if constexpr (ST && sizeof(T) == 2) {
                        // bf16 plans beyond the register-resident kernels (Dpad > 1024): the record of tile (p32, q32), q32 >= 4 (p32 / 4) --
                        // [k-step th = r4 >> 1][lane][8 bf16], lane (p, half) holding E[p][16 th + 8 (r4 & 1) + 4 half + j] -- in the upper-
                        // triangle layout of the register-resident forward with 128-row blocks (stash_tile_index, tpr = 4): what
                        // fast_bwd_dsl_kernel<..., TPRF = 4> reads, the mirrored half through its transposing gather.  8 bytes per lane and r4.
                        // MODE 2 (two-pass soft-max, bf16 register-resident plans): U[p][q] = exp2(x - shift[p]) is not symmetric, so the FULL matrix
                        // is evaluated and every tile's record goes to the rectangular layout [row group][column tile] that
                        // fast_bwd_dsl_kernel<..., MODE 1 / 2> read (crossclr_backward_saved_s: W = U rz_p + U^T rz_q as two launches).
                        const int p32 = (row0 + 64 * wr + 32 * pi) >> 5, q32 = (int)((ct.row0 + 64 * wc + 32 * qi) >> 5);
                        if constexpr (MODE == 2) {
                            struct B4 { bf16_t e[4]; } pk;
#pragma unroll
                            for (int j = 0; j < 4; ++j) pk.e[j] = f32_to_bf16_bits(ev[j]);
                            // (tile index in launch order: the local block's own position; a rectangular launch over other ranks: its range)
                            const size_t q32l = (size_t)4 * t + 2 * wc + qi, NQl = (size_t)g.col_ranks * (size_t)(2 * g.bpad / 32);
                            unsigned char* rec = reinterpret_cast<unsigned char*>(stash) + ((size_t)p32 * NQl + q32l) * 2048 +
                                                 1024 * (r4 >> 1) + 16 * lane + 8 * (r4 & 1);
                            *reinterpret_cast<B4*>(rec) = pk;
                            if (ut_records) {   // Ut[p][q] = exp2(x - shift_all[q]) behind U: crossclr_backward_rect_saved_s weighs it with the columns' statistics
#pragma unroll
                                for (int j = 0; j < 4; ++j) pk.e[j] = f32_to_bf16_bits(etv[j]);
                                *reinterpret_cast<B4*>(rec + (size_t)(2 * g.bpad / 32) * NQl * 2048) = pk;
                            }
                        } else if constexpr (!SYM) {
                            // wide bf16 plans, this rank's rows against OTHER ranks' columns (crossclr_forward_rect_save): the rectangular layout
                            // [row group][tile of the launch's rank range, in launch order] that fast_bwd_dsl_kernel<..., MODE 1> reads
                            struct B4 { bf16_t e[4]; } pk;
#pragma unroll
                            for (int j = 0; j < 4; ++j) pk.e[j] = f32_to_bf16_bits(ev[j]);
                            const size_t q32l = (size_t)4 * t + 2 * wc + qi, NQl = (size_t)g.col_ranks * (size_t)(2 * g.bpad / 32);
                            unsigned char* rec = reinterpret_cast<unsigned char*>(stash) + ((size_t)p32 * NQl + q32l) * 2048 +
                                                 1024 * (r4 >> 1) + 16 * lane + 8 * (r4 & 1);
                            *reinterpret_cast<B4*>(rec) = pk;
                        } else if (q32 >= 4 * (p32 / 4)) {
                            struct B4 { bf16_t e[4]; } pk;
#pragma unroll
                            for (int j = 0; j < 4; ++j) pk.e[j] = f32_to_bf16_bits(ev[j]);
                            unsigned char* rec = reinterpret_cast<unsigned char*>(stash) + stash_tile_index(4, 2 * g.bpad / 32, p32, q32) * 2048 +
                                                 1024 * (r4 >> 1) + 16 * lane + 8 * (r4 & 1);
                            *reinterpret_cast<B4*>(rec) = pk;
                        }
                    } else if constexpr (ST) {
                        // fragment (p32, q32): q32 counts the 32-column fragments of THIS launch's column range (the local block: the stacked
                        // operand; a rectangular launch over other ranks' columns -- exact-fp32 sharded runs, crossclr_forward_rect_save --:
                        // its col_ranks segments in launch order), NQs fragments per row group
                        const size_t p32 = (size_t)(row0 + 64 * wr + 32 * pi) >> 5, q32 = (size_t)4 * t + 2 * wc + qi;
                        const size_t NQs = (size_t)g.col_ranks * (size_t)(2 * g.bpad / 32);
                        const size_t nn = (size_t)(2 * g.bpad) * (size_t)(2 * g.bpad) * (size_t)(SYM ? 1 : g.col_ranks);   // floats of one matrix (local block; rectangular: its rank range)
                        *reinterpret_cast<f32x4*>(stash + ((p32 * NQs + q32) << 10) + 256 * r4 + 4 * lane) = ev;
                        if (MODE == 2) *reinterpret_cast<f32x4*>(stash + nn + ((p32 * NQs + q32) << 10) + 256 * r4 + 4 * lane) = etv;
                        if (SYM && mirror) {   // fragment (q32, p32): lane' = column q, element of row p at [p >> 3][half' = (p >> 2) & 1][p & 3]
                            float* tf = stash + ((q32 * NQs + p32) << 10) + 256 * (l31 >> 3) + 128 * ((l31 >> 2) & 1) + (l31 & 3);
#pragma unroll
                            for (int j = 0; j < 4; ++j) tf[4 * (8 * r4 + 4 * half + j)] = pad_row ? 0.f : etv[j];      // U[q][p]
                            if (MODE == 2) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) tf[nn + 4 * (8 * r4 + 4 * half + j)] = pad_row ? 0.f : ev[j];   // Ut[q][p] = U[p][q]
                            }
                        }
                    }
                }
            }
        if (SYM) {   // column sums of the tile over the block's 128 rows -> colpart[I][128 t + c]  (block-uniform branch)
            if (mirror) {
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    const float cs = MODE == 1 ? halving_max16(es[qi], l31) : halving_sum16(es[qi], l31);
                    if (l31 < 16) red[wr * 128 + 64 * wc + 32 * qi + frag_row(halving_elem16(l31), half)] = cs;
                }
            }
            __syncthreads();
            if (mirror && tid < 128)
                colpart[(size_t)rbk * 2 * g.bpad + (size_t)t * 128 + tid] = MODE == 1 ? fmaxf(red[tid], red[128 + tid]) : red[tid] + red[128 + tid];
            __syncthreads();
            if (MODE == 3) {   // the active counts take the same route, into the second half of colpart
#pragma unroll
                for (int qi = 0; qi < 2; ++qi) {
                    const float cs = halving_sum16(ec[qi], l31);
                    if (l31 < 16) red[wr * 128 + 64 * wc + 32 * qi + frag_row(halving_elem16(l31), half)] = cs;
                }
                __syncthreads();
                if (tid < 128) colpart[(size_t)(g.bpad / 128 + rbk) * 2 * g.bpad + (size_t)t * 128 + tid] = red[tid] + red[128 + tid];
                __syncthreads();
            }
        }
    }
    // combine the two lane halves, then the two column waves
    if (MODE == 1) {
        rowacc[0] = fmaxf(rowacc[0], wave_xor_f32(rowacc[0], 32));
        rowacc[1] = fmaxf(rowacc[1], wave_xor_f32(rowacc[1], 32));
    } else {
        rowacc[0] += wave_xor_f32(rowacc[0], 32);
        rowacc[1] += wave_xor_f32(rowacc[1], 32);
    }
    if (half == 0) {
        red[wc * 128 + 64 * wr + l31] = rowacc[0];
        red[wc * 128 + 64 * wr + 32 + l31] = rowacc[1];
    }
    __syncthreads();
    if (tid < 128)
        part[(size_t)blockIdx.y * 2 * g.bpad + row0 + tid] = MODE == 1 ? fmaxf(red[tid], red[128 + tid]) : red[tid] + red[128 + tid];
    if (MODE == 3) {   // the counts take the same route into `stash`
        __syncthreads();
        rowcnt[0] += wave_xor_f32(rowcnt[0], 32);
        rowcnt[1] += wave_xor_f32(rowcnt[1], 32);
        if (half == 0) {
            red[wc * 128 + 64 * wr + l31] = rowcnt[0];
            red[wc * 128 + 64 * wr + 32 + l31] = rowcnt[1];
        }
        __syncthreads();
        if (tid < 128) stash[(size_t)blockIdx.y * 2 * g.bpad + row0 + tid] = red[tid] + red[128 + tid];
    }
    if (PAIRED) __syncthreads();    // (the next segment reuses `red`)
    }   // segments
}
// score statistics, second half: slot sums -> hinge[p], active[p] (a count, as float), and the block partial of sum_p hinge (double; added up
// in index order by fwd_finish_reduce_kernel: deterministic)
// colpart != NULL (one-pass evaluation): the stacked rows of modality 1 get their statistics from the column sums the row blocks
// of modality 0 left behind (hinges: rows 0 .. bpad/128 - 1 of colpart, counts: the next bpad/128 rows), in row-block order
__global__ void __launch_bounds__(256) score_finish_kernel(const float* part, const float* cnt, int nslots, int bpad, int b,
                                                           float* hinge, float* active, double* loss_ws, const float* colpart) {
    CROSSCLR_SHARED double sh[256];
    const int n = 2 * bpad;
    double acc = 0.0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        float h = 0.f, c = 0.f;
        if (colpart && p >= bpad) {
            const int nrb = bpad / 128;
            for (int k = 0; k < nrb; ++k) { h += colpart[(size_t)k * n + p]; c += colpart[(size_t)(nrb + k) * n + p]; }
        } else {
            for (int s = 0; s < nslots; ++s) { h += part[(size_t)s * n + p]; c += cnt[(size_t)s * n + p]; }
        }
        const bool valid = (p < bpad ? p : p - bpad) < b;
        hinge[p] = valid ? h : 0.f;
        active[p] = valid ? c : 0.f;
        if (valid) acc += (double)h;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_ws[1 + blockIdx.x] = sh[0];
}
// pass 1 of the two-pass soft-max, second half: shift[p] = max(slot maxima, own previous value if `accumulate`, and the masked
// self pair's logit 0 when it is part of the row's soft-max: k_p != 0)
// colpart != NULL: the launch evaluated the upper triangle only (SYM): the column maxima the row blocks above left behind count too
__global__ void __launch_bounds__(256) rowmax_combine_kernel(const float* part, int nslots, int n, const float* krows, int accumulate,
                                                             float* shift, const float* colpart) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    float m = accumulate ? shift[p] : ((krows && krows[p] == 0.f) ? -3.0e38f : 0.f);
    for (int k = 0; k < nslots; ++k) m = fmaxf(m, part[(size_t)k * n + p]);
    if (colpart) {
        // (a column with k_p = 0 inside its modality is not in anybody's soft-max, but as a ROW it still has one: its own
        //  maxima came from the slots; the mirrored maxima were formed with the row-side rule of the kernel)
        for (int k = 0; k < p / 128; ++k) m = fmaxf(m, colpart[(size_t)k * n + p]);
    }
    shift[p] = m;
}

// ---------------------------------------------------------------------------------------------
// K3: forward finish: slots -> logZ, 1/Z, w/Z, and the loss sum (double).
//   log Z = shift + ln(sum_slots + exp(-shift));  the "+exp(-shift)" is the masked intra-modal
//   diagonal whose logit the reference sets to 0.0 (trainer/loss.py:96-97), i.e. exp(0) = 1.
//   Two launches: rows in parallel (each block leaves its partial loss sum in loss_ws[1 + block]),
//   then one wave adds the block partials in index order -> deterministic, no atomics, no memset.
// ---------------------------------------------------------------------------------------------
// header[4*L .. 4*L+3] of launch L (written by the forward launch itself): {kind, tpr, NT, per}
// (tpr = 32-column tiles = 32-row groups per row block: row block I = p / (32*tpr)).
//   kind 0: dense -- header[1] = 0: all `slots_per_launch` slots of the launch are valid for every row;
//           header[1] = c > 0: the first c slots; header[1] < 0: none (an unused launch group)
//   kind 1/2/3: persistent fast forward (symmetric / rectangular / pairs): row block I owns slots
//             0 .. last_block(I) - first_block(I) (fwdw_*: crossclr_device.h, per = cost units per thread block);
//             kind 1 additionally has column sums colpart[I' < I][p]
__global__ void __launch_bounds__(256) fwd_finish_kernel(const float* part, int nlaunch, int slots_per_launch, Geo g,
                                                         const float* diag_cos, float inv_tau, float neg_w, float* logz,
                                                         float* rz, float* wrz, double* loss_ws, const float* colpart,
                                                         const int* header, const float* krows, const float* lw,
                                                         const float* row_shift, int* ticket, double scale) {
    CROSSCLR_SHARED double red[4];
    CROSSCLR_SHARED int last_block;
    const int n = 2 * g.bpad;
    double acc = 0.0;
    // LPR lanes per row: lane q of the group adds the terms k = q, q + LPR, ... (independent loads, fixed order), the group combines by a
    // butterfly.  4 lanes: 10 us at B = 8192; 16 lanes (a quarter of the loads per lane, four times the blocks and the fp64 logs): 13-15 us
    // (profiles/r05i_ab_finish_lpr.txt) -- the kernel is bound by its chain of dependent round trips (header -> sums -> log -> block sum),
    // not by the loads per lane.
#ifndef CROSSCLR_FINISH_LPR
#define CROSSCLR_FINISH_LPR 4
#endif
    constexpr int LPR = CROSSCLR_FINISH_LPR;
    const int q = threadIdx.x & (LPR - 1);
    for (int p = (blockIdx.x * 256 + threadIdx.x) / LPR; p < n; p += gridDim.x * (256 / LPR)) {
        const int mod = p / g.bpad, i = p - mod * g.bpad;
        // shift of this row's sums (natural log): one value for the whole launch, or the row's own maximum (two-pass mode)
        const double shift = (row_shift ? (double)row_shift[p] : (double)g.m2) * (double)kLn2;
        const double self_term = exp(-shift);
        double s = 0.0;
        for (int L = 0; L < nlaunch; ++L) {
            const int kind = header[4 * L], tpr = header[4 * L + 1], NT = header[4 * L + 2], per = header[4 * L + 3];
            const float* base = part + (size_t)L * slots_per_launch * n;
            int count = slots_per_launch;
            if (kind == 0 && tpr != 0) count = tpr > 0 ? tpr : 0;
            if (kind == 4) {          // generic symmetric launch: every slot is valid; column sums of the row blocks above
                const int rb = p / (32 * tpr);
#pragma unroll 4
                for (int k = q; k < rb; k += LPR) s += (double)colpart[(size_t)k * n + p];
            } else if (kind != 0) {
                const int rb = p / (32 * tpr);
                count = fwdw_last_block(kind, tpr, NT, per, rb) - fwdw_first_block(kind, tpr, NT, per, rb) + 1;
                if (kind == 1) {
#pragma unroll 4
                    for (int k = q; k < rb; k += LPR) s += (double)colpart[(size_t)k * n + p];   // independent loads, fixed order
                }
            }
            for (int k = q; k < count; k += LPR) s += (double)base[(size_t)k * n + p];
        }
        s += wave_xor_f64(s, 1);
        s += wave_xor_f64(s, 2);
        if (LPR > 4) { s += wave_xor_f64(s, 4); s += wave_xor_f64(s, 8); }
        s += krows ? self_term * (double)krows[p] : self_term;   // the masked self pair travels with its column
        const bool valid = i < g.b;
        const double lz = shift + log(s);
        const double om = lw ? (double)lw[p] : 1.0;
        const float r = valid ? (float)(om / s) : 0.f;
        if (q == 0) {
            logz[p] = valid ? (float)lz : 0.f;
            rz[p] = r;
            wrz[p] = neg_w * r;
            if (valid) {
                acc += om * lz;
                if (mod == 0) acc -= (lw ? om + (double)lw[g.bpad + i] : 2.0) * (double)diag_cos[i] * (double)inv_tau;
            }
        }
    }
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    const double block_sum = (red[0] + red[1]) + (red[2] + red[3]);
    if (ticket == nullptr) {            // (crossclr_forward_finish*: fwd_finish_reduce_kernel adds the block partials up)
        if (threadIdx.x == 0) loss_ws[1 + blockIdx.x] = block_sum;
        return;
    }
    // crossclr_step_forward: the block that arrives LAST adds the partials up itself -- the same 64 lanes, strides and shuffle order as
    // fwd_finish_reduce_kernel, hence the same bits -- and clears the ticket for whoever uses this workspace next.  The partials travel
    // as written-through stores, complete before the block's ticket, and are read past the L1 (crossclr_device.h, handoff_*): no
    // agent-scope fence (its release writes back every line of the statistics the launch has just written to the XCD's L2):
    // -2 us per step, profiles/r05n_ab_rowkernels.txt; tests/test_gpu_step_handoff.py alternates two batches on one workspace.
    if (threadIdx.x == 0) {
        handoff_store_f64(&loss_ws[1 + blockIdx.x], block_sum);
        handoff_stores_complete();
        last_block = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last_block || threadIdx.x >= 64) return;
    const int nblocks = (int)gridDim.x;
    double tot = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 64) tot += handoff_load_f64(&loss_ws[1 + k]);
    tot = wave_sum_f64(tot);
    if (threadIdx.x == 0) { loss_ws[0] = tot; if (nblocks >= 1) loss_ws[1] = tot * scale; *ticket = 0; }
}
// pairs exchange (sharded forward): out[c] = sum over row blocks of the column sums a kind-3 launch left in colpart
__global__ void __launch_bounds__(256) colsum_reduce_kernel(const float* colpart, int nrb, int ncols, float* out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols) return;
    float s = 0.f;
#pragma unroll 8
    for (int rb = 0; rb < nrb; ++rb) s += colpart[(size_t)rb * ncols + c];
    out[c] = s;
}
// a launch group made of ONE ready-made slot (partial sums received from other ranks), or an empty group
__global__ void __launch_bounds__(256) fwd_add_kernel(const float* vec, int n, float* slot, int* header) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { header[0] = 0; header[1] = vec ? 1 : -1; header[2] = 0; header[3] = 0; }
    if (vec && i < n) slot[i] = vec[i];
}
// loss_ws[0] = the sum; the block partials are dead after this read, so loss_ws[1] receives sum * scale: the rank's
// contribution to the mean loss, ready to be returned without another kernel (scale = 1 / (2 B_global))
__global__ void __launch_bounds__(64) fwd_finish_reduce_kernel(double* loss_ws, int nblocks, double scale) {
    double acc = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 64) acc += loss_ws[1 + k];
    acc = wave_sum_f64(acc);   // (every lane has read its partials before lane 0 overwrites loss_ws[1])
    if (threadIdx.x == 0) { loss_ws[0] = acc; if (nblocks >= 1) loss_ws[1] = acc * scale; }
}

// ---------------------------------------------------------------------------------------------
// K2: backward, generic tiled version.  For a block of 64 rows P and an output slice of DC
// embedding columns it walks every 64-column tile Q:
//   phase A  S^T[q][p] = xhat_Q . xhat_P          (MFMA, contraction over D, operands via LDS)
//   phase B  W[p][q]   = s * exp2(c S - m2) * (rz_p + rz_q), self pair -> 0, to LDS in operand type
//   phase C  G[p][d]  += W[p][:] . xhat_Q[:, d]   (MFMA, contraction over the 64 columns q;
//            the q-contiguous operand comes from ds_read_b64_tr_b16 for bf16, plain ds_read_b32 for fp32)
// grid = (2*bpad/64, Dpad/DC, column slices).  Accumulators: 64 x DC fp32 per block (64 VGPRs per lane at DC=256).
// Phase A is repeated by every D slice: 8 + 16 Dpad/DC flop per pair and embedding column, so DC = 512 (the whole row at
// Dpad = 512: 16 B^2 D executed instead of 24; fp32: the column slice alone fills 128 KiB of the CU's 160 KiB of LDS) where Dpad allows.
// ---------------------------------------------------------------------------------------------
template <typename T, int DC> struct BwdLds {
    static constexpr int kTileP = 0;
    static constexpr int kTileQ = 64 * 128;
    static constexpr int kW = 2 * 64 * 128;                         // W tile  [64][64] of T
    static constexpr int kWBytes = 64 * 64 * (int)sizeof(T);
    static constexpr int kXQ = kW + kWBytes;                        // column slice [64][DC] of T
    static constexpr int kXQBytes = 64 * DC * (int)sizeof(T);
    static constexpr int kTotal = kXQ + kXQBytes;
};

// byte offset of element (q, d) inside the [64][DC] column slice
template <typename T, int DC> __device__ __forceinline__ int xq_off(int q, int d);
template <> __device__ __forceinline__ int xq_off<float, 64>(int q, int d) { return (q * 64 + d) * 4; }
template <> __device__ __forceinline__ int xq_off<float, 128>(int q, int d) { return (q * 128 + d) * 4; }
template <> __device__ __forceinline__ int xq_off<float, 256>(int q, int d) { return (q * 256 + d) * 4; }
template <> __device__ __forceinline__ int xq_off<float, 512>(int q, int d) { return (q * 512 + d) * 4; }
// bf16: 16-byte chunks XOR-swizzled by (q&3)<<2 so the four rows a transpose-read touches land in
// four different 64-byte bank groups (needs >= 16 chunks per row, i.e. DC >= 128)
template <int DC> __device__ __forceinline__ int xq_off_bf16(int q, int d) {
    int chunk = d >> 3, inner = (d & 7) * 2;
    if (DC >= 128) chunk ^= (q & 3) << 2;
    return q * DC * 2 + chunk * 16 + inner;
}
template <> __device__ __forceinline__ int xq_off<bf16_t, 64>(int q, int d) { return xq_off_bf16<64>(q, d); }
template <> __device__ __forceinline__ int xq_off<bf16_t, 128>(int q, int d) { return xq_off_bf16<128>(q, d); }
template <> __device__ __forceinline__ int xq_off<bf16_t, 256>(int q, int d) { return xq_off_bf16<256>(q, d); }
template <> __device__ __forceinline__ int xq_off<bf16_t, 512>(int q, int d) { return xq_off_bf16<512>(q, d); }

// phase B store of 4 consecutive-q weights of row p (q0 multiple of 4)
__device__ __forceinline__ void w_store4(unsigned char* wt, int p, int q0, f32x4 w, float*) {
    // fp32 W tile = two K-tiles of 32 columns each
    unsigned char* t = wt + (q0 >> 5) * (64 * 128);
    *reinterpret_cast<f32x4*>(t + ktile_off(p, (q0 & 31) >> 2)) = w;
}
__device__ __forceinline__ void w_store4(unsigned char* wt, int p, int q0, f32x4 w, bf16_t*) {
    u32x2 pk;
    pk[0] = (uint32_t)f32_to_bf16_bits(w[0]) | ((uint32_t)f32_to_bf16_bits(w[1]) << 16);
    pk[1] = (uint32_t)f32_to_bf16_bits(w[2]) | ((uint32_t)f32_to_bf16_bits(w[3]) << 16);
    *reinterpret_cast<u32x2*>(wt + ktile_off(p, q0 >> 3) + (q0 & 4) * 2) = pk;
}

// phase C inner products for one wave: rows 32*wr.., DC/2 embedding columns starting at dw0
template <int DC>
__device__ __forceinline__ void bwd_gemm2(const unsigned char* wt, const unsigned char* xq, int wr, int dw0,
                                          int lane, f32x16 (&acc)[DC / 64], float*) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {  // 8 columns q per step: k-slot `half` covers q = 8kk+4half+j
        const unsigned char* t = wt + (kk >> 2) * (64 * 128);
        f32x4 a = *reinterpret_cast<const f32x4*>(t + ktile_off(32 * wr + l31, 2 * (kk & 3) + half));
#pragma unroll
        for (int dt = 0; dt < DC / 64; ++dt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float b = *reinterpret_cast<const float*>(xq + xq_off<float, DC>(8 * kk + 4 * half + j, dw0 + 32 * dt + l31));
                acc[dt] = mfma_32x32x2_f32(a[j], b, acc[dt]);
            }
        }
    }
}
template <int DC>
__device__ __forceinline__ void bwd_gemm2(const unsigned char* wt, const unsigned char* xq, int wr, int dw0,
                                          int lane, f32x16 (&acc)[DC / 64], bf16_t*) {
    const int half = lane >> 5, l31 = lane & 31;
    // transpose-read roles inside a 16-lane group: lane 4j+c addresses row j, 8-byte piece c
    const int grp = lane >> 4, i16 = lane & 15, jrow = i16 >> 2, piece = i16 & 3;
    const int dsub = grp & 1;  // which 16-column half of the 32-wide fragment this group delivers
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // 16 columns q per step: k-slot (half, j8) <-> q = 16ks+8half+j8
        bf16x8 a = *reinterpret_cast<const bf16x8*>(wt + ktile_off(32 * wr + l31, 2 * ks + half));
#pragma unroll
        for (int dt = 0; dt < DC / 64; ++dt) {
            const int dcol = dw0 + 32 * dt + 16 * dsub + 4 * piece;
            const int q0 = 16 * ks + 8 * half + jrow;
            s16x4 lo = lds_read_tr16_b64(xq + xq_off<bf16_t, DC>(q0, dcol));
            s16x4 hi = lds_read_tr16_b64(xq + xq_off<bf16_t, DC>(q0 + 4, dcol));
            struct { s16x4 lo, hi; } pair = {lo, hi};  // k-slots 0..3 <- rows q0.., 4..7 <- rows q0+4..
            const bf16x8 bfr = __builtin_bit_cast(bf16x8, pair);
            acc[dt] = mfma_32x32x16_bf16(a, bfr, acc[dt]);
        }
    }
}

// SW (sample weights): the intra-modal weight is s E (wrz_p k_q + wrz_q k_p) instead of s E (wrz_p + wrz_q).
// RM (two-pass soft-max, small temperatures): rz = omega / (row sum relative to the ROW's shift), so the weight is
//    s (exp2(x - shift_p) wrz_p k_q + exp2(x - shift_q) wrz_q k_p)  -- two exponentials, both <= 1: nothing can overflow.
// LOSS = 1: the max-margin ranking loss (trainer/loss.py:29-41) on the same skeleton -- no soft-max: the weight of an inter-modal
//    pair is the number of its active hinges, [g.m2 + S - d_p > 0] + [g.m2 + S - d_q > 0] (d = the positive pairs' cosines, passed
//    in shift_rows / shift_cols; g.m2 = margin), 0 for the positive pair itself and inside a modality (those tiles are skipped).
template <typename T, int DC, bool SW, bool RM, int LOSS = 0>
__global__ void __launch_bounds__(256) bwd_kernel(const T* rows, const T* cols, Geo g, const float* rz_rows,
                                                  const float* wrz_rows, const float* rz_cols, const float* wrz_cols,
                                                  float* gbuf, int accumulate, int tiles_per_slice,
                                                  const float* krows, const float* kcols,
                                                  const float* shift_rows, const float* shift_cols) {
    typedef Operand<T> Op;
    typedef BwdLds<T, DC> L;
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[L::kTotal];
    unsigned char* tileP = lds + L::kTileP;
    unsigned char* tileQ = lds + L::kTileQ;
    unsigned char* wt = lds + L::kW;
    unsigned char* xq = lds + L::kXQ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const size_t pitch = (size_t)g.Dpad * sizeof(T);
    const int nchunks = g.Dpad / Op::kChunkElems;

    const int row0 = blockIdx.x * 64;
    const int rmod = row0 / g.bpad;
    const int r_in_mod0 = row0 - rmod * g.bpad;
    const int d0 = blockIdx.y * DC;
    const unsigned char* rbase = reinterpret_cast<const unsigned char*>(rows) + (size_t)row0 * pitch;

    // phase A/B roles: wave computes S^T for columns 32*wq.., rows 32*wp..
    const int wq = wave & 1, wp = wave >> 1;
    // phase C roles: rows 32*wr.., embedding columns wc*(DC/2)..
    const int wr = wave & 1, wc = wave >> 1;

    f32x16 acc2[DC / 64];
#pragma unroll
    for (int dt = 0; dt < DC / 64; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[dt][r] = 0.f;

    const int p_t = 32 * wp + l31;  // this lane's row inside the block (phase B)
    const float rzp_inter = rz_rows[row0 + p_t];
    const float rzp_intra = wrz_rows[row0 + p_t];
    const float kp = SW ? krows[row0 + p_t] : 1.f;
    const float shp = (RM || LOSS == 1) ? shift_rows[row0 + p_t] : 0.f;

    // column slice z walks its share of the USABLE tiles (the skipped rank's segment is cut out of the numbering,
    // so the slices stay balanced) ...
    const int per_rank = 2 * g.bpad / 64;
    const int skip_seg = (g.skip_rank >= g.col_rank0 && g.skip_rank < g.col_rank0 + g.col_ranks) ? g.skip_rank - g.col_rank0 : -1;
    const int usable = (g.col_ranks - (skip_seg >= 0 ? 1 : 0)) * per_rank;
    const int t_begin = blockIdx.z * tiles_per_slice;
    int t_stop = t_begin + tiles_per_slice;
    if (t_stop > usable) t_stop = usable;
    KTileStage<64, 256> sp, sq;
    for (int u = t_begin; u < t_stop; ++u) {
        const ColTile ct = col_tile(g, (skip_seg >= 0 && u >= skip_seg * per_rank) ? u + per_rank : u, 64);
        if (LOSS == 1 && ct.mod == rmod) continue;   // (uniform for the block: no barrier is skipped by part of it)
        const unsigned char* cbase = reinterpret_cast<const unsigned char*>(cols) + ct.row0 * pitch;
        // ---------------- phase A ----------------
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        sp.fetch(rbase, pitch, 0, tid);
        sq.fetch(cbase, pitch, 0, tid);
        for (int kc = 0; kc < nchunks; ++kc) {
            sp.commit(tileP, tid);
            sq.commit(tileQ, tid);
            __syncthreads();  // also: every wave is past phase C of the previous tile
            if (kc == 0) {
                // column slice for phase C: [64][DC] elements, 16-byte pieces
                constexpr int kPiecesPerRow = DC * (int)sizeof(T) / 16;
                constexpr int kElemsPerPiece = 16 / (int)sizeof(T);
                for (int id = tid; id < 64 * kPiecesPerRow; id += 256) {
                    const int q = id / kPiecesPerRow, c = id - q * kPiecesPerRow;
                    u32x4 v = *reinterpret_cast<const u32x4*>(cbase + (size_t)q * pitch +
                                                             ((size_t)d0 + c * kElemsPerPiece) * sizeof(T));
                    *reinterpret_cast<u32x4*>(xq + xq_off<T, DC>(q, c * kElemsPerPiece)) = v;
                }
            }
            if (kc + 1 < nchunks) {
                sp.fetch(rbase, pitch, (kc + 1) * 128, tid);
                sq.fetch(cbase, pitch, (kc + 1) * 128, tid);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                typename Op::frag a = Op::load(tileQ, 32 * wq + l31, s, half);
                typename Op::frag bfr = Op::load(tileP, 32 * wp + l31, s, half);
                acc = Op::mma(a, bfr, acc);
            }
            __syncthreads();
        }
        // ---------------- phase B ----------------
        const bool same_mod = (ct.mod == rmod);
        const float c2 = same_mod ? g.c_intra : g.c_inter;
        const float rzp = same_mod ? rzp_intra : rzp_inter;
        const float* rzq = (same_mod ? wrz_cols : rz_cols) + ct.stat0;
        const bool diag_tile = (LOSS == 1 ? !same_mod : same_mod) && ct.rank == g.row_rank && ct.in_mod0 == r_in_mod0;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int q0 = 32 * wq + 8 * r4 + 4 * half;  // frag_row(4*r4 + j, half) = q0 - 32wq + j
            const f32x4 rq = *reinterpret_cast<const f32x4*>(rzq + q0);
            f32x4 kq = {1.f, 1.f, 1.f, 1.f};
            if (SW && same_mod) kq = *reinterpret_cast<const f32x4*>(kcols + ct.stat0 + q0);
            const float kpe = (SW && same_mod) ? kp : 1.f;
            f32x4 shq = {0.f, 0.f, 0.f, 0.f};
            if (RM || LOSS == 1) shq = *reinterpret_cast<const f32x4*>(shift_cols + ct.stat0 + q0);
            f32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v;
                if (LOSS == 1) {
                    const float x = g.m2 + acc[4 * r4 + j];
                    v = (x - shp > 0.f ? 1.f : 0.f) + (x - shq[j] > 0.f ? 1.f : 0.f);
                    if (ct.in_mod0 + q0 + j >= g.b) v = 0.f;    // (padding columns: their operand rows are zero anyway)
                } else if (RM) {
                    const float x = acc[4 * r4 + j] * c2;
                    const float ep = fast_exp2(x - shp), eq = fast_exp2(x - shq[j]);
                    v = SW ? (ep * rzp * kq[j] + eq * rq[j] * kpe) : (ep * rzp + eq * rq[j]);
                } else {
                    const float e = fast_exp2(acc[4 * r4 + j] * c2 - g.m2);
                    v = SW ? e * (rzp * kq[j] + rq[j] * kpe) : e * (rzp + rq[j]);
                }
                if (diag_tile && q0 + j == p_t) v = 0.f;
                w[j] = v;
            }
            w_store4(wt, p_t, q0, w, (T*)nullptr);
        }
        __syncthreads();
        // ---------------- phase C ----------------
        bwd_gemm2<DC>(wt, xq, wr, wc * (DC / 2), lane, acc2, (T*)nullptr);
    }
    // G[row][d]: lane holds column d = l31 of each fragment, 16 rows ... into its own slice
    float* gslice = gbuf + (size_t)blockIdx.z * 2 * g.bpad * g.Dpad;
    if (accumulate) {
#pragma unroll
        for (int dt = 0; dt < DC / 64; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                gslice[(size_t)(row0 + 32 * wr + frag_row(r, half)) * g.Dpad + d0 + wc * (DC / 2) + 32 * dt + l31] += acc2[dt][r];
    } else {
#pragma unroll
        for (int dt = 0; dt < DC / 64; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                gslice[(size_t)(row0 + 32 * wr + frag_row(r, half)) * g.Dpad + d0 + wc * (DC / 2) + 32 * dt + l31] = acc2[dt][r];
    }
}

// Max-margin ranking loss (trainer/loss.py:29-41), backward from the SAVED hinge mask (crossclr_score_rows_save): the weight of a pair is
// the number of its active hinges, which the forward has already decided -- no similarity product here, phase C of bwd_kernel alone:
//   G[p][d] += sum_q mask(p, q) x_q[d]      over the 64-column tiles q of the OTHER modality in this block's column slice.
// mask[video row][text column] (bpad x bpad bytes): video rows read their tile as stored, text rows the transposed one (through LDS).
// grid = (2*bpad/64, Dpad/DC, column slices); same slices / same output layout as bwd_kernel<..., LOSS = 1>.
template <typename T, int DC>
__global__ void __launch_bounds__(256) maxmargin_saved_kernel(const T* X, Geo g, const unsigned char* mask, float* gbuf, int tiles_per_slice) {
    typedef BwdLds<T, DC> L;
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[L::kTotal];
    unsigned char* mt = lds;                 // the 64 x 64 byte tile of the mask (the K-tile area of bwd_kernel: no phase A here)
    unsigned char* wt = lds + L::kW;
    unsigned char* xq = lds + L::kXQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const size_t pitch = (size_t)g.Dpad * sizeof(T);
    const int row0 = blockIdx.x * 64;
    const int rmod = row0 / g.bpad;
    const int r_in_mod0 = row0 - rmod * g.bpad;
    const int d0 = blockIdx.y * DC;
    const int wq = wave & 1, wp = wave >> 1;      // weight-tile roles: columns 32*wq.., rows 32*wp..
    const int wr = wave & 1, wc = wave >> 1;      // product roles: rows 32*wr.., embedding columns wc*(DC/2)..
    f32x16 acc2[DC / 64];
#pragma unroll
    for (int dt = 0; dt < DC / 64; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[dt][r] = 0.f;
    const int p_t = 32 * wp + l31;
    const int ntiles = 2 * g.bpad / 64;
    const int t_begin = blockIdx.z * tiles_per_slice;
    int t_stop = t_begin + tiles_per_slice;
    if (t_stop > ntiles) t_stop = ntiles;
    for (int u = t_begin; u < t_stop; ++u) {
        const ColTile ct = col_tile(g, u, 64);
        if (ct.mod == rmod) continue;      // (uniform for the block)
        const unsigned char* cbase = reinterpret_cast<const unsigned char*>(X) + ct.row0 * pitch;
        __syncthreads();                   // every wave is past the product of the previous tile
        {
            constexpr int kPiecesPerRow = DC * (int)sizeof(T) / 16;
            constexpr int kElemsPerPiece = 16 / (int)sizeof(T);
            for (int id = tid; id < 64 * kPiecesPerRow; id += 256) {
                const int q = id / kPiecesPerRow, c = id - q * kPiecesPerRow;
                *reinterpret_cast<u32x4*>(xq + xq_off<T, DC>(q, c * kElemsPerPiece)) =
                    *reinterpret_cast<const u32x4*>(cbase + (size_t)q * pitch + ((size_t)d0 + c * kElemsPerPiece) * sizeof(T));
            }
            // the mask tile, 64 rows of 64 bytes as stored: (video rows of the block, text columns) or (video columns, text rows of the block)
            const int mr = tid >> 2, mc = (tid & 3) * 16;
            const size_t vrow = rmod == 0 ? (size_t)(r_in_mod0 + mr) : (size_t)(ct.in_mod0 + mr);
            const size_t tcol = rmod == 0 ? (size_t)ct.in_mod0 : (size_t)r_in_mod0;
            *reinterpret_cast<u32x4*>(mt + mr * 64 + mc) = *reinterpret_cast<const u32x4*>(mask + vrow * g.bpad + tcol + mc);
        }
        __syncthreads();
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int q0 = 32 * wq + 8 * r4 + 4 * half;
            f32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (float)(rmod == 0 ? mt[p_t * 64 + q0 + j] : mt[(q0 + j) * 64 + p_t]);
            w_store4(wt, p_t, q0, w, (T*)nullptr);
        }
        __syncthreads();
        bwd_gemm2<DC>(wt, xq, wr, wc * (DC / 2), lane, acc2, (T*)nullptr);
    }
    float* gslice = gbuf + (size_t)blockIdx.z * 2 * g.bpad * g.Dpad;
#pragma unroll
    for (int dt = 0; dt < DC / 64; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            gslice[(size_t)(row0 + 32 * wr + frag_row(r, half)) * g.Dpad + d0 + wc * (DC / 2) + 32 * dt + l31] = acc2[dt][r];
}

// ---------------------------------------------------------------------------------------------
// K4: backward finish (autograd of loss.py:79-80 + analytic positive-pair term), one wave per row
//   ghat = sum_slices(gbuf)/(2 B tau) - partner_hat/(B tau);  gx = (ghat - xhat (xhat.ghat)) * inv_norm * grad_out
// HBM-bound: slices + both input rows read once (kept in registers between the dot product and the
// output pass for D <= 1024), gradient row written once, 16-byte accesses where alignment allows.
// ---------------------------------------------------------------------------------------------
template <typename TIN>
__global__ void __launch_bounds__(256) bwd_finish_kernel(const float* gbuf, int nslices, const TIN* video, const TIN* text, long ldv,
                                                         long ldt, Geo g, const float* inv_norm, float inv_tau,
                                                         int Bglobal, const double* grad_out, TIN* gvideo,
                                                         TIN* gtext, long ldgv, long ldgt, const float* lw, int prenormalized) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + wave;  // 0 .. 2*b-1
    if (idx >= 2 * g.b) return;
    const int mod = idx / g.b, i = idx - mod * g.b;
    const TIN* own = (mod == 0 ? video + (size_t)i * ldv : text + (size_t)i * ldt);
    const TIN* oth = (mod == 0 ? text + (size_t)i * ldt : video + (size_t)i * ldv);
    TIN* out = (mod == 0 ? gvideo + (size_t)i * ldgv : gtext + (size_t)i * ldgt);
    // prenormalized == 2: the rows given ARE the unit vectors x^ = y / ||y|| (the packed operand of a fused projection) and inv_norm holds
    // 1 / ||y||: the gradient w.r.t. y is wanted -- same formula with the rows taken as they are (no second division by the norm)
    const bool unit_rows = prenormalized == 2;
    const double io = (double)inv_norm[mod * g.bpad + i];
    const double ip = unit_rows ? 1.0 : (double)inv_norm[(1 - mod) * g.bpad + i];
    const double ix = unit_rows ? 1.0 : io;      // row as given -> unit row
    const float* grow = gbuf + ((size_t)mod * g.bpad + i) * g.Dpad;
    const size_t slice = (size_t)2 * g.bpad * g.Dpad;
    const double sc = (double)inv_tau / (2.0 * (double)Bglobal);
    // positive pair: -(omega_v,i + omega_t,i)/(2 B tau) * partner  (= -1/(B tau) without sample weights)
    const double pc = (double)inv_tau / (double)Bglobal * (lw ? 0.5 * ((double)lw[i] + (double)lw[g.bpad + i]) : 1.0);
    // ||x|| < eps: x/eps, no projection term (inv_norm is stored as float: (float)1e12 = 999999995904); prenormalized rows:
    // the gradient is the one w.r.t. the unit vectors as given (inv_norm = 1, no projection)
    const bool clamped = io >= 9.99e11 || prenormalized == 1;
    const double go = grad_out[0];
    if (g.D <= 256 * kRowCache) {
        double gh[kRowCache][4], xh[kRowCache][4];
        double dot = 0.0;
#pragma unroll
        for (int k = 0; k < kRowCache; ++k) {
            const int d = 4 * lane + 256 * k;
            if (d < g.D) {
                f32x4 sum = *reinterpret_cast<const f32x4*>(grow + d);  // Dpad is a multiple of 64: aligned, in range
                for (int sl = 1; sl < nslices; ++sl) sum += *reinterpret_cast<const f32x4*>(grow + sl * slice + d);
                double o[4], x[4];
                row_load4(oth, d, g.D, o);
                row_load4(own, d, g.D, x);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gh[k][j] = (d + j < g.D) ? ((double)sum[j] * sc - o[j] * ip * pc) : 0.0;
                    xh[k][j] = x[j] * ix;
                    dot += xh[k][j] * gh[k][j];
                }
            }
        }
        dot = wave_sum_f64(dot);
#pragma unroll
        for (int k = 0; k < kRowCache; ++k) {
            const int d = 4 * lane + 256 * k;
            if (d < g.D) {
                double v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (clamped ? gh[k][j] : (gh[k][j] - xh[k][j] * dot)) * io * go;
                row_store4(out, d, g.D, v);
            }
        }
        return;
    }
    // wider rows (D > 256 * kRowCache: the wide bf16 plans): two passes over the row -- the dot product, then the output -- four elements per
    // lane and access (16-byte loads of the gradient slices, row_load4 / row_store4 on the rows: the scalar form this replaces ran at 2.2 TB/s)
    // (the slice sums of the first pass stay in registers -- 64 floats per lane cover D <= 4096 -- so the gradient slices are read once; only
    //  the two raw rows are read again, out of L2)
    constexpr int kWide = 16;                       // stretches of 256 elements whose slice sums are kept
    f32x4 csum[kWide];
    auto gh4 = [&](int d, const f32x4& sum, double (&gh)[4], double (&xh)[4]) {
        double o[4], x[4];
        row_load4(oth, d, g.D, o);
        row_load4(own, d, g.D, x);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gh[j] = (d + j < g.D) ? ((double)sum[j] * sc - o[j] * ip * pc) : 0.0;
            xh[j] = x[j] * ix;
        }
    };
    auto slices4 = [&](int d) {
        f32x4 sum = *reinterpret_cast<const f32x4*>(grow + d);          // Dpad is a multiple of 64: aligned, in range
        for (int sl = 1; sl < nslices; ++sl) sum += *reinterpret_cast<const f32x4*>(grow + sl * slice + d);     // fixed order
        return sum;
    };
    double dot = 0.0;
#pragma unroll
    for (int k = 0; k < kWide; ++k) {
        const int d = 4 * lane + 256 * k;
        if (d < g.D) {
            csum[k] = slices4(d);
            double gh[4], xh[4];
            gh4(d, csum[k], gh, xh);
#pragma unroll
            for (int j = 0; j < 4; ++j) dot += xh[j] * gh[j];
        }
    }
    for (int d = 4 * lane + 256 * kWide; d < g.D; d += 256) {           // (beyond 4096 columns: nothing kept)
        double gh[4], xh[4];
        gh4(d, slices4(d), gh, xh);
#pragma unroll
        for (int j = 0; j < 4; ++j) dot += xh[j] * gh[j];
    }
    dot = wave_sum_f64(dot);
#pragma unroll
    for (int k = 0; k < kWide; ++k) {
        const int d = 4 * lane + 256 * k;
        if (d < g.D) {
            double gh[4], xh[4], v[4];
            gh4(d, csum[k], gh, xh);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (clamped ? gh[j] : (gh[j] - xh[j] * dot)) * io * go;
            row_store4(out, d, g.D, v);
        }
    }
    for (int d = 4 * lane + 256 * kWide; d < g.D; d += 256) {
        double gh[4], xh[4], v[4];
        gh4(d, slices4(d), gh, xh);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (clamped ? gh[j] : (gh[j] - xh[j] * dot)) * io * go;
        row_store4(out, d, g.D, v);
    }
}

// Rows of at most 256 * kRowCache elements: ONE wave finishes the video row i AND the text row i (each is the other's positive
// pair), so both raw rows are read from HBM once instead of twice (as own row and as partner): 134 MB instead of 168 MB per
// launch at B = 8192, D = 512.  Same arithmetic, term by term, as bwd_finish_kernel above.
// KC = stretches of 256 elements a lane caches = ceil(D / 256) <= kRowCache: one instantiation per count, so that a D = 512 launch
// carries the registers of two stretches, not four (169 -> NN VGPRs: more rows in flight per CU for a kernel that only streams).
template <typename TIN, int KC>
__global__ void __launch_bounds__(256) bwd_finish_pair_kernel(const float* gbuf, int nslices, const TIN* video, const TIN* text, long ldv,
                                                              long ldt, Geo g, const float* inv_norm, float inv_tau,
                                                              int Bglobal, const double* grad_out, TIN* gvideo,
                                                              TIN* gtext, long ldgv, long ldgt, const float* lw, int prenormalized) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;  // 0 .. b-1
    if (i >= g.b) return;
    const TIN* pv = video + (size_t)i * ldv;
    const TIN* pt = text + (size_t)i * ldt;
    const bool unit_rows = prenormalized == 2;      // (see bwd_finish_kernel: unit rows given, gradient w.r.t. the un-normalised vectors)
    const double iv = (double)inv_norm[i], it = (double)inv_norm[g.bpad + i];
    const double ixv = unit_rows ? 1.0 : iv, ixt = unit_rows ? 1.0 : it;
    const float* grv = gbuf + (size_t)i * g.Dpad;
    const float* grt = gbuf + ((size_t)g.bpad + i) * g.Dpad;
    const size_t slice = (size_t)2 * g.bpad * g.Dpad;
    const double sc = (double)inv_tau / (2.0 * (double)Bglobal);
    const double pc = (double)inv_tau / (double)Bglobal * (lw ? 0.5 * ((double)lw[i] + (double)lw[g.bpad + i]) : 1.0);
    const bool clamped_v = iv >= 9.99e11 || prenormalized == 1, clamped_t = it >= 9.99e11 || prenormalized == 1;
    const double go = grad_out[0];
    double ghv[KC][4], ght[KC][4], xv[KC][4], xt[KC][4];
    double dotv = 0.0, dott = 0.0;
    f32x4 sv[KC], st[KC];
    double a[KC][4], c[KC][4];
    // fp32 rows made of whole, 16-byte aligned stretches (uniform over the launch): EVERY load of the row pair -- first slice, raw rows,
    // then the other slices, stretch by stretch -- is issued before the first value is used (12 KiB in flight per wave at D = 512 with two
    // slices; the per-stretch form below waits after every pair of loads).  Same additions in the same order: bit-identical results.
    bool whole = false;
    if constexpr (sizeof(TIN) == 4)
        whole = (g.D & 3) == 0 && (ldv & 3) == 0 && (ldt & 3) == 0 &&
                ((reinterpret_cast<uintptr_t>(video) | reinterpret_cast<uintptr_t>(text)) & 15) == 0;
    if (whole) {
        f32x4 ra[KC], rc[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int d = 4 * lane + 256 * k;
            if (d < g.D) {
                // (the slices are read exactly once: streaming loads)
                sv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grv + d));
                st[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grt + d));
                ra[k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(pv) + d);
                rc[k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(pt) + d);
            }
        }
        for (int sl = 1; sl < nslices; ++sl) {
            f32x4 uv[KC], ut[KC];
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int d = 4 * lane + 256 * k;
                if (d < g.D) {
                    uv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grv + sl * slice + d));
                    ut[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grt + sl * slice + d));
                }
            }
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int d = 4 * lane + 256 * k;
                if (d < g.D) {
                    sv[k] += uv[k];
                    st[k] += ut[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[k][j] = (double)ra[k][j];
                c[k][j] = (double)rc[k][j];
            }
    } else {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int d = 4 * lane + 256 * k;
            if (d < g.D) {
                sv[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grv + d));
                st[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grt + d));
                for (int sl = 1; sl < nslices; ++sl) {
                    sv[k] += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grv + sl * slice + d));
                    st[k] += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(grt + sl * slice + d));
                }
                row_load4(pv, d, g.D, a[k]);
                row_load4(pt, d, g.D, c[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int d = 4 * lane + 256 * k;
        if (d < g.D) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = d + j < g.D;
                ghv[k][j] = in ? ((double)sv[k][j] * sc - c[k][j] * ixt * pc) : 0.0;
                ght[k][j] = in ? ((double)st[k][j] * sc - a[k][j] * ixv * pc) : 0.0;
                xv[k][j] = a[k][j] * ixv;
                xt[k][j] = c[k][j] * ixt;
                dotv += xv[k][j] * ghv[k][j];
                dott += xt[k][j] * ght[k][j];
            }
        }
    }
    dotv = wave_sum_f64(dotv);
    dott = wave_sum_f64(dott);
    TIN* ov = gvideo + (size_t)i * ldgv;
    TIN* ot = gtext + (size_t)i * ldgt;
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int d = 4 * lane + 256 * k;
        if (d < g.D) {
            double a[4], c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = (clamped_v ? ghv[k][j] : (ghv[k][j] - xv[k][j] * dotv)) * iv * go;
                c[j] = (clamped_t ? ght[k][j] : (ght[k][j] - xt[k][j] * dott)) * it * go;
            }
            row_store4(ov, d, g.D, a);
            row_store4(ot, d, g.D, c);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K5: influential-sample statistics (NOT in the reference @ v1 -- the recipe is stated in
// oracle/influence_oracle.py).  All O(B Din) / O(B): HBM-bound row kernels, no B x B product:
//   conn_i = mean_j xhat_i . xhat_j (self pair masked) = (xhat_i . sum_j xhat_j - xhat_i . xhat_i) / B
//   a) partial column sums of the normalised rows (+ 1/||x_i||), b) their total, c) conn_i,
//   d) keep_i = conn_i / max(conn) < threshold,  omega_i = B rho_i / sum(rho),  rho = exp(conn / sum(conn) / kappa)
// blockIdx.y = modality (0 video, 1 text).
// ---------------------------------------------------------------------------------------------
constexpr int kInflBlocks = 256;    // partial column sums per modality (one block of 4 waves each)
constexpr int kInflMaxCols = 16;    // Din <= 256 * kInflMaxCols

// a) one wave per row at a time; a lane owns 4 consecutive columns of every 256-column stretch and keeps its running
//    column sums in registers; the 4 waves of a block meet in LDS at the end.  Rows are read once.
template <typename TIN, int KC>
__global__ void __launch_bounds__(256) infl_colsum_kernel(const TIN* xv, const TIN* xt, long ldv, long ldt, int n, int Din,
                                                          float* inv_norm, float* partial) {
    CROSSCLR_SHARED float red[4][256 * KC];
    const TIN* x = blockIdx.y == 0 ? xv : xt;
    const long ld = blockIdx.y == 0 ? ldv : ldt;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float acc[KC][4];
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < n; row += gridDim.x * 4) {
        const TIN* xr = x + (size_t)row * ld;
        double v[KC][4];
        double ss = 0.0;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int d = 4 * lane + 256 * k;
            if (d < Din) {
                row_load4(xr, d, Din, v[k]);
#pragma unroll
                for (int j = 0; j < 4; ++j) ss += v[k][j] * v[k][j];
            }
        }
        ss = wave_sum_f64(ss);
        const double nrm = sqrt(ss);
        const float inv = (float)(1.0 / (nrm > 1e-12 ? nrm : 1e-12));
        if (lane == 0) inv_norm[(size_t)blockIdx.y * n + row] = inv;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int d = 4 * lane + 256 * k;
            if (d < Din) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[k][j] += (float)v[k][j] * inv;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][256 * k + 4 * lane + j] = acc[k][j];
    __syncthreads();
    for (int d = tid; d < Din; d += 256)
        partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * Din + d] = (red[0][d] + red[1][d]) + (red[2][d] + red[3][d]);
}
// b) total of the block partials: 64 columns x 4 groups of blocks per thread block, fixed order
__global__ void __launch_bounds__(256) infl_colsum_finish_kernel(const float* partial, int nblocks, int Din, double* colsum) {
    CROSSCLR_SHARED double red[4][64];
    const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + c;
    double s = 0.0;
    if (d < Din) {
        const float* src = partial + (size_t)blockIdx.y * nblocks * Din + d;
#pragma unroll 16
        for (int b = grp; b < nblocks; b += 4) s += (double)src[(size_t)b * Din];   // independent loads, fixed order
    }
    red[grp][c] = s;
    __syncthreads();
    if (grp == 0 && d < Din) colsum[(size_t)blockIdx.y * Din + d] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
// one wave per row: conn = (xhat . colsum_total - xhat . xhat) / B_global
template <typename TIN>
__global__ void __launch_bounds__(256) infl_conn_kernel(const TIN* xv, const TIN* xt, long ldv, long ldt, int n, int Din,
                                                        const float* inv_norm, const double* colsum, int Bglobal,
                                                        double* conn) {
    const TIN* x = blockIdx.y == 0 ? xv : xt;
    const long ld = blockIdx.y == 0 ? ldv : ldt;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n) return;
    const double inv = (double)inv_norm[(size_t)blockIdx.y * n + row];
    const double* cs = colsum + (size_t)blockIdx.y * Din;
    double dot = 0.0, self = 0.0;
    for (int d = lane; d < Din; d += 64) {
        const double a = (double)(float)(in_load(x + (size_t)row * ld, d) * inv);   // xhat in fp32, like the column sums
        dot += a * cs[d];
        self += a * a;
    }
    dot = wave_sum_f64(dot);
    self = wave_sum_f64(self);
    if (lane == 0) conn[(size_t)blockIdx.y * n + row] = (dot - self) / (double)Bglobal;
}
// block-wide reductions over 1024 threads
__device__ __forceinline__ double block_reduce_f64(double v, bool take_max, double* red) {
    for (int m = 32; m >= 1; m >>= 1) {
        const double o = wave_xor_f64(v, m);
        v = take_max ? (o > v ? o : v) : v + o;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int k = 1; k < 16; ++k) r = take_max ? (red[k] > r ? red[k] : r) : r + red[k];
    return r;
}
// conn_all[world][2][n]; one block per modality writes keep / omega of THIS rank's rows in the [2][bpad] layout
__global__ void __launch_bounds__(1024) infl_finish_kernel(const double* conn_all, int world, int rank, int n, int bpad,
                                                           double threshold, double kappa, float* neg_scale, float* loss_weight) {
    CROSSCLR_SHARED double red[16];
    const int mod = blockIdx.x, tid = threadIdx.x;
    const int total = world * n;
    auto at = [&](int g) { return conn_all[((size_t)(g / n) * 2 + mod) * n + (g % n)]; };
    double mx = -1e300, mn = 1e300, sm = 0.0;
#pragma unroll 8
    for (int g = tid; g < total; g += 1024) { const double c = at(g); mx = c > mx ? c : mx; mn = c < mn ? c : mn; sm += c; }
    mx = block_reduce_f64(mx, true, red);
    mn = -block_reduce_f64(-mn, true, red);
    sm = block_reduce_f64(sm, false, red);
    const double zs = 1.0 / (sm * kappa);                 // z = conn * zs is monotone in conn: its maximum is analytic
    const double zmax = zs >= 0.0 ? mx * zs : mn * zs;
    double rs = 0.0;
#pragma unroll 8
    for (int g = tid; g < total; g += 1024) rs += exp(at(g) * zs - zmax);
    rs = block_reduce_f64(rs, false, red);
    for (int i = tid; i < bpad; i += 1024) {
        float k = 0.f, om = 0.f;
        if (i < n) {
            const double c = conn_all[((size_t)rank * 2 + mod) * n + i];
            k = (mx <= 0.0 || c / mx < threshold) ? 1.f : 0.f;
            om = (float)((double)total * exp(c * zs - zmax) / rs);
        }
        neg_scale[(size_t)mod * bpad + i] = k;
        loss_weight[(size_t)mod * bpad + i] = om;
    }
}

// ---------------------------------------------------------------------------------------------
// hardware assumption probes (run once on the GPU by tests/test_hw_assumptions.py)
// ---------------------------------------------------------------------------------------------
// which = 0: C = A(32x16 bf16) * B(16x32 bf16) through the documented fragment maps -> out f32[32][32]
//            in = A row-major [32][16] bf16 then B row-major [16][32] bf16
// which = 1: C = A(32x2 f32) * B(2x32 f32) -> out f32[32][32];  in = A [32][2] f32 then B [2][32] f32
// which = 2: transpose read: LDS holds bf16 M[r][c] = in[r*64+c] for a [16][64] matrix; every lane
//            issues lds_read_tr16_b64 at row 4*(lane>>5)... exactly like bwd_gemm2 (ks=0, dt=0, DC=64)
//            out s16[64][8] = the B fragment each lane assembled.
// which = 4: C = A(16x32 bf16) * B(32x16 bf16) with v_mfma_f32_16x16x32_bf16 -> out f32[16][16]
// which = 3: the cross-lane exchanges of the symmetric forward: out f32[5][64] = lane_xor<1,2,7,15,16>(in[lane])
__global__ void __launch_bounds__(64) selftest_kernel(int which, const void* in, void* out) {
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[64 * 64 * 2];
    const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
    if (which == 0) {
        const bf16_t* A = reinterpret_cast<const bf16_t*>(in);
        const bf16_t* B = A + 32 * 16;
        struct { bf16_t e[8]; } ta, tb;
        for (int j = 0; j < 8; ++j) {
            ta.e[j] = A[l31 * 16 + 8 * half + j];
            tb.e[j] = B[(8 * half + j) * 32 + l31];
        }
        const bf16x8 a = __builtin_bit_cast(bf16x8, ta), b = __builtin_bit_cast(bf16x8, tb);
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = mfma_32x32x16_bf16(a, b, c);
        float* C = reinterpret_cast<float*>(out);
        for (int r = 0; r < 16; ++r) C[frag_row(r, half) * 32 + l31] = c[r];
    } else if (which == 1) {
        const float* A = reinterpret_cast<const float*>(in);
        const float* B = A + 32 * 2;
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = mfma_32x32x2_f32(A[l31 * 2 + half], B[half * 32 + l31], c);
        float* C = reinterpret_cast<float*>(out);
        for (int r = 0; r < 16; ++r) C[frag_row(r, half) * 32 + l31] = c[r];
    } else if (which == 4) {
        // C = A(16x32 bf16) * B(32x16 bf16) through the 16x16x32 fragment maps -> out f32[16][16]
        const bf16_t* A = reinterpret_cast<const bf16_t*>(in);
        const bf16_t* B = A + 16 * 32;
        const int i16 = lane & 15, g4 = lane >> 4;
        struct { bf16_t e[8]; } ta, tb;
        for (int j = 0; j < 8; ++j) {
            ta.e[j] = A[i16 * 32 + 8 * g4 + j];
            tb.e[j] = B[(8 * g4 + j) * 16 + i16];
        }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = mfma_16x16x32_bf16(__builtin_bit_cast(bf16x8, ta), __builtin_bit_cast(bf16x8, tb), c);
        float* C = reinterpret_cast<float*>(out);
        for (int r = 0; r < 4; ++r) C[(4 * g4 + r) * 16 + i16] = c[r];
    } else if (which == 3) {
        // out f32[5][64]: lane_xor<1,2,7,15,16> of in[lane]
        const float v = reinterpret_cast<const float*>(in)[lane];
        float* O = reinterpret_cast<float*>(out);
        O[0 * 64 + lane] = lane_xor<1>(v);
        O[1 * 64 + lane] = lane_xor<2>(v);
        O[2 * 64 + lane] = lane_xor<7>(v);
        O[3 * 64 + lane] = lane_xor<15>(v);
        O[4 * 64 + lane] = lane_xor<16>(v);
    } else {
        const bf16_t* M = reinterpret_cast<const bf16_t*>(in);
        for (int e = lane; e < 64 * 64; e += 64) {
            int q = e >> 6, d = e & 63;
            *reinterpret_cast<bf16_t*>(lds + xq_off<bf16_t, 64>(q, d)) = M[e];
        }
        __syncthreads();
        const int grp = lane >> 4, i16 = lane & 15, jrow = i16 >> 2, piece = i16 & 3, dsub = grp & 1;
        const int dcol = 16 * dsub + 4 * piece, q0 = 8 * half + jrow;
        s16x4 lo = lds_read_tr16_b64(lds + xq_off<bf16_t, 64>(q0, dcol));
        s16x4 hi = lds_read_tr16_b64(lds + xq_off<bf16_t, 64>(q0 + 4, dcol));
        short* O = reinterpret_cast<short*>(out);
        for (int e = 0; e < 4; ++e) { O[lane * 8 + e] = lo[e]; O[lane * 8 + 4 + e] = hi[e]; }
    }
}

}  // namespace crossclr
