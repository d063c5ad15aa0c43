// crossclr_kernels_saved32.h -- the exact-fp32 backward that does not recompute the similarity product.
//
// compute_mode="fp32" runs the generic tiled kernels; their backward (bwd_kernel) re-evaluates S = X X^T for every 64 x 64 tile
// and every slice of D before it can form the weights: 24 B^2 D executed for 8 B^2 D of gradient products.  Here the forward
// (fwd_sums_kernel<float, SW, 0, true>) leaves its fp32 exponentials behind -- the analogue of the [B, 2B] float64 tensors
// autograd keeps for the reference (trainer/loss.py:96-100, 59-60), 1/12 of their size -- and the backward is the gradient
// product alone:   G[p][d] = sum_q  E[p][q] (rz_p + rz_q)  X[q][d]            (v_mfma_f32_32x32x2_f32).
//
//   block = 64 rows x DC embedding columns, 4 waves as (32-row half) x (DC/2 column half); 32-column tiles of the stacked operand
//   * the E fragment of (rows, tile) comes from the stash in the register layout the forward's MFMA left it in (16 bytes per lane,
//     coalesced) -- which IS the A-operand layout of the gradient product (lane = row p, k-slot = column q): W never touches LDS;
//   * the [32][DC] slice of X arrives by buffer-addressed LDS-DMA, two stages, one DMA-preserving barrier per tile;
//   * the loads of tile t+1 (E, statistics, X slice) are issued before the 64 MFMAs of tile t;
//   * 2 blocks per CU (<= 80 KiB of LDS, <= 256 registers): the other block's MFMAs cover this block's weight arithmetic.
// RECT (round 4): the same product for a RECTANGULAR block -- this rank's rows against the columns of other ranks (g = the rank range of
// crossclr_forward_rect_save, wrapping inside the gathered operand `xc`; column statistics from the gathered arrays) -- from the fp32
// fragments the generic forward saved for that launch ([row group][fragments of the launch's column range]).
#pragma once

namespace crossclr {

// RM (two-pass soft-max, small temperatures): the stash holds U[p][q] = exp2(x - shift_p) and, behind it, Ut[p][q] = U[q][p];
// rz = omega / (row sum relative to the ROW's shift), so the weight is U[p][q] rz_p + Ut[p][q] rz_q.  RM + RECT: the same for the block
// against other ranks' columns (crossclr_forward_rect_save_s evaluated Ut with the gathered shifts of those ranks' rows).
template <int DC, bool SW, bool RM = false, bool RECT = false>
__global__ void __launch_bounds__(256, 2) bwd_saved32_kernel(const float* x, const float* stash, Geo g, const float* rz, const float* wrz,
                                                             float* gbuf, int accumulate, int tiles_per_slice, const float* k,
                                                             const float* rzc, const float* wrzc, const float* kc) {
    // (RECT: x = the gathered column operand, rz / wrz / k = this rank's ROW statistics, rzc / wrzc / kc = the gathered column statistics)
    constexpr int QT = 32;
    constexpr int STG = QT * DC * 4;          // bytes per stage: [32][DC] floats, rows contiguous
    constexpr int NR = DC / 32;               // DMA rounds per tile: 256 threads x 16 B = 4 KiB each
    constexpr int NDT = DC / 64;              // 32-column output fragments per wave
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[2 * STG];

    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wr = wave & 1, wc = wave >> 1;
    const int row0 = blockIdx.x * 64, d0 = blockIdx.y * DC;
    const int rmod = row0 / g.bpad;
    const int NQ = (RECT ? g.col_ranks : 1) * (2 * g.bpad / QT);
    const int t_begin = blockIdx.z * tiles_per_slice;
    int t_stop = t_begin + tiles_per_slice;
    if (t_stop > NQ) t_stop = NQ;

    f32x16 acc[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;

    const int p = row0 + 32 * wr + l31;                     // this lane's row of the stacked operand
    const float rzp_inter = rz[p], rzp_intra = wrz[p];
    const float kp = SW ? k[p] : 1.f;
    const size_t pitch = (size_t)g.Dpad * 4;
    const BufRsrc rs_x = make_rsrc(x, (unsigned)((size_t)(RECT ? (g.col_wrap > 0 ? g.col_wrap : g.col_ranks) : 1) * 2 * g.bpad * pitch));
    unsigned voff[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int L = (4 * r + wave) * 1024 + lane * 16;   // LDS byte this lane fills in round r
        const int q = L / (DC * 4), byte = L - q * (DC * 4);
        voff[r] = (unsigned)((size_t)q * pitch + (size_t)d0 * 4 + byte);
    }
    auto issue_x = [&](int t, int stage) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
            lds_dma16_buf(rs_x, voff[r], (unsigned)((RECT ? col_tile(g, t, QT).row0 : (size_t)t * QT) * pitch), lds + stage * STG + (4 * r + wave) * 1024);
    };
    // fragment (p32, t) of the stash: [r4][lane][4]
    const float* frag_row0 = stash + (((size_t)(row0 / 32 + wr) * (size_t)NQ) << 10) + 4 * lane;
    f32x4 e[4], et[4], rq[4], kq[4];
    const size_t nn = (size_t)(2 * g.bpad) * (size_t)NQ * 32;      // floats of U (Ut behind it): rows x columns of the launch
    auto fetch = [&](int t) {
        const ColTile ct = RECT ? col_tile(g, t, QT) : ColTile{};
        const bool same = RECT ? (ct.mod == rmod) : ((t * QT >= g.bpad) == (rmod == 1));
        const size_t s0 = RECT ? ct.stat0 : (size_t)t * QT;
        const float* stat = (same ? (RECT ? wrzc : wrz) : (RECT ? rzc : rz)) + s0 + 4 * half;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            e[r4] = *reinterpret_cast<const f32x4*>(frag_row0 + ((size_t)t << 10) + 256 * r4);
            if (RM) et[r4] = *reinterpret_cast<const f32x4*>(frag_row0 + nn + ((size_t)t << 10) + 256 * r4);
            rq[r4] = *reinterpret_cast<const f32x4*>(stat + 8 * r4);
            if (SW) kq[r4] = *reinterpret_cast<const f32x4*>((RECT ? kc : k) + s0 + 4 * half + 8 * r4);
        }
    };
    if (t_begin < t_stop) { issue_x(t_begin, 0); fetch(t_begin); }
    int stage = 0;
    for (int t = t_begin; t < t_stop; ++t) {
        wait_dma();                 // X slice of tile t (own pieces) and the E / statistics registers of tile t
        barrier_keep_dma();         // ... for every wave; and every wave is done with tile t-1's stage
        // weights of tile t: W[p][q] = E (rz_p + rz_q); sample weights: E (rz_p k_q + rz_q k_p) inside a modality
        const bool same = RECT ? (col_tile(g, t, QT).mod == rmod) : ((t * QT >= g.bpad) == (rmod == 1));
        const float rzp = same ? rzp_intra : rzp_inter;
        float w[16];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w[4 * r4 + j] = RM ? ((SW && same) ? e[r4][j] * rzp * kq[r4][j] + et[r4][j] * rq[r4][j] * kp : e[r4][j] * rzp + et[r4][j] * rq[r4][j])
                                   : ((SW && same) ? e[r4][j] * (rzp * kq[r4][j] + rq[r4][j] * kp) : e[r4][j] * (rzp + rq[r4][j]));
        if (t + 1 < t_stop) { issue_x(t + 1, stage ^ 1); fetch(t + 1); }
        // G[p][d] += W[p][q] X[q][d]: k-slot `half` of step (kk, j) is column q = 8 kk + 4 half + j -- the fragment's own order
        const unsigned char* xs = lds + stage * STG + ((4 * half) * DC + wc * (DC / 2) + l31) * 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const float bv = *reinterpret_cast<const float*>(xs + ((8 * kk + j) * DC + 32 * dt) * 4);
                    acc[dt] = mfma_32x32x2_f32(w[4 * kk + j], bv, acc[dt]);
                }
        stage ^= 1;
    }
    wait_dma();
    float* gslice = gbuf + (size_t)blockIdx.z * 2 * g.bpad * g.Dpad;
    if (accumulate) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                gslice[(size_t)(row0 + 32 * wr + frag_row(r, half)) * g.Dpad + d0 + wc * (DC / 2) + 32 * dt + l31] += acc[dt][r];
    } else {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                gslice[(size_t)(row0 + 32 * wr + frag_row(r, half)) * g.Dpad + d0 + wc * (DC / 2) + 32 * dt + l31] = acc[dt][r];
    }
}

}  // namespace crossclr
