// crossclr_kernels_hvp.h -- second-order terms of the loss on the device (crossclr_second_order, include/crossclr.h).
//
// The reference's forward (trainer/loss.py:79-114) is a chain of eager PyTorch ops, so autograd can differentiate its backward
// again (create_graph=True: gradient penalties, Hessian-vector products).  What that double backward computes, in closed form:
// with g = grad_out * dL/d(rows) the first backward's output and u the cotangent handed in for it,
//     d<u, g>/d(rows)      = grad_out * H u        (H = the Hessian of the loss w.r.t. the input rows)
//     d<u, g>/d(grad_out)  = <u, dL/d(rows)>.
// H is symmetric, so H u is the DIRECTIONAL DERIVATIVE of the gradient along u (forward-mode through SURVEY.md 3.5's closed form):
//     unit rows x_p = r_p / n_p            tangent  v_p = (u_p - x_p (x_p . u_p)) / n_p
//     logits a_pq = s_pq x_p . x_q / tau   tangent  adot_pq = s_pq (v_p . x_q + x_p . v_q) / tau        (s = 1 across, w inside a modality)
//     E_pq = exp(a_pq - shift), Z_p = sum_q k_q E_pq (+ the masked self pair's constant)   dZ_p = sum_q k_q E_pq adot_pq
//     rz_p = omega_p / Z_p                 tangent  drz_p = -rz_p dZ_p / Z_p
//     G_p  = sum_q s E_pq (rz_p k_q + rz_q k_p) x_q                                         (the first-order gradient product)
//     dG_p = sum_q s E_pq [adot_pq (rz_p k_q + rz_q k_p) + (drz_p k_q + drz_q k_p)] x_q  +  sum_q s E_pq (rz_p k_q + rz_q k_p) v_q
// followed by the row-local chain through the normalisation and the positive-pair term (hvp_finish_kernel).
// Two passes of one tiled kernel on the skeleton of the generic backward (crossclr_kernels_generic.h, bwd_kernel): PASS 1 leaves the row sums
// dZ, PASS 2 the product dG.  Exact-fp32 products (v_mfma_f32_32x32x2_f32) whatever mode the first-order step ran in: the reference
// differentiates float64 logits here, and second-order terms are not on the training step's hot path.
#pragma once

#include "crossclr_kernels_generic.h"

namespace crossclr {

// packed tangent operand V[2][bpad][Dpad] (fp32) from the cotangent rows u (input dtype) and the raw rows: v = (u - x (x . u)) / n; rows whose norm
// was clamped (||r|| < eps: x = r / eps, loss.py:79-80's F.normalize) and unit rows given as such: v = u * inv_norm.  One wave per row.
template <typename TIN>
__global__ void __launch_bounds__(256) hvp_tangent_kernel(const TIN* video, const TIN* text, long ldv, long ldt, const TIN* uvideo, const TIN* utext,
                                                          long lduv, long ldut, Geo g, const float* inv_norm, int prenormalized, float* V) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + wave;      // 0 .. 2*bpad-1 (padding rows are cleared)
    if (idx >= 2 * g.bpad) return;
    const int mod = idx / g.bpad, i = idx - mod * g.bpad;
    float* out = V + (size_t)idx * g.Dpad;
    if (i >= g.b) {
        for (int d = 4 * lane; d < g.Dpad; d += 256) { const float z[4] = {0.f, 0.f, 0.f, 0.f}; op_store4(out, d, z); }
        return;
    }
    const TIN* own = mod == 0 ? video + (size_t)i * ldv : text + (size_t)i * ldt;
    const TIN* u = mod == 0 ? uvideo + (size_t)i * lduv : utext + (size_t)i * ldut;
    const double io = (double)inv_norm[mod * g.bpad + i];
    const bool clamped = io >= 9.99e11 || prenormalized != 0;
    double dot = 0.0;
    if (!clamped) {
        for (int d = 4 * lane; d < g.D; d += 256) {
            double x[4], uu[4];
            row_load4(own, d, g.D, x);
            row_load4(u, d, g.D, uu);
#pragma unroll
            for (int j = 0; j < 4; ++j) dot += x[j] * io * uu[j];
        }
        dot = wave_sum_f64(dot);
    }
    for (int d = 4 * lane; d < g.Dpad; d += 256) {
        double x[4], uu[4];
        row_load4(own, d, g.D, x);
        row_load4(u, d, g.D, uu);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (d + j < g.D) ? (float)((clamped ? uu[j] : uu[j] - x[j] * io * dot) * io) : 0.f;
        op_store4(out, d, v);
    }
}

template <int DC, int PASS> struct HvpLds {
    static constexpr int kPX = 0, kQX = 64 * 128, kPV = 2 * 64 * 128, kQV = 3 * 64 * 128;
    // PASS 2: the two weight tiles [64][64] fp32 and the column slices [64][DC] of X and V.  DC = 256: the slices alone are 128 KiB, so the
    // weight tiles take the place of the four K-tiles (dead between phase A and the next tile's first commit: one more barrier per tile) --
    // 160 KiB, the whole LDS of a CU, for half the phase-A recomputation of DC = 128
    static constexpr bool kAlias = PASS == 2 && DC > 128;
    static constexpr int kWM = kAlias ? 0 : 4 * 64 * 128;
    static constexpr int kWW = kWM + 64 * 64 * 4;
    static constexpr int kXQ = 4 * 64 * 128 + (kAlias ? 0 : 2 * 64 * 64 * 4);
    static constexpr int kVQ = kXQ + 64 * DC * 4;
    static constexpr int kTotal = PASS == 2 ? kVQ + 64 * DC * 4 : 4 * 64 * 128 + 2 * 64 * 4;      // PASS 1: two waves' partial row sums
};

// grid = (2*bpad/64 row blocks, PASS 2: Dpad/DC output slices, column slices z).  Single device: rows and columns are the same operand.
//   PASS 1: out[z][2*bpad]          = this slice's share of dZ_p = sum_q k_q E_pq adot_pq            (natural units, E relative to the row's shift)
//   PASS 2: out[z][2*bpad][Dpad]    = this slice's share of dG_p (without 1 / tau, like the first-order gbuf)
// rz / wrz: omega / Z and w omega / Z (crossclr_forward_finish); drz / dwrz: their tangents (hvp_stats_kernel; PASS 2 only);
// k: negative scales (SW) or NULL; shift: per-row soft-max shifts of the two-pass regime (log2 domain) or NULL (g.m2 for every row).
// NS (PASS 2): output slices of DC columns one block owns (grid.y = Dpad / (DC * NS)): the tile's S and T -- three of the five products -- are
// evaluated once and multiplied with NS column slices that pass through the same LDS one after the other (NS * DC / 64 accumulators per wave).
template <int DC, bool SW, int PASS, int NS = 1>
__global__ void __launch_bounds__(256) hvp_kernel(const float* X, const float* V, Geo g, const float* rz, const float* wrz, const float* drz,
                                                  const float* dwrz, const float* k, const float* shift, float* out, int tiles_per_slice) {
    typedef Operand<float> Op;
    typedef HvpLds<DC, PASS> L;
    CROSSCLR_SHARED __attribute__((aligned(16))) unsigned char lds[L::kTotal];
    unsigned char *tPX = lds + L::kPX, *tQX = lds + L::kQX, *tPV = lds + L::kPV, *tQV = lds + L::kQV;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const size_t pitch = (size_t)g.Dpad * sizeof(float);
    const int nchunks = g.Dpad / Op::kChunkElems;
    const int row0 = blockIdx.x * 64;
    const int rmod = row0 / g.bpad;
    const int r_in_mod0 = row0 - rmod * g.bpad;
    const int d0 = PASS == 2 ? blockIdx.y * DC * NS : 0;
    const unsigned char* xrow = reinterpret_cast<const unsigned char*>(X) + (size_t)row0 * pitch;
    const unsigned char* vrow = reinterpret_cast<const unsigned char*>(V) + (size_t)row0 * pitch;
    const int wq = wave & 1, wp = wave >> 1;      // phase A / B roles: S^T for columns 32*wq.., rows 32*wp..
    const int wr = wave & 1, wc = wave >> 1;      // phase C roles (PASS 2): rows 32*wr.., embedding columns wc*(DC/2)..

    constexpr int kAcc = PASS == 2 ? NS * (DC / 64) : 1;
    f32x16 acc2[kAcc];
#pragma unroll
    for (int dt = 0; dt < kAcc; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[dt][r] = 0.f;
    float dz = 0.f;

    const int p_t = 32 * wp + l31;                // this lane's row inside the block (phase B)
    const float rzp_inter = rz[row0 + p_t], rzp_intra = wrz[row0 + p_t];
    const float drzp_inter = PASS == 2 ? drz[row0 + p_t] : 0.f, drzp_intra = PASS == 2 ? dwrz[row0 + p_t] : 0.f;
    const float kp = SW ? k[row0 + p_t] : 1.f;
    const float shp = shift ? shift[row0 + p_t] : g.m2;

    const int ntiles = 2 * g.bpad / 64;
    const int t_begin = blockIdx.z * tiles_per_slice;
    int t_stop = t_begin + tiles_per_slice;
    if (t_stop > ntiles) t_stop = ntiles;
    KTileStage<64, 256> spx, sqx, spv, sqv;
    // the column slices [64][DC] of X and V the product phase multiplies with, columns dcol .. dcol + DC - 1 of the tile's rows
    auto load_slices = [&](const unsigned char* xcol, const unsigned char* vcol, int dcol) {
        constexpr int kPiecesPerRow = DC * 4 / 16;
        unsigned char *xq = lds + L::kXQ, *vq = lds + L::kVQ;
        for (int id = tid; id < 64 * kPiecesPerRow; id += 256) {
            const int q = id / kPiecesPerRow, c = id - q * kPiecesPerRow;
            const size_t src = (size_t)q * pitch + ((size_t)dcol + c * 4) * sizeof(float);
            *reinterpret_cast<u32x4*>(xq + xq_off<float, DC>(q, c * 4)) = *reinterpret_cast<const u32x4*>(xcol + src);
            *reinterpret_cast<u32x4*>(vq + xq_off<float, DC>(q, c * 4)) = *reinterpret_cast<const u32x4*>(vcol + src);
        }
    };
    for (int u = t_begin; u < t_stop; ++u) {
        const ColTile ct = col_tile(g, u, 64);
        const unsigned char* xcol = reinterpret_cast<const unsigned char*>(X) + ct.row0 * pitch;
        const unsigned char* vcol = reinterpret_cast<const unsigned char*>(V) + ct.row0 * pitch;
        // ---------------- phase A: S^T = X_Q X_P^T and T^T = V_Q X_P^T + X_Q V_P^T ----------------
        f32x16 accS, accT;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accS[r] = 0.f; accT[r] = 0.f; }
        spx.fetch(xrow, pitch, 0, tid); sqx.fetch(xcol, pitch, 0, tid);
        spv.fetch(vrow, pitch, 0, tid); sqv.fetch(vcol, pitch, 0, tid);
        if (L::kAlias) __syncthreads();      // (the previous tile's weight tiles live where the K-tiles are committed next: every wave past its phase C)
        for (int kc = 0; kc < nchunks; ++kc) {
            spx.commit(tPX, tid); sqx.commit(tQX, tid); spv.commit(tPV, tid); sqv.commit(tQV, tid);
            __syncthreads();      // also: every wave is past phase C of the previous tile
            if (PASS == 2 && NS == 1 && kc == 0) load_slices(xcol, vcol, d0);      // (one slice: its loads ride behind phase A)
            if (kc + 1 < nchunks) {
                spx.fetch(xrow, pitch, (kc + 1) * 128, tid); sqx.fetch(xcol, pitch, (kc + 1) * 128, tid);
                spv.fetch(vrow, pitch, (kc + 1) * 128, tid); sqv.fetch(vcol, pitch, (kc + 1) * 128, tid);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const Op::frag ax = Op::load(tQX, 32 * wq + l31, s, half), av = Op::load(tQV, 32 * wq + l31, s, half);
                const Op::frag bx = Op::load(tPX, 32 * wp + l31, s, half), bv = Op::load(tPV, 32 * wp + l31, s, half);
                accS = Op::mma(ax, bx, accS);
                accT = Op::mma(av, bx, accT);
                accT = Op::mma(ax, bv, accT);
            }
            __syncthreads();
        }
        // ---------------- phase B ----------------
        const bool same_mod = ct.mod == rmod;
        const float c2 = same_mod ? g.c_intra : g.c_inter;           // log2(e) s / tau
        const float cn = c2 * kLn2;                                   // s / tau
        const float rzp = same_mod ? rzp_intra : rzp_inter;
        const float drzp = same_mod ? drzp_intra : drzp_inter;
        const float* rzq = (same_mod ? wrz : rz) + ct.stat0;
        const float* drzq = PASS == 2 ? (same_mod ? dwrz : drz) + ct.stat0 : nullptr;
        const bool diag_tile = same_mod && ct.in_mod0 == r_in_mod0;
        const float kpe = (SW && same_mod) ? kp : 1.f;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int q0 = 32 * wq + 8 * r4 + 4 * half;      // frag_row(4*r4 + j, half) = q0 - 32wq + j
            f32x4 kq = {1.f, 1.f, 1.f, 1.f};
            if (SW && same_mod) kq = *reinterpret_cast<const f32x4*>(k + ct.stat0 + q0);
            f32x4 shq = {g.m2, g.m2, g.m2, g.m2};
            if (shift) shq = *reinterpret_cast<const f32x4*>(shift + ct.stat0 + q0);
            f32x4 rq = {0.f, 0.f, 0.f, 0.f}, drq = {0.f, 0.f, 0.f, 0.f};
            if (PASS == 2) { rq = *reinterpret_cast<const f32x4*>(rzq + q0); drq = *reinterpret_cast<const f32x4*>(drzq + q0); }
            f32x4 wm, ww;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = accS[4 * r4 + j] * c2;
                const float adot = accT[4 * r4 + j] * cn;
                const float ep = fast_exp2(x - shp);
                const bool masked = diag_tile && q0 + j == p_t;      // loss.py:96-97: the self pair's logit is the constant 0
                if (PASS == 1) {
                    if (!masked) dz += kq[j] * ep * adot;
                } else {
                    const float eq = shift ? fast_exp2(x - shq[j]) : ep;
                    const float w0 = ep * rzp * kq[j] + eq * rq[j] * kpe;                 // the first-order weight
                    const float w1 = ep * drzp * kq[j] + eq * drq[j] * kpe;
                    ww[j] = masked ? 0.f : w0;
                    wm[j] = masked ? 0.f : adot * w0 + w1;
                }
            }
            if (PASS == 2) {
                w_store4(lds + L::kWM, p_t, q0, wm, (float*)nullptr);
                w_store4(lds + L::kWW, p_t, q0, ww, (float*)nullptr);
            }
        }
        if (PASS == 2) {
            // ---------------- phase C: dG += M X_Q + W V_Q ----------------
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                __syncthreads();      // the weight tiles are written (sl = 0) / every wave is past the product of slice sl - 1
                if (NS > 1) {
                    load_slices(xcol, vcol, d0 + sl * DC);
                    __syncthreads();
                }
                f32x16(&acc)[DC / 64] = *reinterpret_cast<f32x16(*)[DC / 64]>(&acc2[sl * (DC / 64)]);
                bwd_gemm2<DC>(lds + L::kWM, lds + L::kXQ, wr, wc * (DC / 2), lane, acc, (float*)nullptr);
                bwd_gemm2<DC>(lds + L::kWW, lds + L::kVQ, wr, wc * (DC / 2), lane, acc, (float*)nullptr);
            }
        }
    }
    if (PASS == 1) {
        // row p's share: the two wave halves (q = ..+4*half), then the two waves that hold the same rows (wq = 0, 1), in that order
        dz += wave_xor_f32(dz, 32);
        float* red = reinterpret_cast<float*>(lds + 4 * 64 * 128);
        __syncthreads();
        if (wq == 1 && half == 0) red[p_t] = dz;
        __syncthreads();
        if (wq == 0 && half == 0) out[(size_t)blockIdx.z * 2 * g.bpad + row0 + p_t] = dz + red[p_t];
        return;
    }
    float* gslice = out + (size_t)blockIdx.z * 2 * g.bpad * g.Dpad;
#pragma unroll
    for (int sl = 0; sl < (PASS == 2 ? NS : 1); ++sl)
#pragma unroll
        for (int dt = 0; dt < (PASS == 2 ? DC / 64 : 1); ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                gslice[(size_t)(row0 + 32 * wr + frag_row(r, half)) * g.Dpad + d0 + sl * DC + wc * (DC / 2) + 32 * dt + l31] = acc2[sl * (DC / 64) + dt][r];
}

// dZ slices -> drz = -rz dZ / Z and dwrz = w drz.  1 / Z = rz / omega (omega = the row's loss weight; 1 without sample weights): a row of weight 0
// has rz = 0 and contributes nothing to anybody's gradient, whatever its Z is.
__global__ void __launch_bounds__(256) hvp_stats_kernel(const float* dzpart, int nz, int n2, const float* rz, const float* lw, float negative_weight,
                                                        float* drz, float* dwrz) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n2) return;
    float dz = 0.f;
    for (int z = 0; z < nz; ++z) dz += dzpart[(size_t)z * n2 + p];       // fixed order
    const float r = rz[p];
    const float om = lw ? lw[p] : 1.f;
    const float inv_z = om != 0.f ? r / om : 0.f;
    const float d = -r * dz * inv_z;
    drz[p] = d;
    dwrz[p] = negative_weight * d;
}

// Row-local end of the double backward, one wave per row (fp64 row arithmetic like bwd_finish_kernel):
//   Ghat  = sum_slices(gbuf1) / (2 B tau) - pc xo            (xo = the positive partner's unit row, pc = (omega_p + omega_p') / (2 B tau))
//   dGhat = sum_slices(gbuf2) / (2 B tau) - pc vo            (vo = the partner's tangent)
//   g  = (Ghat - x (x . Ghat)) / n                                                      = dL/d(row)     (grad_out = 1)
//   dg = (dGhat - v (x . Ghat) - x (v . Ghat) - x (x . dGhat)) / n - g (x . u) / n      = (H u)_row
// out row = grad_out * dg; dgo_rows[row] = <u, g> (summed by hvp_reduce_kernel into d<u, grad>/d(grad_out)).
// Clamped rows (||r|| < eps) and unit rows given as such (prenormalized = 1): n is a constant and there is no projection: g = Ghat / n, dg = dGhat / n.
template <typename TIN>
__global__ void __launch_bounds__(256) hvp_finish_kernel(const float* gbuf1, int nsl1, const float* gbuf2, int nsl2, const TIN* video, const TIN* text,
                                                         long ldv, long ldt, const TIN* uvideo, const TIN* utext, long lduv, long ldut, Geo g,
                                                         const float* inv_norm, float inv_tau, int Bglobal, const double* grad_out, const float* lw,
                                                         int prenormalized, TIN* hvideo, TIN* htext, long ldhv, long ldht, double* dgo_rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + wave;      // 0 .. 2*b-1
    if (idx >= 2 * g.b) return;
    const int mod = idx / g.b, i = idx - mod * g.b;
    const TIN* own = mod == 0 ? video + (size_t)i * ldv : text + (size_t)i * ldt;
    const TIN* oth = mod == 0 ? text + (size_t)i * ldt : video + (size_t)i * ldv;
    const TIN* uown = mod == 0 ? uvideo + (size_t)i * lduv : utext + (size_t)i * ldut;
    const TIN* uoth = mod == 0 ? utext + (size_t)i * ldut : uvideo + (size_t)i * lduv;
    TIN* out = mod == 0 ? hvideo + (size_t)i * ldhv : htext + (size_t)i * ldht;
    const double io = (double)inv_norm[mod * g.bpad + i], ip = (double)inv_norm[(1 - mod) * g.bpad + i];
    const bool pre = prenormalized != 0;
    const bool clamped = io >= 9.99e11 || pre, oclamped = ip >= 9.99e11 || pre;
    const size_t rowoff = ((size_t)mod * g.bpad + i) * g.Dpad;
    const size_t slice = (size_t)2 * g.bpad * g.Dpad;
    const double sc = (double)inv_tau / (2.0 * (double)Bglobal);
    const double pc = (double)inv_tau / (double)Bglobal * (lw ? 0.5 * ((double)lw[i] + (double)lw[g.bpad + i]) : 1.0);
    const double go = grad_out[0];
    // pass A: x . u (own) and xo . uo (partner)
    double xu = 0.0, xouo = 0.0;
    for (int d = 4 * lane; d < g.D; d += 256) {
        double x[4], o[4], u[4], uo[4];
        row_load4(own, d, g.D, x); row_load4(oth, d, g.D, o); row_load4(uown, d, g.D, u); row_load4(uoth, d, g.D, uo);
#pragma unroll
        for (int j = 0; j < 4; ++j) { xu += x[j] * io * u[j]; xouo += o[j] * ip * uo[j]; }
    }
    xu = wave_sum_f64(xu);
    xouo = wave_sum_f64(xouo);
    auto row_values = [&](int d, double (&x)[4], double (&v)[4], double (&gh)[4], double (&dgh)[4], double (&u)[4]) {
        double r[4], o[4], uo[4];
        row_load4(own, d, g.D, r); row_load4(oth, d, g.D, o); row_load4(uown, d, g.D, u); row_load4(uoth, d, g.D, uo);
        f32x4 s1 = *reinterpret_cast<const f32x4*>(gbuf1 + rowoff + d);
        for (int sl = 1; sl < nsl1; ++sl) s1 += *reinterpret_cast<const f32x4*>(gbuf1 + sl * slice + rowoff + d);
        f32x4 s2 = *reinterpret_cast<const f32x4*>(gbuf2 + rowoff + d);
        for (int sl = 1; sl < nsl2; ++sl) s2 += *reinterpret_cast<const f32x4*>(gbuf2 + sl * slice + rowoff + d);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = d + j < g.D;
            x[j] = r[j] * io;
            v[j] = (clamped ? u[j] : u[j] - x[j] * xu) * io;
            const double xo = o[j] * ip;
            const double vo = (oclamped ? uo[j] : uo[j] - xo * xouo) * ip;
            gh[j] = in ? (double)s1[j] * sc - xo * pc : 0.0;
            dgh[j] = in ? (double)s2[j] * sc - vo * pc : 0.0;
        }
    };
    // pass B: the three dot products
    double xg = 0.0, vg = 0.0, xdg = 0.0;
    for (int d = 4 * lane; d < g.D; d += 256) {
        double x[4], v[4], gh[4], dgh[4], u[4];
        row_values(d, x, v, gh, dgh, u);
#pragma unroll
        for (int j = 0; j < 4; ++j) { xg += x[j] * gh[j]; vg += v[j] * gh[j]; xdg += x[j] * dgh[j]; }
    }
    xg = wave_sum_f64(xg); vg = wave_sum_f64(vg); xdg = wave_sum_f64(xdg);
    // pass C: the row of H u, and <u, g>
    double ug = 0.0;
    for (int d = 4 * lane; d < g.D; d += 256) {
        double x[4], v[4], gh[4], dgh[4], u[4], h[4];
        row_values(d, x, v, gh, dgh, u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double gr = (clamped ? gh[j] : gh[j] - x[j] * xg) * io;
            const double dg = clamped ? dgh[j] * io : (dgh[j] - v[j] * xg - x[j] * vg - x[j] * xdg) * io - gr * xu * io;
            h[j] = dg * go;
            ug += u[j] * gr;
        }
        row_store4(out, d, g.D, h);
    }
    ug = wave_sum_f64(ug);
    if (lane == 0) dgo_rows[idx] = ug;
}

// out[0] = sum of the per-row values, one wave, fixed order (the lanes and strides of fwd_finish_reduce_kernel)
__global__ void __launch_bounds__(64) hvp_reduce_kernel(const double* rows, int n, double* out) {
    double acc = 0.0;
    for (int k = threadIdx.x; k < n; k += 64) acc += rows[k];
    acc = wave_sum_f64(acc);
    if (threadIdx.x == 0) out[0] = acc;
}

}  // namespace crossclr
