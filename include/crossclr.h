/*
 * crossclr.h -- C-ABI of the MI355X-native CrossCLR contrastive-loss hot path.
 *
 * The reference (amazon-science/crossmodal-contrastive-learning @ v1) has no FFI: the operator API
 * of this path is the Python class `CrossCLR_onlyIntraModality` (trainer/loss.py:44-114).  This
 * header is the boundary a binding for that class sits on: plain pointers and sizes, no torch
 * types, caller-owned buffers, no device allocation, no host synchronisation; every entry point
 * enqueues on the HIP stream it is given and returns 0 or a negative error code
 * (`crossclr_last_error()` has the text).  `INTEGRATION.md` shows the ctypes stub.
 *
 * Which reference lines each entry point replaces:
 *   crossclr_normalize      trainer/loss.py:79-80   F.normalize of both modalities (+ diag of :83)
 *   crossclr_forward        trainer/loss.py:83-100, 59-60 (softmax denominators), 111-112
 *   crossclr_forward_finish trainer/loss.py:60 (-log), :114 (means)  -> per-row logZ, loss sum
 *   crossclr_backward       autograd of :83-112 (SURVEY.md section 3.5 closed form)
 *   crossclr_backward_finish autograd of :79-80 (normalize backward) + the analytic -2*delta term
 *   crossclr_*_w            the same four with per-sample weights (influential-sample
 *                           pruning / loss weighting -- NOT in the reference @ v1, whose only weight is the scalar
 *                           `negative_weight` of :56,99-100); NULL weights == the plain entry points
 *
 * Data layout ("packed operand"): one rank's normalised embeddings are a dense row-major array
 *   X[2][bpad][Dpad]   (modality 0 = video rows, 1 = text rows), zero padded,
 * element type fp32 (CROSSCLR_MODE_FP32) or bf16 (CROSSCLR_MODE_BF16).  A column operand made of
 * `col_ranks` such arrays back to back (what an RCCL all-gather of packed operands produces) is
 *   Xcols[col_ranks][2][bpad][Dpad].
 * Per-row statistics use the same [2][bpad] (rows) / [col_ranks][2][bpad] (columns) indexing.
 */
#ifndef CROSSCLR_H
#define CROSSCLR_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CROSSCLR_LAUNCH_GROUPS 8   /* launch groups the forward workspace has room for */
#define CROSSCLR_ABI_VERSION 7

/* input element types (crossclr_normalize / crossclr_backward_finish) */
#define CROSSCLR_IN_F32 0
#define CROSSCLR_IN_F16 1
#define CROSSCLR_IN_BF16 2
#define CROSSCLR_IN_F64 3

/* compute modes */
#define CROSSCLR_MODE_FP32 0 /* v_mfma_f32_32x32x2_f32: exact fp32 products            */
#define CROSSCLR_MODE_BF16 1 /* v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulate */

/* error codes */
#define CROSSCLR_OK 0
#define CROSSCLR_E_ARG (-1)      /* bad shape / pointer / enum                       */
#define CROSSCLR_E_RANGE (-2)    /* temperature too small for the fixed-shift softmax: use the two-pass entry points */
#define CROSSCLR_E_HIP (-3)      /* a HIP call failed (text in crossclr_last_error)   */
#define CROSSCLR_E_WORKSPACE (-4)/* workspace too small                               */

typedef struct crossclr_plan {
    /* inputs */
    int b;          /* valid rows per modality on this rank                 */
    int D;          /* embedding dimension                                  */
    int world;      /* ranks sharing the global batch (1 = single GPU)      */
    int rank;       /* this rank                                            */
    int mode;       /* CROSSCLR_MODE_*                                      */
    /* derived by crossclr_make_plan */
    int bpad;       /* b rounded up to 128                                  */
    int Dpad;       /* D rounded up to what the selected kernels need       */
    int fast_path;  /* forward: 1 register-resident bf16 kernel, 0 generic tiled */
    int fast_bwd;   /* backward: 0 generic tiled, 1 32-row-wave kernel, 2 16-row-wave kernel */
    int fwd_blocks; /* persistent thread blocks of the fast forward (0: generic grid) */
    int fwd_slots;  /* partial-sum slots one crossclr_forward launch owns */
    size_t fwd_ws_floats; /* floats in the forward workspace `part` (slots of two launches + column sums + flag) */
    int bwd_slices; /* gradient slices crossclr_backward writes (summed by _finish) */
    int loss_ws_doubles; /* doubles in the loss_sum buffer of crossclr_forward_finish: [0] = result */
    size_t operand_bytes;   /* one packed operand X[2][bpad][Dpad]          */
    size_t gbuf_bytes;      /* fp32 d(loss)/d(xhat) accumulator [bwd_slices][2][bpad][Dpad] */
    size_t stash_bytes;     /* ABI 3: bytes of the saved-exponentials buffer of crossclr_forward_save (0: not available
                               for this plan -- use crossclr_forward / crossclr_backward, which recompute) */
    size_t xf_bytes;        /* ABI 4: bytes of the fragment-major copy of the packed operand that crossclr_normalize_xf / crossclr_pack_xf
                               write and crossclr_backward_saved_xf / _xfp read (0: not available -- fp32 plans, bf16 plans beyond D = 8192) */
} crossclr_plan;

int crossclr_abi_version(void);
const char* crossclr_last_error(void);
/* "hip-gfx950" for the product library, "emu-host" for the test-only SIMT emulation build */
const char* crossclr_backend(void);

int crossclr_make_plan(int b, int D, int world, int rank, int mode, crossclr_plan* plan);

/* Row L2-normalisation of both modalities into a packed operand (loss.py:79-80).
 * inv_norm[2][bpad] = 1/max(||x||,1e-12); diag_cos[bpad] = vhat_i . that_i in fp32.            */
int crossclr_normalize(const crossclr_plan* plan, const void* video, const void* text,
                       long ld_video, long ld_text, int in_dtype,
                       void* xhat, float* inv_norm, float* diag_cos, void* stream);

/* Shifted soft-max denominators of the plan's rows against the given columns, into the forward
 * workspace `part` (plan->fwd_ws_floats floats; up to CROSSCLR_LAUNCH_GROUPS (8) launch groups per step: slot0 = L * fwd_slots,
 * L = 0..7 -- the plain sharded scheme uses two):
 *   part[slot][2*bpad] = sum over the slot's columns q of exp(s(p,q) * xhat_p . xhat_q / tau - shift)
 * with s = 1 across modalities, negative_weight inside a modality, the intra-modal self pair
 * skipped (its exp(0) = 1 is added by crossclr_forward_finish).  Writes `plan->fwd_slots` slots
 * starting at slot0.  col_ranks/col_rank0 describe the column operand; skip_rank (or -1) lets a
 * second launch over the all-gathered operand skip the rank already covered by a local launch.
 * When xhat_rows == xhat_cols (local block, slot0 = 0) the bf16 fast path exploits the symmetry of
 * the stacked matrix (upper triangle only; the mirrored contributions arrive as column sums in the
 * workspace) -- transparent to the caller, crossclr_forward_finish reads a flag from the workspace. */
int crossclr_forward(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols,
                     int col_ranks, int col_rank0, int skip_rank,
                     float temperature, float negative_weight,
                     float* part, int slot0, void* stream);

/* Reduce `nslots` (= fwd_slots times the number of launch groups used, 1..8) partial slots: logz[2][bpad] (natural log of the full denominator),
 * rz = 1/Z_shifted, wrz = negative_weight * rz (both 0 on padding rows) and
 * loss_sum[0] = sum over valid rows of (logZv + logZt - 2 A_ii)  (double); loss_sum must hold
 * plan->loss_ws_doubles doubles ([1..] are per-block partials, added in index order; afterwards
 * loss_sum[1] = loss_sum[0] / (2 * b * world): this rank's share of the mean loss of loss.py:114).  */
int crossclr_forward_finish(const crossclr_plan* plan, const float* part, int nslots,
                            const float* diag_cos, float temperature, float negative_weight,
                            float* logz, float* rz, float* wrz, double* loss_sum, void* stream);

/* gbuf[slice][2][bpad][Dpad] (+)= sum over the slice's columns q of s E(p,q) (rz_p + rz_q) xhat_q
 * (unscaled); the slices are disjoint column ranges, summed by crossclr_backward_finish. */
int crossclr_backward(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols,
                      int col_ranks, int col_rank0, int skip_rank,
                      float temperature, float negative_weight,
                      const float* rz_rows, const float* wrz_rows,
                      const float* rz_cols, const float* wrz_cols,
                      float* gbuf, int accumulate, void* stream);

/* grad_x = normalize-backward( gbuf/(2 B tau) - partner_hat/(B tau) ) * grad_out[0], written in
 * the input dtype.  B = plan->b * plan->world.  grad_out is a DEVICE pointer to one double.     */
int crossclr_backward_finish(const crossclr_plan* plan, const float* gbuf,
                             const void* video, const void* text, long ld_video, long ld_text,
                             int in_dtype, const float* inv_norm, float temperature,
                             const double* grad_out, void* grad_video, void* grad_text,
                             long ld_gvideo, long ld_gtext, void* stream);

/* ---- per-sample weights (ABI version 2) ------------------------------------------------------------
 * k >= 0  "negative scale": multiplier of exp(logit) wherever the sample is an INTRA-modal negative column
 *         (0 = pruned from the negative set, 1 = reference); its masked self pair (logit 0, loss.py:96-97)
 *         travels with the column:   Z_p = sum_inter E + sum_{q != p, same modality} k_q E_pq + k_p e^0
 * omega   weight of the sample's own loss term, normalised so that omega == 1 is the reference's mean:
 *         loss = sum_p omega_p (log Z_p - A_pp) / (2 B)
 * Both are constants w.r.t. the embeddings.  Arrays use the [2][bpad] / [col_ranks][2][bpad] indexing of
 * the statistics; padding entries are ignored.  Any pointer (or the struct) may be NULL = all ones.      */
typedef struct crossclr_sample_weights {
    const float* neg_scale_rows; /* k of this rank's rows            [2][bpad]            */
    const float* neg_scale_cols; /* k of the column operand's rows   [col_ranks][2][bpad] */
    const float* loss_weight;    /* omega of this rank's rows        [2][bpad]            */
} crossclr_sample_weights;

/* uses neg_scale_cols (and neg_scale_rows for the mirrored tiles of the symmetric evaluation) */
int crossclr_forward_w(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols,
                       int col_ranks, int col_rank0, int skip_rank,
                       float temperature, float negative_weight, const crossclr_sample_weights* sw,
                       float* part, int slot0, void* stream);
/* uses neg_scale_rows (self pair) and loss_weight:  rz = omega/Z, wrz = negative_weight*omega/Z,
 * loss_sum[0] = sum_p omega_p log Z_p - sum_i (omega_v,i + omega_t,i) A_ii                          */
int crossclr_forward_finish_w(const crossclr_plan* plan, const float* part, int nslots,
                              const float* diag_cos, float temperature, float negative_weight,
                              const crossclr_sample_weights* sw,
                              float* logz, float* rz, float* wrz, double* loss_sum, void* stream);
/* intra-modal weight becomes  s E (wrz_p k_q + wrz_q k_p);  uses neg_scale_rows and neg_scale_cols */
int crossclr_backward_w(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols,
                        int col_ranks, int col_rank0, int skip_rank,
                        float temperature, float negative_weight,
                        const float* rz_rows, const float* wrz_rows,
                        const float* rz_cols, const float* wrz_cols,
                        const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream);
/* positive-pair term becomes -(omega_v,i + omega_t,i)/(2 B tau) partner_hat;  uses loss_weight      */
int crossclr_backward_finish_w(const crossclr_plan* plan, const float* gbuf,
                               const void* video, const void* text, long ld_video, long ld_text,
                               int in_dtype, const float* inv_norm, float temperature,
                               const crossclr_sample_weights* sw,
                               const double* grad_out, void* grad_video, void* grad_text,
                               long ld_gvideo, long ld_gtext, void* stream);

/* ---- pair evaluation for sharded runs (ABI version 2) -----------------------------------------------------
 * Ranks r and s both need the (r, s) block of the stacked matrix of exponentials; it is symmetric, so it is enough
 * that ONE of them evaluates it: crossclr_forward_pairs evaluates this rank's rows against ranks
 * first_rank, first_rank+1, ... (mod plan->world; `nranks` of them) of the WHOLE gathered operand `xhat_all`
 * [world][2][bpad][Dpad], writes this rank's partial row sums to the slots at slot0 like crossclr_forward, AND
 * colsum_out[nranks][2][bpad] = for each of those ranks' rows the sum over THIS rank's rows -- the partial row sums
 * that rank needs from this block.  The caller ships colsum_out[i] to rank first_rank+i and feeds what it receives
 * (summed) back with crossclr_forward_add, which makes a launch group out of one ready-made slot (vec == NULL: an empty
 * group).  A step may use up to eight launch groups (slot0 = L * fwd_slots, L = 0..7: local block, one per pair partner or
 * one for the whole pair range, antipodal rank, received sums; crossclr_forward_finish:
 * nslots = 4 * fwd_slots then).  bf16 register-resident path only (plan->fast_path); sample weights: sw->neg_scale_cols
 * is indexed like xhat_all.                                                                                   */
int crossclr_forward_pairs(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_all,
                           int first_rank, int nranks, float temperature, float negative_weight,
                           const crossclr_sample_weights* sw, float* part, int slot0, float* colsum_out, void* stream);
int crossclr_forward_add(const crossclr_plan* plan, float* part, int slot0, const float* vec, void* stream);

/* ---- save-for-backward pair (ABI version 3) ------------------------------------------------------------------
 * What autograd's "saved tensors" are for the reference (it keeps every fp64 [B,2B] intermediate of loss.py:96-112
 * alive for the backward: 12.5 GB at B = 8192), reduced to its minimum: crossclr_forward_save is crossclr_forward_w for
 * the LOCAL symmetric block (rows = columns = this rank's packed operand) that ALSO writes the bf16 exponentials
 * E = exp(logit - shift) of the upper triangle of the stacked 2b x 2b matrix to `stash` (plan->stash_bytes bytes,
 * 0.27 GB at b = 8192; caller-owned), and crossclr_backward_saved is crossclr_backward_w for the same block fed from
 * that stash instead of recomputing the similarity product: it executes the algorithmic 8 b^2 D flop instead of
 * 16 b^2 D.  Available when plan->stash_bytes > 0: the bf16 register-resident path (Dpad <= 1024; bf16 exponentials, upper
 * triangle, 2 KiB per 32 x 32 tile) and CROSSCLR_MODE_FP32 plans (the fp32 exponentials of the whole stacked matrix, 4 KiB per
 * 32 x 32 fragment in the forward's register layout: (2 bpad)^2 * 4 bytes, offered up to 16 GiB).  The layout is private to the
 * pair; `stash` must reach the backward unmodified.  sw->neg_scale_rows (== the columns') and rz/wrz are this rank's
 * [2][bpad] arrays.                                                                                                    */
int crossclr_forward_save(const crossclr_plan* plan, const void* xhat, float temperature, float negative_weight,
                          const crossclr_sample_weights* sw, float* part, int slot0, void* stash, void* stream);
int crossclr_backward_saved(const crossclr_plan* plan, const void* xhat, const void* stash,
                            float temperature, float negative_weight, const float* rz, const float* wrz,
                            const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream);
/* ABI 4 -- the fragment-major operand (plan->xf_bytes > 0: bf16 register-resident path, Dpad <= 1024, local block).
 * crossclr_normalize_xf / crossclr_pack_xf are crossclr_normalize (loss.py:79-80) / crossclr_pack that ALSO write the bf16 unit rows
 * a second time, to `xhat_xf` (plan->xf_bytes bytes, caller-owned), laid out as the MFMA B fragments of the gradient product
 *   xhat_xf[stacked row / 32][column / 32][k-step][lane][8]   (1 KiB per fragment; crossclr_kernels_generic.h: normalize_xf_kernel)
 * and crossclr_backward_saved_xf is crossclr_backward_saved (autograd of loss.py:83-112 from the saved exponentials) that loads those
 * fragments straight into registers instead of staging the row-major column tile through LDS: same arguments otherwise, and the SAME
 * RESULT BITS (every accumulator receives the same MFMA sequence; tests/test_gpu_xf.py, tools/soak_xf.py).  Faster from D = 512 up
 * (DESIGN.md 3.7), slower below; the row-major `xhat` is still what every forward and every cross-rank block reads.                  */
int crossclr_normalize_xf(const crossclr_plan* plan, const void* video, const void* text, long ld_video, long ld_text,
                          int in_dtype, void* xhat, void* xhat_xf, float* inv_norm, float* diag_cos, void* stream);
int crossclr_pack_xf(const crossclr_plan* plan, const void* video_hat, const void* text_hat, long ld_video, long ld_text,
                     int in_dtype, void* xhat, void* xhat_xf, float* inv_norm, float* diag_cos, void* stream);
int crossclr_backward_saved_xf(const crossclr_plan* plan, const void* xhat_xf, const void* stash,
                               float temperature, float negative_weight, const float* rz, const float* wrz,
                               const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream);
/* The same backward -- same arguments, the same bits in `gbuf` -- with TWO 32-column tiles per barrier interval
 * (fast_bwd_xfp_kernel: one barrier, one closing wait and one set of cursor updates per 64 MFMAs; 32-bit scalar offsets into the
 * saved exponentials, hence plan->stash_bytes < 4 GiB -- CROSSCLR_E_ARG otherwise: take crossclr_backward_saved_xf).               */
int crossclr_backward_saved_xfp(const crossclr_plan* plan, const void* xhat_xf, const void* stash,
                               float temperature, float negative_weight, const float* rz, const float* wrz,
                               const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream);
/* Two-pass regime, save-for-backward pair (exact-fp32 plans, local block; ABI 3): the second pass also leaves
 * U[p][q] = exp2(x_pq - shift_p) and Ut[p][q] = U[q][p] behind (crossclr_stash_bytes_s = twice plan->stash_bytes, 0 = not
 * available), and the backward forms the weights U rz_p + Ut rz_q from them instead of recomputing the similarity product
 * (replaces autograd of loss.py:96-100, 59-60 like crossclr_backward_saved).  `shift` = the row maxima of crossclr_forward_rowmax.
 * bf16 register-resident plans (round 4): the same three entry points; the stash is the FULL matrix of bf16 records U[p][q] in the
 * rectangular layout of a one-rank remote block followed by 2 bpad floats of zeros (crossclr_stash_bytes_s says how much), and the backward
 * is two launches of the saved D-slice kernel -- direct with the row statistics, transposed with the contracted rows' (accumulating).     */
size_t crossclr_stash_bytes_s(const crossclr_plan* plan);
int crossclr_forward_save_s(const crossclr_plan* plan, const void* xhat, float temperature, float negative_weight,
                            const crossclr_sample_weights* sw, const float* shift, float* part, int slot0, void* stash, void* stream);
int crossclr_backward_saved_s(const crossclr_plan* plan, const void* xhat, const void* stash, float temperature,
                              float negative_weight, const float* rz, const float* wrz, const crossclr_sample_weights* sw,
                              float* gbuf, int accumulate, void* stream);


/* ---- sharded step: rectangular blocks with saved exponentials (ABI version 3) -------------------------------------------
 * In a sharded run this rank evaluates, besides its local block, the blocks of its pair partners (ranks rank+1 ..
 * rank+(world-1)/2, whose column sums it ships to them: crossclr_forward_pairs) and of the antipodal rank.  With a backward
 * to follow those launches can save their exponentials too: crossclr_forward_rect_save is crossclr_forward_pairs
 * (with_colsums = 1) or a plain rectangular launch over `nranks` ranks from `first_rank` (with_colsums = 0; modulo plan->world,
 * columns taken from the whole gathered operand xhat_all [world][2][bpad][Dpad]) that also fills `stash`
 * (crossclr_rect_stash_bytes(plan, nranks) bytes: 2 KiB per 32 x 32 tile, 0.54 GB per rank at b = 8192), and
 * crossclr_backward_rect_saved consumes it: gbuf += W . xhat over those ranks' columns without recomputing the similarity
 * product (rz_all / wrz_all / sw->neg_scale_cols: the gathered [world][2][bpad] statistics).  The blocks OTHER ranks
 * evaluated (ranks rank-1 .. rank-(world-1)/2) are not held here: crossclr_backward_ranks recomputes them
 * (crossclr_backward_w over a wrapping rank range).
 * Exact-fp32 plans (round 4): the same three entry points with with_colsums = 0 -- the generic forward over the rank range leaves its fp32
 * fragments (4 KiB per 32 x 32 fragment: 1 GiB per rank at b = 8192, up to 16 GiB; crossclr_rect_stash_bytes says 0 beyond) and
 * crossclr_backward_rect_saved is the gradient product alone (bwd_saved32_kernel<..., RECT>).
 * Wide bf16 plans (1024 < D <= 8192; round 4, extended from 4096 in round 6): the same three entry points with with_colsums = 0 as well -- the generic forward over the rank
 * range leaves bf16 records in the rectangular layout (2 KiB per 32 x 32 tile: 0.5 GiB per rank at b = 8192) and the backward is the
 * D-slice kernel in column parts (fast_bwd_dsl_kernel<..., MODE 1, XP, 4>): 13.9 ms of recompute per 8192 x 8192 block at D = 1536 gone.   */
size_t crossclr_rect_stash_bytes(const crossclr_plan* plan, int nranks);
int crossclr_forward_rect_save(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_all,
                               int first_rank, int nranks, int with_colsums, float temperature, float negative_weight,
                               const crossclr_sample_weights* sw, float* part, int slot0, float* colsum_out,
                               void* stash, void* stream);
int crossclr_backward_rect_saved(const crossclr_plan* plan, const void* xhat_all, const void* stash,
                                 int first_rank, int nranks, float temperature, float negative_weight,
                                 const float* rz_rows, const float* wrz_rows, const float* rz_all, const float* wrz_all,
                                 const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream);
/* Two-pass regime (round 4, ABI version 5), exact-fp32 plans: the block against other ranks' columns saves U[p][q] = exp2(x - shift_rows[p]) and,
 * behind it, Ut[p][q] = exp2(x - shift_all[q]) (shift_all = the gathered [world][2][bpad] row maxima: the ranks exchange them between
 * the two passes; twice crossclr_rect_stash_bytes, up to 32 GiB), and crossclr_backward_rect_saved_s forms
 * W[p][q] = U rz_p + Ut rz_q from them (bwd_saved32_kernel<..., RM, RECT>): no remote block of an exact-fp32 run is recomputed at any
 * temperature.  bf16 register-resident plans (D <= 1024): two arrays of bf16 records (U, Ut: 2 x crossclr_rect_stash_bytes) followed by
 * (world + 1) x 2 bpad floats of zeros, and the backward is two rectangular launches of the saved D-slice kernel (U with the rows'
 * statistics, Ut with the columns').  Wide bf16 plans (D > 1024): the same -- and their LOCAL block's pair (crossclr_forward_save_s /
 * crossclr_backward_saved_s) works that way too: crossclr_stash_bytes_s = 2 x records (U, Ut) + 2 bpad zeros, two direct launches.       */
size_t crossclr_rect_stash_bytes_s(const crossclr_plan* plan, int nranks);
int crossclr_forward_rect_save_s(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_all, int first_rank, int nranks,
                                 float temperature, float negative_weight, const crossclr_sample_weights* sw,
                                 const float* shift_rows, const float* shift_all, float* part, int slot0, void* stash, void* stream);
int crossclr_backward_rect_saved_s(const crossclr_plan* plan, const void* xhat_all, const void* stash, int first_rank, int nranks,
                                   float temperature, float negative_weight, const float* rz_rows, const float* wrz_rows,
                                   const float* rz_all, const float* wrz_all, const crossclr_sample_weights* sw, float* gbuf,
                                   int accumulate, void* stream);
/* The other half of a pair block (ABI version 3): the rank that evaluated block (r, s) in the forward holds its exponentials, so it
 * can also form what that block contributes to rank s's gradient -- sum over r's rows p of W[p][q] xhat_r[p] for every row q of s: the
 * TRANSPOSE of the block, read from the same stash (8 b^2 D flop) -- instead of rank s recomputing the block (16 b^2 D).
 * `stash` = the rectangular stash of crossclr_forward_rect_save(first_rank, nranks), `which` = the partner's position in that range;
 * xhat_rows / rz_rows / wrz_rows = this rank's own operand and statistics, rz_all / wrz_all = the gathered statistics;
 * gpartner = [plan->bwd_slices][2 bpad][Dpad] fp32 (plan->gbuf_bytes), overwritten: the partner adds the sum of the slices to its own
 * gradient buffer before crossclr_backward_finish.                                                                              */
int crossclr_backward_rect_saved_t(const crossclr_plan* plan, const void* xhat_rows, const void* stash, int first_rank, int nranks,
                                   int which, float temperature, float negative_weight, const float* rz_rows, const float* wrz_rows,
                                   const float* rz_all, const float* wrz_all, const crossclr_sample_weights* sw, float* gpartner,
                                   void* stream);
/* ---- remote blocks on the fragment-major operand (ABI version 5) --------------------------------------------------------------
 * crossclr_pack_xf_from_packed: the fragment-major copy (layout: crossclr_normalize_xf) of `nranks` consecutive PACKED operands --
 * what a rank does with the slices of the gathered operand it received (the exchange moves the row-major operand only).
 * crossclr_backward_rect_saved_xfp / _t_xfp: crossclr_backward_rect_saved / _t with the pair kernel (fast_bwd_xfp_kernel: column tiles
 * as MFMA B fragments straight from that copy, two tiles per barrier interval); same arguments except the operand -- the copy of the
 * WHOLE gathered array (only the slices of first_rank .. first_rank+nranks-1 are read) resp. of this rank's own operand -- and the same
 * gradients within the summation order of the slices (the pair kernel cuts even slices).  CROSSCLR_E_ARG for operands / stashes of
 * 4 GiB or more (32-bit offsets): take the LDS-staged entry points then.                                                         */
int crossclr_pack_xf_from_packed(const crossclr_plan* plan, const void* xhat_packed, int nranks, void* xhat_xf, void* stream);
int crossclr_backward_rect_saved_xfp(const crossclr_plan* plan, const void* xf_all, const void* stash, int first_rank, int nranks,
                                     float temperature, float negative_weight, const float* rz_rows, const float* wrz_rows,
                                     const float* rz_all, const float* wrz_all, const crossclr_sample_weights* sw, float* gbuf,
                                     int accumulate, void* stream);
int crossclr_backward_rect_saved_t_xfp(const crossclr_plan* plan, const void* xf_rows, const void* stash, int first_rank, int nranks,
                                       int which, float temperature, float negative_weight, const float* rz_rows, const float* wrz_rows,
                                       const float* rz_all, const float* wrz_all, const crossclr_sample_weights* sw, float* gpartner,
                                       void* stream);

int crossclr_backward_ranks(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_all,
                            int first_rank, int nranks, float temperature, float negative_weight,
                            const float* rz_rows, const float* wrz_rows, const float* rz_all, const float* wrz_all,
                            const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream);

/* ---- caller-side fusion: embeddings that are already unit vectors (ABI version 3) --------------
 * When the producer (a projection head with a fused L2-norm epilogue) hands over unit rows, the normalisation of
 * loss.py:79-80 is not repeated: crossclr_pack only casts / lays the rows out as the packed operand (inv_norm := 1) and forms
 * the fp32 positive-pair cosine; crossclr_backward_finish_p with prenormalized = 1 returns d(loss)/d(the unit rows as given)
 * -- no projection, no 1/||x|| -- so that the producer's own normalise-backward (autograd of F.normalize upstream) applies.
 * prenormalized = 2: the rows given are the unit vectors (e.g. the packed bf16 operand itself: video = X[0], text = X[1], ld = Dpad,
 * in_dtype bf16) and inv_norm holds 1 / ||y|| of the vectors they came from: returns d(loss)/dy = (G - xhat (xhat . G)) / ||y|| in the rows'
 * dtype -- the normalise-backward of a fused projection head without a separate pass (crossclr_project_pack below).
 * A producer that writes X[2][bpad][Dpad] itself in the plan's element type skips crossclr_pack's copy and only needs
 * diag_cos[i] = vhat_i . that_i.                                                                                       */
int crossclr_pack(const crossclr_plan* plan, const void* video_hat, const void* text_hat, long ld_video, long ld_text,
                  int in_dtype, void* xhat, float* inv_norm, float* diag_cos, void* stream);
int crossclr_backward_finish_p(const crossclr_plan* plan, const float* gbuf,
                               const void* video, const void* text, long ld_video, long ld_text,
                               int in_dtype, const float* inv_norm, float temperature,
                               const crossclr_sample_weights* sw,
                               const double* grad_out, void* grad_video, void* grad_text,
                               long ld_gvideo, long ld_gtext, int prenormalized, void* stream);

/* ---- producer-side fusion: projection head + L2-norm + pack in one launch (bf16 plans, Dpad <= 1024) ----
 * /root/reference/README.md:24-38 feeds the criterion "features: [bsz, f_dim]" that a projection layer produced; this entry point IS
 * that layer's forward fused with trainer/loss.py:79-80:   y_m = x_m W_m^T + b_m;   xhat_m = y_m / max(||y_m||, 1e-12)
 *   x_video / x_text  [b, Din_video] / [b, Din_text] row-major (row strides in elements), any supported in_dtype
 *   w_video / w_text  bf16 [D, ldw_*] row-major = torch.nn.Linear.weight layout, columns zero-padded to ldw_* = a multiple of 64 >= Din_*
 *   bias_*            fp32 [D] or NULL
 * writes the packed operand, inv_norm[2][bpad] (= 1 / ||y||) and the fp32 positive-pair cosines diag_cos[bpad] exactly as
 * crossclr_normalize does -- everything downstream (forward, backward, crossclr_backward_finish_p with prenormalized = 1) is unchanged.
 * Backward: crossclr_backward_finish_p(prenormalized = 2) on the packed operand returns g_y directly (bf16); the older two-step form --
 * crossclr_project_backward_prep turns the gradient w.r.t. the unit rows (what crossclr_backward_finish_p(prenormalized = 1) returns)
 * into the gradient w.r.t. y:  g_y = (G - xhat (xhat . G)) / ||y||, fp32 [b, D] per modality; the projection's own backward
 * (dW = g_y^T x, dx = g_y W, db = column sums) is two plain GEMMs on the caller's side.                                          */
int crossclr_project_pack(const crossclr_plan* plan, const void* x_video, const void* x_text, long ld_video, long ld_text, int Din_video,
                          int Din_text, int in_dtype, const void* w_video, const void* w_text, int ldw_video, int ldw_text, const float* bias_video,
                          const float* bias_text, void* xhat, float* inv_norm, float* diag_cos, void* stream);
/* The same launch with the weights FRAGMENT-MAJOR (what the host module passes): wf_* = bf16 [Dpad / 32][ldw / 16][64][8], lane
 * (l31 = lane & 31, half = lane >> 5) of record (d32, ks) holding W[32 d32 + l31][16 ks + 8 half .. + 7], rows beyond D and columns beyond
 * Din zero -- a wave's MFMA B fragment is then ONE coalesced 1-KiB load instead of 64 scattered 16-byte pieces (3x on the MI355X).       */
int crossclr_project_pack_wf(const crossclr_plan* plan, const void* x_video, const void* x_text, long ld_video, long ld_text, int Din_video,
                             int Din_text, int in_dtype, const void* wf_video, const void* wf_text, int ldw_video, int ldw_text,
                             const float* bias_video, const float* bias_text, void* xhat, float* inv_norm, float* diag_cos, void* stream);
/* The projection's weight gradient (round 4): dW_m[D][Din_m] = g_y,m^T x_m (fp32, row-major = torch.nn.Linear.weight.grad layout) and
 * db_m = column sums of g_y,m (NULL: not wanted) for both modalities -- one split-K MFMA launch + one reduce, deterministic.
 *   gy_*  bf16 [b, D] (row stride ld_gy elements): what crossclr_backward_finish_p(prenormalized = 2) wrote
 *   x_*   the projection's inputs [b, Din_*] in `in_dtype` (the same tensors crossclr_project_pack read)
 *   ws    crossclr_project_dw_ws_floats(b, D, Din_video, Din_text) floats of scratch (caller-owned)                                  */
size_t crossclr_project_dw_ws_floats(int b, int D, int Din_video, int Din_text);
int crossclr_project_dw(int b, int D, const void* gy_video, const void* gy_text, long ld_gy, const void* x_video, const void* x_text,
                        long ld_xv, long ld_xt, int Din_video, int Din_text, int in_dtype, float* ws, float* dw_video, float* dw_text,
                        long ld_dwv, long ld_dwt, float* db_video, float* db_text, void* stream);
int crossclr_project_backward_prep(const crossclr_plan* plan, const float* g_video, const float* g_text, long ld_gv, long ld_gt,
                                   const void* xhat, const float* inv_norm, float* gy_video, float* gy_text, long ld_out,
                                   void* stream);

/* ---- two-pass soft-max for small temperatures (ABI version 3) ------------------------------------------------------
 * The reference's soft-max runs in float64 with a per-row maximum (loss.py:60 after the promotion at :96-100), so it is
 * finite for any temperature; the single common shift of the entry points above covers max |logit| =
 * max(1, |negative_weight|) / temperature <= 128 (they return CROSSCLR_E_RANGE beyond it).  crossclr_needs_row_shift says when
 * that limit is exceeded; then
 *   1. crossclr_forward_rowmax: shift_rows[2][bpad] = the rows' maxima of the scaled logits (log2 domain) over the given
 *      columns and -- unless k_p = 0 -- the masked self pair's logit 0; `accumulate` = 1 folds a further column operand into an
 *      existing result (remote ranks' columns); `part` is scratch (the slots of one launch group);
 *   2. crossclr_forward_s / crossclr_forward_finish_s: the sums relative to the row's own shift; rz / wrz come out relative to
 *      that shift too (values in [1/(2B), 1]);
 *   3. crossclr_backward_s with the shifts of the rows and of the columns (the statistics layout): the weight becomes
 *      s (exp2(x - shift_p) rz_p k_q + exp2(x - shift_q) rz_q k_p), two exponentials <= 1.
 * These run on the generic tiled kernels in both compute modes (the register-resident kernels and the save-for-backward
 * pair assume the common shift).  With shift pointers NULL the _s entry points are the _w ones.                         */
int crossclr_needs_row_shift(float temperature, float negative_weight);
int crossclr_forward_rowmax(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols,
                            int col_ranks, int col_rank0, int skip_rank, float temperature, float negative_weight,
                            const crossclr_sample_weights* sw, float* part, float* shift_rows, int accumulate, void* stream);
int crossclr_forward_s(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols,
                       int col_ranks, int col_rank0, int skip_rank, float temperature, float negative_weight,
                       const crossclr_sample_weights* sw, const float* shift_rows, float* part, int slot0, void* stream);
int crossclr_forward_finish_s(const crossclr_plan* plan, const float* part, int nslots,
                              const float* diag_cos, float temperature, float negative_weight,
                              const crossclr_sample_weights* sw, const float* shift_rows,
                              float* logz, float* rz, float* wrz, double* loss_sum, void* stream);
int crossclr_backward_s(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols,
                        int col_ranks, int col_rank0, int skip_rank, float temperature, float negative_weight,
                        const float* rz_rows, const float* wrz_rows, const float* rz_cols, const float* wrz_cols,
                        const crossclr_sample_weights* sw, const float* shift_rows, const float* shift_cols,
                        float* gbuf, int accumulate, void* stream);

/* ---- influential-sample statistics (ABI version 2; SURVEY.md 8(f) rank 1, not in the reference @ v1) ----
 * From INPUT-space features x[b][Din] of both modalities (any float dtype, row stride ld):
 *   conn_i = mean_j xhat_i . xhat_j (self pair masked) = (xhat_i . sum_j xhat_j - xhat_i . xhat_i) / B     O(B Din)
 *   neg_scale_i = conn_i / max(conn) < score_threshold          (1 keeps the sample among the negatives, 0 prunes it)
 *   loss_weight = B rho / sum(rho),  rho = exp(conn / sum(conn) / temperature_weights)
 * Three calls so that a sharded caller can all-reduce `colsum` and all-gather `conn` in between:
 *   _colsum: inv_norm[2][b] = 1/max(||x||,1e-12), colsum[2][Din] = sum of this rank's normalised rows (double);
 *            partial_ws = 2*CROSSCLR_INFL_BLOCKS*Din floats of scratch
 *   _conn:   conn[2][b] (double) from colsum summed over all ranks and B_global
 *   _finish: conn_all[world][2][b] -> neg_scale[2][bpad], loss_weight[2][bpad] of plan->rank's rows (padding 0),
 *            ready to be used as crossclr_sample_weights                                                      */
#define CROSSCLR_INFL_BLOCKS 256
#define CROSSCLR_INFL_MAX_DIN 4096
int crossclr_influence_colsum(const void* x_video, const void* x_text, long ld_video, long ld_text, int in_dtype,
                              int b, int Din, float* inv_norm, float* partial_ws, double* colsum, void* stream);
int crossclr_influence_conn(const void* x_video, const void* x_text, long ld_video, long ld_text, int in_dtype,
                            int b, int Din, const float* inv_norm, const double* colsum_total, int B_global,
                            double* conn, void* stream);
int crossclr_influence_finish(const crossclr_plan* plan, const double* conn_all, float score_threshold,
                              float temperature_weights, float* neg_scale, float* loss_weight, void* stream);

/* ---- score statistics of the inter-modal block (single device: plan->world == 1) -------------
 * The B x B matrix S = im . s^T of /root/reference/trainer/loss.py:30 (`cosine_sim`, loss.py:7-15: a plain product; the rows
 * are used as given -- lay them out with crossclr_pack, or crossclr_normalize for cosines of raw features), never materialised:
 *   crossclr_score_diag    diag[2][bpad]: the positive pairs' scores S_ii (loss.py:31), once per stacked row, from the same
 *                          MFMA sequence crossclr_score_rows compares against (S_ij > S_ii is exact; no tie with oneself)
 *   crossclr_score_rows    per stacked row p (p < bpad: im_i against every s_j; p >= bpad: s_j against every im_i), over the
 *                          other modality's columns q != partner(p)                                   (loss.py:32-40):
 *                            hinge[p]  = sum_q max(0, margin + S_pq - S_pp)      (rows of cost_s / columns of cost_im)
 *                            active[p] = #{q : margin + S_pq - S_pp > 0}         (as float; exact below 2^24)
 *                          loss_sum[0] = sum_p hinge[p]; loss_sum[1] = loss_sum[0] / (B * B) = MaxMargin_coot.forward (loss.py:41).
 *                          margin = 0: active[p] is the retrieval rank of p's partner (0 = retrieved first): R@k = mean(active < k).
 *                          `part` = the plan's forward workspace (plan->fwd_ws_floats floats); loss_sum as for crossclr_forward_finish.
 *   crossclr_maxmargin_backward          gbuf = d(sum of hinges)/dS . X for every stacked row except the positive-pair terms
 *                                         (replaces autograd of loss.py:30-41; weight of a pair = its active hinges, 0..2)
 *   crossclr_maxmargin_backward_finish   column slices summed, positive-pair terms -(active_im_i + active_s_i) * partner_i,
 *                                         / (B * B), x grad_out, gradients in the input dtype.  `ones` = the inv_norm array
 *                                         crossclr_pack wrote (all ones); im / s = the tensors given to crossclr_pack.      */
int crossclr_score_diag(const crossclr_plan* plan, const void* xhat, float* diag, void* stream);
int crossclr_score_rows(const crossclr_plan* plan, const void* xhat, const float* diag, float margin, float* part,
                        float* hinge, float* active, double* loss_sum, void* stream);
int crossclr_maxmargin_backward(const crossclr_plan* plan, const void* xhat, const float* diag, float margin, float* gbuf,
                                void* stream);
/* ABI 7 -- the save-for-backward pair of the ranking loss: crossclr_score_rows_save is crossclr_score_rows (same bits) that also leaves,
 * per (row of the first modality, row of the second), the number of ACTIVE hinges of the pair -- 0, 1 or 2; 0 for the positive pair and
 * for padding -- as one byte: hinge_mask[bpad][bpad] (crossclr_maxmargin_mask_bytes; 64 MiB at b = 8192).  crossclr_maxmargin_backward_saved
 * forms gbuf (the layout and slices of crossclr_maxmargin_backward, same crossclr_maxmargin_backward_finish behind it) from that mask:
 * one product with the mask instead of evaluating the scores again (autograd of loss.py:30-41 keeps the two B x B hinge matrices for this).
 * crossclr_score_rows_save fails with CROSSCLR_E_ARG under CROSSCLR_DISABLE_SYMMETRIC (the one-pass evaluation writes the mask).        */
size_t crossclr_maxmargin_mask_bytes(const crossclr_plan* plan);
int crossclr_score_rows_save(const crossclr_plan* plan, const void* xhat, const float* diag, float margin, float* part,
                             float* hinge, float* active, double* loss_sum, void* hinge_mask, void* stream);
int crossclr_maxmargin_backward_saved(const crossclr_plan* plan, const void* xhat, const void* hinge_mask, float* gbuf, void* stream);
int crossclr_maxmargin_backward_finish(const crossclr_plan* plan, const float* gbuf, const void* im, const void* s, long ld_im,
                                       long ld_s, int in_dtype, const float* ones, const float* active, const double* grad_out,
                                       void* grad_im, void* grad_s, long ld_gim, long ld_gs, void* stream);

/* ---- THE WHOLE STEP BEHIND TWO CALLS (ABI 6; layout handed from call to call and split workspace: ABI 7.  Single device: plan->world == 1)
 * What the reference's one class is to its caller (trainer/loss.py:68-114 `forward`, and autograd's backward through it): a binding needs
 * nothing but crossclr_make_plan, crossclr_step_plan, crossclr_step_forward and crossclr_step_backward.  The library chooses the kernels:
 *   two-pass soft-max when max(1, |negative_weight|) / temperature > 128 (crossclr_needs_row_shift), the fixed-shift kernels otherwise;
 *   save-for-backward (the forward leaves its exponentials in the workspace, the backward is the gradient product alone) whenever the plan
 *     offers it, the caller did not forbid it (CROSSCLR_STEP_NO_SAVE / _FORWARD_ONLY), the stash is at most CROSSCLR_MAX_STASH_GB (default 8)
 *     and the workspace budget the caller named is large enough -- otherwise the recomputing pair;
 *   the fragment-major operand copy + the pair kernel (crossclr_backward_saved_xfp), the one-tile fragment-major kernel (_xf) or the
 *     LDS-staged saved backward, by padded width and batch (measured break-even: Dpad in {512, 768, 1024, wide plans} from 2048 / 4096 padded
 *     rows; CROSSCLR_XF_WIDTHS overrides) -- CROSSCLR_STEP_NO_XFP / _NO_XF take a hand-scheduled kernel out (a caller that verifies
 *     them on its device first, as this repository's Python module does, passes the verdict here).
 * crossclr_step_plan makes ALL of these decisions, once, and writes them into a crossclr_step_layout; crossclr_step_forward and
 * crossclr_step_backward act on the layout they are given and decide nothing themselves (they never read the environment), so the two
 * calls of one step cannot disagree: hand both the SAME layout, unmodified.
 *
 * The workspace is TWO caller-owned device regions (ABI 7), so that a caller can give most of it back early:
 *   `persistent` (layout.persistent_bytes): everything crossclr_step_backward reads.  With CROSSCLR_STEP_EAGER that is 1 / ||x|| and the
 *       gradient slices (67 MB at b = 8192, D = 512); otherwise the packed operand and the row statistics (17 MB).
 *   `transient`  (layout.transient_bytes): the rest -- saved exponentials (0.27 GB at that shape), fragment-major copy, partial sums.
 *       With CROSSCLR_STEP_EAGER it is dead as soon as crossclr_step_forward has returned (stream-ordered: free it on the launch
 *       stream); otherwise crossclr_step_backward reads it once more, and a backward that is given transient == NULL (a second backward
 *       through the same step after the caller released it) recomputes the similarity product from the persistent region instead.
 *   One allocation works too: transient = (char*)persistent + layout.persistent_bytes (offsets below are into that concatenation).
 * Other buffers (caller-owned, device): `loss_ws` (max(2, plan->loss_ws_doubles) doubles: loss_ws[1] = the mean loss of loss.py:114 after
 * the forward), `scratch` (layout.backward_scratch_bytes, backward only), the inputs and the gradients.  sw (optional): neg_scale_rows =
 * k[2][bpad], loss_weight = omega[2][bpad] (neg_scale_cols is ignored: the local block's columns are its rows).  Nothing is allocated,
 * nothing synchronises; errors as everywhere (CROSSCLR_E_WORKSPACE: workspace_bytes below even the recomputing layout; CROSSCLR_E_ARG: a
 * layout that crossclr_step_plan did not write for this plan).                                                                       */
#define CROSSCLR_STEP_NO_SAVE 1u        /* the backward recomputes the similarity product (smallest workspace)            */
#define CROSSCLR_STEP_FORWARD_ONLY 2u   /* no backward will follow (evaluation / no_grad): implies NO_SAVE                 */
#define CROSSCLR_STEP_PRENORMALIZED 4u  /* the rows are unit vectors already (loss.py:79-80 skipped; gradients w.r.t. the unit rows) */
#define CROSSCLR_STEP_NO_XFP 8u         /* do not take crossclr_backward_saved_xfp                                         */
#define CROSSCLR_STEP_NO_XF 16u         /* do not take crossclr_backward_saved_xf either (LDS-staged saved backward)       */
#define CROSSCLR_STEP_EAGER 32u         /* crossclr_step_forward ALSO enqueues the gradient product of the backward (it does not depend on grad_out: only the
                                           finish kernel scales by it) into a gbuf region of the persistent workspace; crossclr_step_backward is then the finish
                                           kernel alone.  Same kernels, same results; the GPU does not idle while the host walks from forward() to backward()
                                           (autograd's thread hand-over), a second backward through the same step re-runs the finish only, and the transient
                                           region can be released right after the forward call.  Cost: a forward whose backward() is never called has still
                                           paid for the gradient product. */
#define CROSSCLR_STEP_NONE ((size_t)-1) /* a layout offset that this step does not have                                    */

typedef struct crossclr_step_layout {
    size_t total_bytes;             /* persistent_bytes + transient_bytes                                              */
    size_t persistent_bytes;        /* region 1: what crossclr_step_backward reads (256-byte multiple; may be 0: FORWARD_ONLY) */
    size_t transient_bytes;         /* region 2: dead after the forward call (EAGER) / after the first backward call   */
    size_t backward_scratch_bytes;  /* bytes of crossclr_step_backward's `scratch` (the gradient slices, plan->gbuf_bytes; 0 with EAGER) */
    /* byte offsets into the concatenation [persistent | transient] (256-byte aligned; an offset >= persistent_bytes lives in `transient`
       at offset - persistent_bytes; CROSSCLR_STEP_NONE: not part of this step) */
    size_t xhat, inv_norm, diag, logz, rz, wrz, part, shift, xf, stash;
    size_t gbuf;                    /* CROSSCLR_STEP_EAGER: the gradient slices live in the persistent region (backward_scratch_bytes == 0)     */
    size_t ticket;                  /* one int: cleared by the step's first kernel, counts the finish kernel's blocks (its last block forms the loss) */
    size_t stash_bytes, xf_bytes;   /* sizes of xf and stash                                                           */
    float temperature, negative_weight;   /* what the layout was planned for (the step calls take them from here)     */
    unsigned flags;                 /* CROSSCLR_STEP_* the layout was planned for                                      */
    int two_pass;                   /* 1: per-row soft-max shifts (small temperature)                                  */
    int saved;                      /* 1: the forward saves its exponentials                                           */
    int backward_kernel;            /* 0 recomputing, 1 saved (LDS-staged / generic), 2 saved fragment-major, 3 saved fragment-major pair kernel */
    unsigned check;                 /* written by crossclr_step_plan over (plan, the fields above); the step calls refuse a layout whose check does not match */
} crossclr_step_layout;

/* workspace_bytes = 0: the best layout for these arguments (ask, allocate layout.persistent_bytes + layout.transient_bytes);
 * otherwise the best layout whose total_bytes fits into workspace_bytes.  The only step call that reads the environment
 * (CROSSCLR_MAX_STASH_GB, CROSSCLR_XF_WIDTHS, CROSSCLR_XFP).                                                           */
int crossclr_step_plan(const crossclr_plan* plan, float temperature, float negative_weight, unsigned flags,
                       size_t workspace_bytes, crossclr_step_layout* layout);
int crossclr_step_forward(const crossclr_plan* plan, const crossclr_step_layout* layout, const void* video, const void* text,
                          long ld_video, long ld_text, int in_dtype, const crossclr_sample_weights* sw,
                          void* persistent, void* transient, double* loss_ws, void* stream);
int crossclr_step_backward(const crossclr_plan* plan, const crossclr_step_layout* layout, const void* video, const void* text,
                           long ld_video, long ld_text, int in_dtype, const crossclr_sample_weights* sw,
                           void* persistent, void* transient, void* scratch, const double* grad_out,
                           void* grad_video, void* grad_text, long ld_gvideo, long ld_gtext, void* stream);

/* ---- second-order terms (ABI 7; single device, exact-fp32 products) ------------------------------------------------------
 * The reference's forward is a chain of eager PyTorch ops (trainer/loss.py:79-114), so autograd differentiates its backward again
 * (create_graph=True: gradient penalties, Hessian-vector products).  This entry point is that double backward: with
 * g = grad_out * dL/d(rows) what the first backward returned and (u_video, u_text) the cotangent handed in for g,
 *     h_video, h_text = d<u, g>/d(video, text) = grad_out * (Hessian of the loss) . u       (input dtype, like the gradients)
 *     d_grad_out[0]   = d<u, g>/d(grad_out)    = <u, dL/d(rows)>                            (one double, device)
 * computed in closed form on the device (csrc/crossclr_kernels_hvp.h: the directional derivative of SURVEY.md 3.5's gradient along the
 * tangent of the unit rows; two passes of a tiled kernel on the generic backward's skeleton + row kernels) -- no B x B tensor.
 * `plan` must be a CROSSCLR_MODE_FP32 plan (whatever mode the first-order step ran in: the reference differentiates float64 logits here);
 * the two-pass soft-max regime, per-sample weights (sw: neg_scale_rows = k[2][bpad], loss_weight = omega[2][bpad]) and prenormalized = 1
 * (rows are unit vectors as given, derivatives w.r.t. them as given) are covered.  `workspace`: crossclr_second_order_workspace_bytes(plan)
 * bytes, caller-owned, contents irrelevant before and after.  Nothing is allocated, nothing synchronises.                              */
size_t crossclr_second_order_workspace_bytes(const crossclr_plan* plan);
int crossclr_second_order(const crossclr_plan* plan, const void* video, const void* text, long ld_video, long ld_text, int in_dtype,
                          float temperature, float negative_weight, const crossclr_sample_weights* sw, int prenormalized,
                          const void* u_video, const void* u_text, long ld_uvideo, long ld_utext, const double* grad_out,
                          void* workspace, size_t workspace_bytes, void* h_video, void* h_text, long ld_hvideo, long ld_htext,
                          double* d_grad_out, void* stream);

/* Which kernels the most recent forward (which = 0) / gradient-product (which = 1) launch of this process went to: the kernel template's name
 * (e.g. "fast_fwd_pair_kernel", "fast_bwd_xfp_kernel"; "" before the first launch).  Reporting aid (bench.py labels its dominant kernel
 * with it); no reference counterpart.                                                                                   */
const char* crossclr_last_kernel(int which);

/* Hardware assumption checks (MFMA fragment layouts, ds_read_b64_tr_b16 gather).  `out` is a
 * device buffer of at least 64 KiB; the caller compares it with the documented layouts.        */
int crossclr_selftest(int which, const void* in, void* out, void* stream);

/* Measurement aid (bench.py `roofline.sustained_mfma`; no reference counterpart): `blocks` x 4 waves x `iters` x 16
 * v_mfma_f32_32x32x16_bf16 on 16 independent accumulators per wave and nothing else -- what the matrix pipe of THIS device
 * sustains with toggling (pseudo-random) or all-zero operands.  `out`: device buffer of blocks * 256 floats (a checksum
 * nobody reads).  flop per launch = blocks * 4 * iters * 16 * 32768.                                                     */
int crossclr_mfma_sustained(float* out, int blocks, int iters, unsigned seed, int zero_operands, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CROSSCLR_H */
