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

#define CROSSCLR_ABI_VERSION 1

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
#define CROSSCLR_E_RANGE (-2)    /* temperature too small for the fixed-shift softmax */
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
 * workspace `part` (plan->fwd_ws_floats floats; at most two launches per step: slot0 = 0 and
 * slot0 = fwd_slots):
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

/* Reduce `nslots` partial slots: logz[2][bpad] (natural log of the full denominator),
 * rz = 1/Z_shifted, wrz = negative_weight * rz (both 0 on padding rows) and
 * loss_sum[0] = sum over valid rows of (logZv + logZt - 2 A_ii)  (double); loss_sum must hold
 * plan->loss_ws_doubles doubles ([1..] are per-block partials, added in index order).          */
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

/* Hardware assumption checks (MFMA fragment layouts, ds_read_b64_tr_b16 gather).  `out` is a
 * device buffer of at least 64 KiB; the caller compares it with the documented layouts.        */
int crossclr_selftest(int which, const void* in, void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CROSSCLR_H */
