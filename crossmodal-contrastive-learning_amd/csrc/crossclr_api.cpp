// crossclr_api.cpp -- the extern "C" boundary declared in include/crossclr.h.
// Built by hipcc for gfx950 into libcrossclr_hip.so (the product).  The same file is built by the
// host clang with -DCROSSCLR_EMU into tests/emu/libcrossclr_emu.so, where "launch" means running
// the kernel source lane by lane on CPU threads -- test infrastructure only.
#include "../../include/crossclr.h"
#include "crossclr_kernels_generic.h"
#include "crossclr_kernels_hvp.h"
#ifndef CROSSCLR_NO_FAST
#include "crossclr_kernels_fast.h"
#include "crossclr_kernels_project.h"
#include "crossclr_kernels_saved32.h"
#endif

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef CROSSCLR_DEFAULT_BWD_KERNEL
#define CROSSCLR_DEFAULT_BWD_KERNEL 1   // 1: 32-row waves, 2: 16-row waves (for Dpad <= 512, where both exist)
#endif

using namespace crossclr;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// which kernel template the most recent forward (0) / gradient-product (1) launch of this process went to (crossclr_last_kernel; pointers to
// string literals, written whole: a reader on another thread -- autograd runs backward() on its own -- sees the old or the new name)
static const char* volatile g_last_kernel[2] = {"", ""};
namespace crossclr {
void note_kernel(int which, const char* name) { g_last_kernel[which & 1] = name; }
}  // namespace crossclr
static void note_generic_launch(const char* what) {      // (the generic kernels are launched from this file: their launch_status label names them)
    if (!strncmp(what, "fwd_sums_kernel", 15)) crossclr::note_kernel(0, what);
    else if (!strncmp(what, "bwd_kernel", 10) || !strncmp(what, "bwd_saved32_kernel", 18)) crossclr::note_kernel(1, what);
}

#ifdef CROSSCLR_EMU
#define LAUNCH(kernel, grid, block, stream, ...) emu::launch(kernel, grid, block, __VA_ARGS__)
static int launch_status(const char* what) { note_generic_launch(what); return CROSSCLR_OK; }
#else
#define LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, 0, (hipStream_t)(stream), __VA_ARGS__)
static int launch_status(const char* what) {
    note_generic_launch(what);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CROSSCLR_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return CROSSCLR_OK;
}
#endif

extern "C" int crossclr_abi_version(void) { return CROSSCLR_ABI_VERSION; }
extern "C" const char* crossclr_last_error(void) { return g_err; }
extern "C" const char* crossclr_last_kernel(int which) { return g_last_kernel[which & 1]; }
extern "C" const char* crossclr_backend(void) {
#ifdef CROSSCLR_EMU
    return "emu-host";
#else
    return "hip-gfx950";
#endif
}

static int round_up(int x, int m) { return (x + m - 1) / m * m; }
static const float* const kNoF = nullptr;     // (kernel arguments a launch does not use)

// tuning knobs from the environment, read ONCE per process (not per call: crossclr_make_plan sits on the step's host path)
struct EnvKnobs {
    bool disable_fast, disable_symmetric, disable_save, bwd_dc256, disable_xf;
    int bwd_kernel, fwd_blocks, bwd_slices;
    EnvKnobs() {
        disable_fast = getenv("CROSSCLR_DISABLE_FAST") != nullptr;
        disable_symmetric = getenv("CROSSCLR_DISABLE_SYMMETRIC") != nullptr;
        disable_save = getenv("CROSSCLR_DISABLE_SAVE") != nullptr;
        disable_xf = getenv("CROSSCLR_DISABLE_XF") != nullptr;  // no fragment-major operand copy: the saved backward stages the column tiles through LDS
        bwd_dc256 = getenv("CROSSCLR_BWD_DC256") != nullptr;    // generic backward: D slices of 256 columns even where 512 divides Dpad (A/B)
        const char* e = getenv("CROSSCLR_BWD_KERNEL");
        bwd_kernel = e ? atoi(e) : 0;
        e = getenv("CROSSCLR_FWD_BLOCKS");
        fwd_blocks = e ? atoi(e) : 0;
        e = getenv("CROSSCLR_BWD_SLICES");
        bwd_slices = e ? atoi(e) : 0;
    }
};
static const EnvKnobs& env_knobs() {
#ifdef CROSSCLR_EMU
    static thread_local EnvKnobs k;   // the CPU tests flip these variables between calls (monkeypatch): re-read there
    k = EnvKnobs();
    return k;
#else
    static const EnvKnobs k;
    return k;
#endif
}
static int forward_generic(const crossclr_plan* plan, const Geo& g, const void* rows, const void* cols, float* out,
                           const float* kcols, const float* shift, int mode, void* stream, float* stash = nullptr,
                           const float* shift_cols = nullptr);

// forward workspace ("part") layout, in floats:
//   [4 launch groups][fwd_slots][2*bpad] | colpart (symmetric launch) [<= 2*bpad/128 row blocks][2*bpad]
//   | colpart (pairs launch) [row blocks][(world-1)/2 ranks * 2*bpad] | header [4][4] ints
// (the symmetric launch's column sums are read by the finish kernel, so the pairs launch needs its own region)
static const int kLaunchGroups = CROSSCLR_LAUNCH_GROUPS;
static size_t ws_colpart_off(const crossclr_plan* p) { return (size_t)kLaunchGroups * p->fwd_slots * 2 * p->bpad; }
static size_t ws_colpart_rows(const crossclr_plan* p) { return (size_t)(2 * p->bpad / 128 + 1); }
static size_t ws_paircol_off(const crossclr_plan* p) { return ws_colpart_off(p) + ws_colpart_rows(p) * 2 * p->bpad; }
static size_t ws_flag_off(const crossclr_plan* p) {
    const size_t k = p->world > 2 ? (size_t)(p->world - 1) / 2 : 0;
    return ws_paircol_off(p) + ws_colpart_rows(p) * k * 2 * p->bpad;
}

static int device_zero(void* where, size_t bytes, void* stream) {
#ifdef CROSSCLR_EMU
    (void)stream;
    memset(where, 0, bytes);
    return CROSSCLR_OK;
#else
    hipError_t e = hipMemsetAsync(where, 0, bytes, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CROSSCLR_E_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    return CROSSCLR_OK;
#endif
}
static int device_zero_header(int* where, void* stream) {  // kind 0 = dense slots (generic kernels)
#ifdef CROSSCLR_EMU
    (void)stream;
    memset(where, 0, 16);
    return CROSSCLR_OK;
#else
    hipError_t e = hipMemsetAsync(where, 0, 16, (hipStream_t)stream);
    if (e != hipSuccess) return fail(CROSSCLR_E_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    return CROSSCLR_OK;
#endif
}

extern "C" int crossclr_make_plan(int b, int D, int world, int rank, int mode, crossclr_plan* plan) {
    if (!plan) return fail(CROSSCLR_E_ARG, "plan is NULL");
    if (b < 1 || D < 1) return fail(CROSSCLR_E_ARG, "need b >= 1 and D >= 1 (got b=%d D=%d)", b, D);
    if (world < 1 || rank < 0 || rank >= world) return fail(CROSSCLR_E_ARG, "bad world/rank %d/%d", world, rank);
    if (mode != CROSSCLR_MODE_FP32 && mode != CROSSCLR_MODE_BF16) return fail(CROSSCLR_E_ARG, "bad mode %d", mode);
    // the forward's flat work list, the stash and the column-sum workspace are indexed with 32-bit integers:
    // (2b/128 row blocks) x (2B/32 column tiles) must stay below 2^31 (whichever kernel the plan ends up with)
    {
        const long long bp = ((long long)b + kRowPad - 1) / kRowPad * kRowPad;
        const long long items = (2 * bp / 128) * (2 * bp * world / 32);
        if ((long long)b * world > (1 << 21) || items >= (1LL << 31))
            return fail(CROSSCLR_E_ARG, "batch too large: b=%d x world=%d gives %lld work items (limit 2^31; global rows limit 2^21)", b, world, items);
    }
    memset(plan, 0, sizeof(*plan));
    plan->b = b; plan->D = D; plan->world = world; plan->rank = rank; plan->mode = mode;
    plan->bpad = round_up(b, kRowPad);
    plan->fast_path = 0;
    int dpad = round_up(D, 64);
#ifndef CROSSCLR_NO_FAST
    const EnvKnobs& env = env_knobs();
    if (mode == CROSSCLR_MODE_BF16 && !env.disable_fast) {  // env knob: A/B against the generic path
        int fp = fast_dpad(D);
        if (fp > 0) { dpad = fp; plan->fast_path = 1; }
    }
#endif
    if (!plan->fast_path && dpad > 256) dpad = round_up(D, 256);  // generic backward slices D by 256
    // wide bf16 plans (1024 < D <= 8192): generic forward, but its exponentials are saved for the D-slice backward, which runs as 3 ... 16
    // column parts of 384 / 512 columns -- the operand is padded to parts x columns
    bool wide = false;
#ifndef CROSSCLR_NO_FAST
    if (mode == CROSSCLR_MODE_BF16 && !plan->fast_path && !env.disable_fast && !env.disable_save && !env.disable_symmetric && wide_bf16_dpad(D) > 0) {
        dpad = wide_bf16_dpad(D);
        wide = true;
    }
#endif
    plan->Dpad = dpad;
    // backward kernel: 0 generic tiled, 1 register-resident 32-row waves (Dpad <= 512), 2 16-row waves (Dpad <= 1024)
    plan->fast_bwd = 0;
#ifndef CROSSCLR_NO_FAST
    if (mode == CROSSCLR_MODE_BF16 && !env.disable_fast) {
        if (plan->fast_path) plan->fast_bwd = dpad <= 512 ? CROSSCLR_DEFAULT_BWD_KERNEL : 2;
        if (env.bwd_kernel == 16 && plan->fast_path) plan->fast_bwd = 2;              // tuning knob: 16 or 32
        if (env.bwd_kernel == 32 && plan->fast_path && dpad <= 512) plan->fast_bwd = 1;
    }
#endif
    // forward partial-sum slots.  generic kernels: (row block x column split) grid, enough work items to
    // fill 256 CUs a few times over.  fast path: persistent blocks over a flat work list (see FwdWork);
    // a row block's slots = the thread blocks whose range touches it.
    const int row_blocks = 2 * plan->bpad / 128;
    const int col_tiles = 2 * plan->bpad / 128;  // per column rank
    int nsplit = (1024 + row_blocks - 1) / row_blocks;
    if (nsplit > col_tiles) nsplit = col_tiles;
    if (nsplit < 1) nsplit = 1;
    plan->fwd_blocks = 0;
#ifndef CROSSCLR_NO_FAST
    if (plan->fast_path) {
        plan->fwd_blocks = 256;  // one persistent block per MI355X CU (LDS-limited to one block per CU)
        if (env.fwd_blocks >= 1 && env.fwd_blocks <= 4096) plan->fwd_blocks = env.fwd_blocks;   // tuning knob
        int slots = fwd_max_slots(fast_forward_work(plan, 1, -1, true));
        int s2 = fwd_max_slots(fast_forward_work(plan, 1, -1, false));
        if (s2 > slots) slots = s2;
        if (world > 1) {
            int s3 = fwd_max_slots(fast_forward_work(plan, world, 0, false));
            if (s3 > slots) slots = s3;
        }
        if (world > 2) {   // crossclr_forward_pairs over (world-1)/2 ranks
            int s4 = fwd_max_slots(fast_forward_work(plan, (world - 1) / 2, -1, false, true));
            if (s4 > slots) slots = s4;
        }
        nsplit = slots;
    }
#endif
    plan->fwd_slots = nsplit;
    plan->fwd_ws_floats = 0;  // set below
    // backward column slices: each slice walks its share of the column tiles and writes its own gradient
    // slice (summed by crossclr_backward_finish): enough thread blocks to occupy 256 CUs, >= 2 tiles each
    {
        const int tile = plan->fast_bwd ? 32 : 64;
        int row_blk = 64;
#ifndef CROSSCLR_NO_FAST
        if (plan->fast_bwd) row_blk = fast_bwd_rows_per_block(plan->Dpad, plan->fast_bwd == 2);
#endif
        int dsl = 1;
        if (!plan->fast_bwd) dsl = plan->Dpad % 256 == 0 ? plan->Dpad / 256 : (plan->Dpad % 128 == 0 ? plan->Dpad / 128 : plan->Dpad / 64);
        const int blocks = (2 * plan->bpad / row_blk) * dsl;
        int sl = (256 + blocks - 1) / blocks;
        if (wide) {   // saved backward: (128-row blocks) x (column parts) x slices thread blocks, one per CU: the cut with the fewest rounds of 256
            const int wb = (2 * plan->bpad / 128) * (dpad / (dpad % 512 == 0 ? 512 : 384));
            double best = 1e30;
            for (int c = 1; c <= 4; ++c) {
                const double rounds = (double)((wb * c + 255) / 256) / c;
                if (rounds < best - 1e-9) { best = rounds; sl = c; }
            }
            if (wb < 256 && sl < (256 + wb - 1) / wb) sl = (256 + wb - 1) / wb;
        }
        const int tiles = 2 * plan->bpad / tile;
        if (sl > tiles / 2) sl = tiles / 2;
        if (sl > 16) sl = 16;
        if (sl < 1) sl = 1;
        if (env.bwd_slices >= 1 && env.bwd_slices <= tiles / 2 && env.bwd_slices <= 64) sl = env.bwd_slices;   // tuning knob
        plan->bwd_slices = sl;
    }
    {
#ifndef CROSSCLR_FINISH_LPR
#define CROSSCLR_FINISH_LPR 4
#endif
        const int rows_per_block = 256 / CROSSCLR_FINISH_LPR;
        int nb = (2 * plan->bpad + rows_per_block - 1) / rows_per_block;      // finish kernels: 256 / LPR rows per block, grid-stride beyond
        if (nb > 1024) nb = 1024;
        plan->loss_ws_doubles = 1 + nb;
    }
    plan->fwd_ws_floats = ws_flag_off(plan) + 4 * kLaunchGroups;   // + the launch groups' headers
    const size_t esz = mode == CROSSCLR_MODE_FP32 ? 4 : 2;
    plan->operand_bytes = (size_t)2 * plan->bpad * plan->Dpad * esz;
    plan->gbuf_bytes = (size_t)plan->bwd_slices * 2 * plan->bpad * plan->Dpad * 4;
    plan->stash_bytes = 0;
#ifndef CROSSCLR_NO_FAST
    if (plan->fast_path && plan->fast_bwd && !env.disable_save) plan->stash_bytes = fast_stash_bytes(plan->bpad, plan->Dpad);
    if (wide) plan->stash_bytes = wide_stash_bytes(plan->bpad);
    if (wide && plan->stash_bytes && !env.disable_xf && plan->operand_bytes < ((size_t)1 << 32)) plan->xf_bytes = plan->operand_bytes;
    // exact-fp32 mode: the whole stacked [2 bpad] x [2 bpad] matrix of fp32 exponentials (1 GiB at b = 8192), up to 16 GiB
    if (mode == CROSSCLR_MODE_FP32 && !env.disable_save && plan->operand_bytes < (1ull << 32)) {
        const size_t sb = (size_t)2 * plan->bpad * (size_t)2 * plan->bpad * 4;
        if (sb <= ((size_t)16 << 30)) plan->stash_bytes = sb;
    }
    // the fragment-major copy of the bf16 operand (crossclr_normalize_xf -> crossclr_backward_saved_xf): local block, Dpad <= 1024
    if (plan->fast_path && plan->fast_bwd && plan->stash_bytes && plan->Dpad <= 1024 && !env.disable_xf) plan->xf_bytes = plan->operand_bytes;
#else
    plan->xf_bytes = 0;
#endif
    return CROSSCLR_OK;
}

static bool needs_row_shift(float temperature, float negative_weight) {
    const double it = 1.0 / (double)temperature, aw = fabs((double)negative_weight);
    return it * (aw > 1.0 ? aw : 1.0) > 128.0;   // |logit| <= this bound because the rows are unit vectors
}

static int make_geo(const crossclr_plan* p, int col_ranks, int col_rank0, int skip_rank, float temperature,
                    float negative_weight, Geo* g, bool allow_row_shift = false) {
    if (!p) return fail(CROSSCLR_E_ARG, "plan is NULL");
    if (!(temperature > 0.f) || !isfinite(temperature)) return fail(CROSSCLR_E_ARG, "temperature must be > 0");
    if (!isfinite(negative_weight)) return fail(CROSSCLR_E_ARG, "negative_weight must be finite");
    if (col_ranks < 1) return fail(CROSSCLR_E_ARG, "col_ranks must be >= 1");
    g->b = p->b; g->bpad = p->bpad; g->D = p->D; g->Dpad = p->Dpad;
    g->col_ranks = col_ranks; g->col_rank0 = col_rank0; g->row_rank = p->rank; g->skip_rank = skip_rank;
    g->col_wrap = 0;
    const double it = 1.0 / (double)temperature;
    const double aw = fabs((double)negative_weight);
    const double bound = it * (aw > 1.0 ? aw : 1.0);  // |logit| <= bound because the rows are unit vectors
    // fixed soft-max shift: exp(logit - shift) must neither overflow (<= e^64 per term) nor push the
    // always-present exp(0 - shift) self term out of fp32 range (shift <= 64)
    double shift = bound > 64.0 ? bound - 64.0 : 0.0;
    g->row_shift = 0;
    if (shift > 64.0) {
        if (!allow_row_shift)
            return fail(CROSSCLR_E_RANGE, "temperature %g too small for the fixed-shift soft-max (max |logit| %g > 128): use the "
                        "two-pass entry points (crossclr_forward_rowmax + crossclr_*_s)", (double)temperature, bound);
        g->row_shift = 1;   // per-row shifts (the row maxima of crossclr_forward_rowmax) replace the common one
        shift = 0.0;
    }
    g->c_inter = (float)(it * (double)kLog2e);
    g->c_intra = (float)(it * (double)negative_weight * (double)kLog2e);
    g->m2 = (float)(shift * (double)kLog2e);
    return CROSSCLR_OK;
}

// ------------------------------------------------------------------------------------------------
template <typename TIN, bool NORM>
static int normalize_t(const crossclr_plan* p, const void* v, const void* t, long ldv, long ldt, void* xhat,
                       float* inv_norm, float* diag, void* stream, int* zero_word) {
    Geo g; memset(&g, 0, sizeof(g));
    g.b = p->b; g.bpad = p->bpad; g.D = p->D; g.Dpad = p->Dpad;
    dim3 grid((p->bpad + 3) / 4), block(256);
    if (p->mode == CROSSCLR_MODE_FP32)
        LAUNCH((normalize_kernel<TIN, float, NORM>), grid, block, stream, (const TIN*)v, (const TIN*)t, ldv, ldt, g,
               (float*)xhat, inv_norm, diag, zero_word);
    else
        LAUNCH((normalize_kernel<TIN, bf16_t, NORM>), grid, block, stream, (const TIN*)v, (const TIN*)t, ldv, ldt, g,
               (bf16_t*)xhat, inv_norm, diag, zero_word);
    return launch_status("normalize_kernel");
}
template <bool NORM>
static int normalize_any(const crossclr_plan* plan, const void* video, const void* text, long ld_video, long ld_text, int in_dtype,
                         void* xhat, float* inv_norm, float* diag_cos, void* stream, int* zero_word = nullptr) {
    if (!plan || !video || !text || !xhat || !inv_norm || !diag_cos) return fail(CROSSCLR_E_ARG, "NULL argument");
    if (ld_video < plan->D || ld_text < plan->D) return fail(CROSSCLR_E_ARG, "row stride smaller than D");
    switch (in_dtype) {
        case CROSSCLR_IN_F32: return normalize_t<float, NORM>(plan, video, text, ld_video, ld_text, xhat, inv_norm, diag_cos, stream, zero_word);
        case CROSSCLR_IN_F64: return normalize_t<double, NORM>(plan, video, text, ld_video, ld_text, xhat, inv_norm, diag_cos, stream, zero_word);
        case CROSSCLR_IN_F16: return normalize_t<in_f16, NORM>(plan, video, text, ld_video, ld_text, xhat, inv_norm, diag_cos, stream, zero_word);
        case CROSSCLR_IN_BF16: return normalize_t<in_bf16, NORM>(plan, video, text, ld_video, ld_text, xhat, inv_norm, diag_cos, stream, zero_word);
    }
    return fail(CROSSCLR_E_ARG, "bad in_dtype %d", in_dtype);
}

#ifndef CROSSCLR_NO_FAST
template <typename TIN, bool NORM>
static int normalize_xf_t(const crossclr_plan* p, const void* v, const void* t, long ldv, long ldt, void* xhat, void* xf,
                          float* inv_norm, float* diag, void* stream, int* zero_word) {
    Geo g; memset(&g, 0, sizeof(g));
    g.b = p->b; g.bpad = p->bpad; g.D = p->D; g.Dpad = p->Dpad;
    if (p->Dpad <= 512)
        LAUNCH((normalize_xf_kernel<TIN, NORM, 2>), dim3(p->bpad / 16), dim3(512), stream, (const TIN*)v, (const TIN*)t, ldv, ldt, g,
               (bf16_t*)xhat, (unsigned char*)xf, inv_norm, diag, zero_word);
    else
        LAUNCH((normalize_xf_kernel<TIN, NORM, 4>), dim3(p->bpad / 16), dim3(512), stream, (const TIN*)v, (const TIN*)t, ldv, ldt, g,
               (bf16_t*)xhat, (unsigned char*)xf, inv_norm, diag, zero_word);
    return launch_status("normalize_xf_kernel");
}
#endif
template <bool NORM>
static int normalize_xf_any(const crossclr_plan* plan, const void* video, const void* text, long ld_video, long ld_text, int in_dtype,
                            void* xhat, void* xf, float* inv_norm, float* diag_cos, void* stream, int* zero_word = nullptr) {
    if (!plan || !video || !text || !xhat || !xf || !inv_norm || !diag_cos) return fail(CROSSCLR_E_ARG, "NULL argument");
    if (ld_video < plan->D || ld_text < plan->D) return fail(CROSSCLR_E_ARG, "row stride smaller than D");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "the fragment-major operand needs the register-resident path");
#else
    if (!plan->xf_bytes) return fail(CROSSCLR_E_ARG, "this plan has no fragment-major operand (xf_bytes == 0): use crossclr_normalize / crossclr_pack");
    if (plan->Dpad > 1024) {   // wide plans: the row kernel, then the packed rows re-laid fragment-major (1024 columns of a 32-row tile per block)
        if (int rc = normalize_any<NORM>(plan, video, text, ld_video, ld_text, in_dtype, xhat, inv_norm, diag_cos, stream, zero_word)) return rc;
        return crossclr_pack_xf_from_packed(plan, xhat, 1, xf, stream);
    }
    switch (in_dtype) {
        case CROSSCLR_IN_F32: return normalize_xf_t<float, NORM>(plan, video, text, ld_video, ld_text, xhat, xf, inv_norm, diag_cos, stream, zero_word);
        case CROSSCLR_IN_F64: return normalize_xf_t<double, NORM>(plan, video, text, ld_video, ld_text, xhat, xf, inv_norm, diag_cos, stream, zero_word);
        case CROSSCLR_IN_F16: return normalize_xf_t<in_f16, NORM>(plan, video, text, ld_video, ld_text, xhat, xf, inv_norm, diag_cos, stream, zero_word);
        case CROSSCLR_IN_BF16: return normalize_xf_t<in_bf16, NORM>(plan, video, text, ld_video, ld_text, xhat, xf, inv_norm, diag_cos, stream, zero_word);
    }
    return fail(CROSSCLR_E_ARG, "bad in_dtype %d", in_dtype);
#endif
}
extern "C" int crossclr_normalize_xf(const crossclr_plan* plan, const void* video, const void* text, long ld_video, long ld_text,
                                     int in_dtype, void* xhat, void* xhat_xf, float* inv_norm, float* diag_cos, void* stream) {
    return normalize_xf_any<true>(plan, video, text, ld_video, ld_text, in_dtype, xhat, xhat_xf, inv_norm, diag_cos, stream);
}
extern "C" int crossclr_pack_xf(const crossclr_plan* plan, const void* video_hat, const void* text_hat, long ld_video, long ld_text,
                                int in_dtype, void* xhat, void* xhat_xf, float* inv_norm, float* diag_cos, void* stream) {
    return normalize_xf_any<false>(plan, video_hat, text_hat, ld_video, ld_text, in_dtype, xhat, xhat_xf, inv_norm, diag_cos, stream);
}

extern "C" int crossclr_pack(const crossclr_plan* plan, const void* video_hat, const void* text_hat, long ld_video,
                             long ld_text, int in_dtype, void* xhat, float* inv_norm, float* diag_cos, void* stream) {
    return normalize_any<false>(plan, video_hat, text_hat, ld_video, ld_text, in_dtype, xhat, inv_norm, diag_cos, stream);
}

extern "C" int crossclr_normalize(const crossclr_plan* plan, const void* video, const void* text, long ld_video,
                                  long ld_text, int in_dtype, void* xhat, float* inv_norm, float* diag_cos,
                                  void* stream) {
    return normalize_any<true>(plan, video, text, ld_video, ld_text, in_dtype, xhat, inv_norm, diag_cos, stream);
}

// ------------------------------------------------------------------------------------------------
// producer-side fusion (crossclr_kernels_project.h)
#ifndef CROSSCLR_NO_FAST
template <typename TIN, bool WF>
static int project_pack_t(const crossclr_plan* p, const void* xv, const void* xt, long ldv, long ldt, int Din_v, int Din_t, const void* wv,
                          const void* wt, int ldw_v, int ldw_t, const float* bv, const float* bt, void* xhat, float* inv_norm, float* diag,
                          void* stream) {
    Geo g; memset(&g, 0, sizeof(g));
    g.b = p->b; g.bpad = p->bpad; g.D = p->D; g.Dpad = p->Dpad;
    // 64 rows per block up to Dpad = 512 when that still gives every CU a block; 32 rows otherwise (and always above 512: accumulators)
    static const int rows_knob = [] { const char* e = getenv("CROSSCLR_PROJECT_ROWS"); return e ? atoi(e) : 0; }();      // 32 / 64: tuning knob (A/B)
    const bool rows32 = rows_knob == 32 || (rows_knob != 64 && p->bpad / 64 < 256);
    dim3 grid(p->bpad / 64), grid32(p->bpad / 32), block(256);
#define CROSSCLR_LPPX(DKP, RFV, GRID) LAUNCH((project_pack_kernel<TIN, DKP, RFV, WF>), GRID, block, stream, (const TIN*)xv, (const TIN*)xt, ldv, ldt, Din_v, Din_t, \
                                             (const bf16_t*)wv, (const bf16_t*)wt, ldw_v, ldw_t, bv, bt, g, (bf16_t*)xhat, inv_norm, diag)
#define CROSSCLR_LPP(DKP) do { if (rows32) CROSSCLR_LPPX(DKP, 1, grid32); else CROSSCLR_LPPX(DKP, 2, grid); } while (0)
    switch (p->Dpad) {
        case 128: CROSSCLR_LPP(1); break;
        case 256: CROSSCLR_LPP(2); break;
        case 384: CROSSCLR_LPP(3); break;
        case 512: CROSSCLR_LPP(4); break;
        case 768: CROSSCLR_LPPX(6, 1, grid32); break;        // wide embeddings: 32 rows per block (the accumulators of 64 x 1024 x 2 would not fit)
        case 1024: CROSSCLR_LPPX(8, 1, grid32); break;
        default: return fail(CROSSCLR_E_ARG, "crossclr_project_pack: Dpad %d (supported: <= 1024)", p->Dpad);
    }
#undef CROSSCLR_LPP
#undef CROSSCLR_LPPX
    return launch_status("project_pack_kernel");
}
#endif

template <bool WF>
static int project_pack_any(const crossclr_plan* plan, const void* x_video, const void* x_text, long ld_video, long ld_text,
                            int Din_video, int Din_text, int in_dtype, const void* w_video, const void* w_text, int ldw_video,
                            int ldw_text, const float* bias_video, const float* bias_text, void* xhat, float* inv_norm, float* diag_cos, void* stream) {
    if (!plan || !x_video || !x_text || !w_video || !w_text || !xhat || !inv_norm || !diag_cos) return fail(CROSSCLR_E_ARG, "NULL argument");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_project_pack needs the register-resident path");
#else
    if (plan->mode != CROSSCLR_MODE_BF16 || !plan->fast_path || plan->Dpad > 1024)
        return fail(CROSSCLR_E_ARG, "crossclr_project_pack: bf16 plans with D <= 1024 only");
    if (Din_video < 1 || Din_text < 1 || ld_video < Din_video || ld_text < Din_text) return fail(CROSSCLR_E_ARG, "row stride smaller than Din");
    if (ldw_video % 64 != 0 || ldw_video < Din_video || ldw_text % 64 != 0 || ldw_text < Din_text)
        return fail(CROSSCLR_E_ARG, "weights must be zero-padded to ldw = a multiple of 64 >= Din (got %d / %d for Din %d / %d)", ldw_video, ldw_text, Din_video, Din_text);
    switch (in_dtype) {
        case CROSSCLR_IN_F32: return project_pack_t<float, WF>(plan, x_video, x_text, ld_video, ld_text, Din_video, Din_text, w_video, w_text, ldw_video, ldw_text, bias_video, bias_text, xhat, inv_norm, diag_cos, stream);
        case CROSSCLR_IN_F64: return project_pack_t<double, WF>(plan, x_video, x_text, ld_video, ld_text, Din_video, Din_text, w_video, w_text, ldw_video, ldw_text, bias_video, bias_text, xhat, inv_norm, diag_cos, stream);
        case CROSSCLR_IN_F16: return project_pack_t<in_f16, WF>(plan, x_video, x_text, ld_video, ld_text, Din_video, Din_text, w_video, w_text, ldw_video, ldw_text, bias_video, bias_text, xhat, inv_norm, diag_cos, stream);
        case CROSSCLR_IN_BF16: return project_pack_t<in_bf16, WF>(plan, x_video, x_text, ld_video, ld_text, Din_video, Din_text, w_video, w_text, ldw_video, ldw_text, bias_video, bias_text, xhat, inv_norm, diag_cos, stream);
    }
    return fail(CROSSCLR_E_ARG, "bad in_dtype %d", in_dtype);
#endif
}
extern "C" int crossclr_project_pack(const crossclr_plan* plan, const void* x_video, const void* x_text, long ld_video, long ld_text,
                                     int Din_video, int Din_text, int in_dtype, const void* w_video, const void* w_text, int ldw_video,
                                     int ldw_text, const float* bias_video,
                                     const float* bias_text, void* xhat, float* inv_norm, float* diag_cos, void* stream) {
    return project_pack_any<false>(plan, x_video, x_text, ld_video, ld_text, Din_video, Din_text, in_dtype, w_video, w_text, ldw_video, ldw_text,
                                   bias_video, bias_text, xhat, inv_norm, diag_cos, stream);
}
extern "C" int crossclr_project_pack_wf(const crossclr_plan* plan, const void* x_video, const void* x_text, long ld_video, long ld_text,
                                        int Din_video, int Din_text, int in_dtype, const void* wf_video, const void* wf_text, int ldw_video,
                                        int ldw_text, const float* bias_video,
                                        const float* bias_text, void* xhat, float* inv_norm, float* diag_cos, void* stream) {
    return project_pack_any<true>(plan, x_video, x_text, ld_video, ld_text, Din_video, Din_text, in_dtype, wf_video, wf_text, ldw_video, ldw_text,
                                  bias_video, bias_text, xhat, inv_norm, diag_cos, stream);
}

// the projection's weight gradient: split-K MFMA kernel + reduce (crossclr_kernels_project.h)
static int project_dw_splits(int b, int D, int Din_v, int Din_t) {
    const int Dp = round_up(D, 128), Dinp = round_up(Din_v > Din_t ? Din_v : Din_t, 128);
    const int tiles = 2 * (Dp / 128) * (Dinp / 128);
    int ns = (512 + tiles - 1) / tiles;                     // two thread blocks per CU: one wave per SIMD alone hides no latency
    const int max_by_rows = b / 64 > 1 ? b / 64 : 1;        // at least two 32-row chunks per split
    if (ns > max_by_rows) ns = max_by_rows;
    if (ns > 32) ns = 32;
    return ns < 1 ? 1 : ns;
}
extern "C" size_t crossclr_project_dw_ws_floats(int b, int D, int Din_video, int Din_text) {
    if (b < 1 || D < 1 || Din_video < 1 || Din_text < 1) return 0;
    const size_t Dp = round_up(D, 128), Dinp = round_up(Din_video > Din_text ? Din_video : Din_text, 128);
    const size_t ns = project_dw_splits(b, D, Din_video, Din_text);
    return 2 * ns * Dp * Dinp + 2 * ns * Dp;
}
template <typename TIN>
static int project_dw_t(int b, int D, const void* gyv, const void* gyt, long ldgy, const void* xv, const void* xt, long ldxv, long ldxt,
                        int Din_v, int Din_t, float* ws, float* dwv, float* dwt, long lddwv, long lddwt, float* dbv, float* dbt, void* stream) {
    const int Dp = round_up(D, 128), Dinp = round_up(Din_v > Din_t ? Din_v : Din_t, 128);
    const int ns = project_dw_splits(b, D, Din_v, Din_t);
    float* partial = ws;
    float* dbpart = ws + (size_t)2 * ns * Dp * Dinp;
    LAUNCH((project_dw_kernel<TIN>), dim3(2 * ns, Dp / 128, Dinp / 128), dim3(256), stream, (const in_bf16*)gyv, (const in_bf16*)gyt, ldgy,
           (const TIN*)xv, (const TIN*)xt, ldxv, ldxt, b, D, Din_v, Din_t, ns, partial, Dp, Dinp, dbpart);
    LAUNCH(project_dw_reduce_kernel, dim3((Dinp + 255) / 256, D, 2), dim3(256), stream, (const float*)partial, (const float*)dbpart, ns, D, Dp, Dinp,
           Din_v, Din_t, dwv, dwt, lddwv, lddwt, dbv, dbt);
    return launch_status("project_dw_kernel");
}
extern "C" int crossclr_project_dw(int b, int D, const void* gy_video, const void* gy_text, long ld_gy, const void* x_video, const void* x_text,
                                   long ld_xv, long ld_xt, int Din_video, int Din_text, int in_dtype, float* ws, float* dw_video, float* dw_text,
                                   long ld_dwv, long ld_dwt, float* db_video, float* db_text, void* stream) {
    if (!gy_video || !gy_text || !x_video || !x_text || !ws || !dw_video || !dw_text) return fail(CROSSCLR_E_ARG, "NULL argument");
    if (b < 1 || D < 1 || Din_video < 1 || Din_text < 1 || ld_gy < D || ld_xv < Din_video || ld_xt < Din_text || ld_dwv < Din_video || ld_dwt < Din_text)
        return fail(CROSSCLR_E_ARG, "crossclr_project_dw: bad sizes / strides");
    switch (in_dtype) {
        case CROSSCLR_IN_F32: return project_dw_t<float>(b, D, gy_video, gy_text, ld_gy, x_video, x_text, ld_xv, ld_xt, Din_video, Din_text, ws, dw_video, dw_text, ld_dwv, ld_dwt, db_video, db_text, stream);
        case CROSSCLR_IN_F64: return project_dw_t<double>(b, D, gy_video, gy_text, ld_gy, x_video, x_text, ld_xv, ld_xt, Din_video, Din_text, ws, dw_video, dw_text, ld_dwv, ld_dwt, db_video, db_text, stream);
        case CROSSCLR_IN_F16: return project_dw_t<in_f16>(b, D, gy_video, gy_text, ld_gy, x_video, x_text, ld_xv, ld_xt, Din_video, Din_text, ws, dw_video, dw_text, ld_dwv, ld_dwt, db_video, db_text, stream);
        case CROSSCLR_IN_BF16: return project_dw_t<in_bf16>(b, D, gy_video, gy_text, ld_gy, x_video, x_text, ld_xv, ld_xt, Din_video, Din_text, ws, dw_video, dw_text, ld_dwv, ld_dwt, db_video, db_text, stream);
    }
    return fail(CROSSCLR_E_ARG, "bad in_dtype %d", in_dtype);
}

extern "C" int crossclr_project_backward_prep(const crossclr_plan* plan, const float* g_video, const float* g_text, long ld_gv,
                                              long ld_gt, const void* xhat, const float* inv_norm, float* gy_video, float* gy_text,
                                              long ld_out, void* stream) {
    if (!plan || !g_video || !g_text || !xhat || !inv_norm || !gy_video || !gy_text) return fail(CROSSCLR_E_ARG, "NULL argument");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_project_backward_prep needs the register-resident path");
#else
    if (plan->mode != CROSSCLR_MODE_BF16) return fail(CROSSCLR_E_ARG, "crossclr_project_backward_prep: bf16 plans only");
    if (ld_gv < plan->D || ld_gt < plan->D || ld_out < plan->D) return fail(CROSSCLR_E_ARG, "row stride smaller than D");
    Geo g; memset(&g, 0, sizeof(g));
    g.b = plan->b; g.bpad = plan->bpad; g.D = plan->D; g.Dpad = plan->Dpad;
    LAUNCH(project_backward_prep_kernel, dim3((plan->b + 3) / 4), dim3(256), stream, g_video, g_text, ld_gv, ld_gt, g, (const bf16_t*)xhat,
           inv_norm, gy_video, gy_text, ld_out);
    return launch_status("project_backward_prep_kernel");
#endif
}

// ------------------------------------------------------------------------------------------------
// sample weights: both k arrays or neither (a missing struct / NULL arrays = all ones = the reference's loss)
static int unpack_k(const crossclr_sample_weights* sw, const float** krows, const float** kcols) {
    *krows = sw ? sw->neg_scale_rows : nullptr;
    *kcols = sw ? sw->neg_scale_cols : nullptr;
    if ((*krows == nullptr) != (*kcols == nullptr))
        return fail(CROSSCLR_E_ARG, "neg_scale_rows and neg_scale_cols must be given together");
    return CROSSCLR_OK;
}

// generic tiled forward: MODE 0 sums with the common shift, 1 row maxima, 2 sums with per-row shifts
template <typename T>
static void forward_generic_t(const crossclr_plan* plan, const Geo& g, const void* rows, const void* cols, float* out,
                              const float* kcols, const float* shift, int mode, int tps, dim3 grid, void* stream) {
    dim3 block(256);
    (void)plan;
#define CROSSCLR_LFG(SW, MODE) LAUNCH((fwd_sums_kernel<T, SW, MODE>), grid, block, stream, (const T*)rows, (const T*)cols, g, tps, out, kcols, shift, (float*)nullptr, (int*)nullptr, (float*)nullptr)
    if (kcols) { if (mode == 0) CROSSCLR_LFG(true, 0); else if (mode == 1) CROSSCLR_LFG(true, 1); else CROSSCLR_LFG(true, 2); }
    else { if (mode == 0) CROSSCLR_LFG(false, 0); else if (mode == 1) CROSSCLR_LFG(false, 1); else CROSSCLR_LFG(false, 2); }
#undef CROSSCLR_LFG
}
// symmetric evaluation of the local block by the generic forward (rows == columns): upper triangle + column sums; `shift` != NULL:
// the second pass of the two-pass soft-max (sums relative to per-row shifts); `stash` != NULL (exact-fp32 plans): save the exponentials
template <typename T>
static int forward_generic_sym(const crossclr_plan* plan, const Geo& g, const void* x, float* out, const float* k, float* colpart,
                               int* header, void* stream, float* stash = nullptr, const float* shift = nullptr, bool rowmax = false) {
    dim3 grid(2 * plan->bpad / 256, plan->fwd_slots), block(256);     // one blockIdx.x per PAIR of row blocks (I, ntiles - 1 - I)
#define CROSSCLR_LSY(TT, SW, MODE, ST) \
    LAUNCH((fwd_sums_kernel<TT, SW, MODE, ST, true>), grid, block, stream, (const TT*)x, (const TT*)x, g, 0, out, k, shift, stash, header, colpart)
    if (stash) {   // exact-fp32 plans (T = float: fp32 fragments, both triangles) or wide bf16 plans (bf16 records, upper triangle)
        if constexpr (sizeof(T) == 2) {
            if (shift) return fail(CROSSCLR_E_ARG, "bf16 plans save their exponentials in the single-pass soft-max only");
            if (k) CROSSCLR_LSY(bf16_t, true, 0, true); else CROSSCLR_LSY(bf16_t, false, 0, true);
            return launch_status("fwd_sums_kernel (symmetric, save, bf16 records)");
        } else {
            if (shift) { if (k) CROSSCLR_LSY(float, true, 2, true); else CROSSCLR_LSY(float, false, 2, true); }
            else { if (k) CROSSCLR_LSY(float, true, 0, true); else CROSSCLR_LSY(float, false, 0, true); }
            return launch_status("fwd_sums_kernel (symmetric, save)");
        }
    }
    if (rowmax) { if (k) CROSSCLR_LSY(T, true, 1, false); else CROSSCLR_LSY(T, false, 1, false); }
    else if (shift) { if (k) CROSSCLR_LSY(T, true, 2, false); else CROSSCLR_LSY(T, false, 2, false); }
    else { if (k) CROSSCLR_LSY(T, true, 0, false); else CROSSCLR_LSY(T, false, 0, false); }
#undef CROSSCLR_LSY
    return launch_status("fwd_sums_kernel (symmetric)");
}

static int forward_generic(const crossclr_plan* plan, const Geo& g, const void* rows, const void* cols, float* out,
                           const float* kcols, const float* shift, int mode, void* stream, float* stash, const float* shift_cols) {
    const int ntiles = g.col_ranks * 2 * plan->bpad / 128;
    const int nsplit = plan->fwd_slots;
    const int tps = (ntiles + nsplit - 1) / nsplit;
    dim3 grid(2 * plan->bpad / 128, nsplit);
    if (stash && plan->mode == CROSSCLR_MODE_BF16) {   // two-pass regime of a bf16 plan: the full second pass leaves bf16 records (rectangular layout)
        dim3 block(256);
        if (mode == 0) {   // wide plans, this rank's rows against other ranks' columns: every tile's record, rectangular layout
            if (kcols) LAUNCH((fwd_sums_kernel<bf16_t, true, 0, true>), grid, block, stream, (const bf16_t*)rows, (const bf16_t*)cols, g, tps, out, kcols, shift, stash, (int*)nullptr, (float*)nullptr);
            else LAUNCH((fwd_sums_kernel<bf16_t, false, 0, true>), grid, block, stream, (const bf16_t*)rows, (const bf16_t*)cols, g, tps, out, kcols, shift, stash, (int*)nullptr, (float*)nullptr);
            return launch_status("fwd_sums_kernel (rectangular, save, bf16 records)");
        }
        if (mode != 2) return fail(CROSSCLR_E_ARG, "bf16 plans save through the generic forward in modes 0 (rectangular) and 2 (two-pass)");
        if (kcols) LAUNCH((fwd_sums_kernel<bf16_t, true, 2, true>), grid, block, stream, (const bf16_t*)rows, (const bf16_t*)cols, g, tps, out, kcols, shift, stash, (int*)nullptr, const_cast<float*>(shift_cols));
        else LAUNCH((fwd_sums_kernel<bf16_t, false, 2, true>), grid, block, stream, (const bf16_t*)rows, (const bf16_t*)cols, g, tps, out, kcols, shift, stash, (int*)nullptr, const_cast<float*>(shift_cols));
        return launch_status("fwd_sums_kernel (save, bf16 records)");
    }
    if (stash) {   // exact-fp32 forward that also saves its exponentials (local block; common shift, or per-row shifts: mode 2)
        dim3 block(256);
#define CROSSCLR_LSV(SW, MODE) LAUNCH((fwd_sums_kernel<float, SW, MODE, true>), grid, block, stream, (const float*)rows, (const float*)cols, g, tps, out, kcols, shift, stash, (int*)nullptr, const_cast<float*>(mode == 2 ? shift_cols : nullptr))
        if (mode == 2) { if (kcols) CROSSCLR_LSV(true, 2); else CROSSCLR_LSV(false, 2); }
        else { if (kcols) CROSSCLR_LSV(true, 0); else CROSSCLR_LSV(false, 0); }
#undef CROSSCLR_LSV
        return launch_status("fwd_sums_kernel (save)");
    }
    if (plan->mode == CROSSCLR_MODE_FP32) forward_generic_t<float>(plan, g, rows, cols, out, kcols, shift, mode, tps, grid, stream);
    else forward_generic_t<bf16_t>(plan, g, rows, cols, out, kcols, shift, mode, tps, grid, stream);
    return launch_status("fwd_sums_kernel");
}

extern "C" int crossclr_forward(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols,
                                int col_ranks, int col_rank0, int skip_rank, float temperature,
                                float negative_weight, float* part, int slot0, void* stream) {
    return crossclr_forward_w(plan, xhat_rows, xhat_cols, col_ranks, col_rank0, skip_rank, temperature, negative_weight,
                              nullptr, part, slot0, stream);
}

extern "C" int crossclr_forward_w(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols,
                                  int col_ranks, int col_rank0, int skip_rank, float temperature,
                                  float negative_weight, const crossclr_sample_weights* sw, float* part, int slot0,
                                  void* stream) {
    if (!plan || !xhat_rows || !xhat_cols || !part || slot0 < 0) return fail(CROSSCLR_E_ARG, "NULL/negative argument");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = make_geo(plan, col_ranks, col_rank0, skip_rank, temperature, negative_weight, &g);
    if (rc) return rc;
    if (plan->fwd_slots <= 0 || slot0 % plan->fwd_slots != 0 || slot0 / plan->fwd_slots >= kLaunchGroups)
        return fail(CROSSCLR_E_ARG, "slot0 must be L * plan->fwd_slots, L = 0..%d", kLaunchGroups - 1);
    float* out = part + (size_t)slot0 * 2 * plan->bpad;
    int* header = reinterpret_cast<int*>(part + ws_flag_off(plan)) + 4 * (slot0 / plan->fwd_slots);
#ifndef CROSSCLR_NO_FAST
    if (plan->fast_path) {
        // rows and columns are the same packed operand (the single-GPU case and the local block of a
        // sharded run): evaluate only the upper triangle of the symmetric matrix
        const bool symmetric = xhat_rows == xhat_cols && col_ranks == 1 && col_rank0 == plan->rank && skip_rank < 0 &&
                               !env_knobs().disable_symmetric;
        const bool skipping = skip_rank >= col_rank0 && skip_rank < col_rank0 + col_ranks;
        if (col_ranks - (skipping ? 1 : 0) <= 0) {
            // nothing to do (every column rank is skipped): leave a dense, all-zero launch behind
            rc = device_zero_header(header, stream);
            if (rc) return rc;
            return device_zero(out, (size_t)plan->fwd_slots * 2 * plan->bpad * sizeof(float), stream);
        }
        rc = fast_forward(plan, g, xhat_rows, xhat_cols, out, part + ws_colpart_off(plan), header, symmetric, krows, kcols, stream);
        return rc ? fail(rc, "fast_forward: unsupported Dpad %d", plan->Dpad) : launch_status("fast_fwd_kernel");
    }
#endif
    if (xhat_rows == xhat_cols && col_ranks == 1 && col_rank0 == plan->rank && skip_rank < 0 && krows == kcols &&
        !env_knobs().disable_symmetric) {
        // the local block (single device, or the local block of a sharded run), forward only: upper triangle + column sums
        float* colpart = part + ws_colpart_off(plan);
        return plan->mode == CROSSCLR_MODE_FP32 ? forward_generic_sym<float>(plan, g, xhat_rows, out, kcols, colpart, header, stream)
                                                : forward_generic_sym<bf16_t>(plan, g, xhat_rows, out, kcols, colpart, header, stream);
    }
    rc = device_zero_header(header, stream);
    if (rc) return rc;
    return forward_generic(plan, g, xhat_rows, xhat_cols, out, kcols, nullptr, 0, stream);
}

extern "C" int crossclr_forward_save(const crossclr_plan* plan, const void* xhat, float temperature, float negative_weight,
                                     const crossclr_sample_weights* sw, float* part, int slot0, void* stash, void* stream) {
    if (!plan || !xhat || !part || !stash || slot0 < 0) return fail(CROSSCLR_E_ARG, "NULL/negative argument");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_forward_save needs the register-resident path");
#else
    if (!plan->stash_bytes) return fail(CROSSCLR_E_ARG, "this plan has no save-for-backward path (stash_bytes == 0)");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    if (krows != kcols) return fail(CROSSCLR_E_ARG, "the local block's row and column negative scales are the same array");
    Geo g;
    int rc = make_geo(plan, 1, plan->rank, -1, temperature, negative_weight, &g);
    if (rc) return rc;
    if (plan->fwd_slots <= 0 || slot0 % plan->fwd_slots != 0 || slot0 / plan->fwd_slots >= kLaunchGroups)
        return fail(CROSSCLR_E_ARG, "slot0 must be L * plan->fwd_slots, L = 0..%d", kLaunchGroups - 1);
    float* out = part + (size_t)slot0 * 2 * plan->bpad;
    int* header = reinterpret_cast<int*>(part + ws_flag_off(plan)) + 4 * (slot0 / plan->fwd_slots);
    if (!plan->fast_path && plan->mode == CROSSCLR_MODE_BF16)   // wide bf16 plan: upper triangle, bf16 records in the register-resident layout
        return forward_generic_sym<bf16_t>(plan, g, xhat, out, kcols, part + ws_colpart_off(plan), header, stream, static_cast<float*>(stash));
    if (!plan->fast_path) {   // exact-fp32 mode
        if (!env_knobs().disable_symmetric)   // upper triangle; every fragment stored twice (as evaluated + transposed)
            return forward_generic_sym<float>(plan, g, xhat, out, kcols, part + ws_colpart_off(plan), header, stream, static_cast<float*>(stash));
        rc = device_zero_header(header, stream);
        if (rc) return rc;
        return forward_generic(plan, g, xhat, xhat, out, kcols, nullptr, 0, stream, static_cast<float*>(stash));
    }
    rc = fast_forward_save(plan, g, xhat, out, part + ws_colpart_off(plan), header, krows, stash, stream);
    return rc ? fail(rc, "fast_forward_save: unsupported Dpad %d", plan->Dpad) : launch_status("fast_fwd_kernel (save)");
#endif
}

extern "C" int crossclr_backward_saved(const crossclr_plan* plan, const void* xhat, const void* stash, float temperature,
                                       float negative_weight, const float* rz, const float* wrz,
                                       const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream) {
    if (!plan || !xhat || !stash || !rz || !wrz || !gbuf) return fail(CROSSCLR_E_ARG, "NULL argument");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_backward_saved needs the register-resident path");
#else
    if (!plan->stash_bytes) return fail(CROSSCLR_E_ARG, "this plan has no save-for-backward path (stash_bytes == 0)");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    if (krows != kcols) return fail(CROSSCLR_E_ARG, "the local block's row and column negative scales are the same array");
    Geo g;
    int rc = make_geo(plan, 1, plan->rank, -1, temperature, negative_weight, &g);
    if (rc) return rc;
    if (!plan->fast_path && plan->mode == CROSSCLR_MODE_FP32) {   // exact-fp32 mode
        const int NQ = 2 * plan->bpad / 32;
        const int tps = (NQ + plan->bwd_slices - 1) / plan->bwd_slices;
        const unsigned rb = 2 * plan->bpad / 64, nz = (unsigned)plan->bwd_slices;
        dim3 block(256);
#define CROSSCLR_LS32(DC)                                                                                                              \
    do {                                                                                                                               \
        if (krows) LAUNCH((bwd_saved32_kernel<DC, true>), dim3(rb, plan->Dpad / DC, nz), block, stream, (const float*)xhat,            \
                          (const float*)stash, g, rz, wrz, gbuf, accumulate, tps, krows, kNoF, kNoF, kNoF);                                               \
        else LAUNCH((bwd_saved32_kernel<DC, false>), dim3(rb, plan->Dpad / DC, nz), block, stream, (const float*)xhat,                 \
                    (const float*)stash, g, rz, wrz, gbuf, accumulate, tps, krows, kNoF, kNoF, kNoF);                                                     \
    } while (0)
        if (plan->Dpad % 256 == 0) CROSSCLR_LS32(256);
        else if (plan->Dpad % 128 == 0) CROSSCLR_LS32(128);
        else CROSSCLR_LS32(64);
#undef CROSSCLR_LS32
        return launch_status("bwd_saved32_kernel");
    }
    rc = fast_backward_saved(plan, g, xhat, stash, rz, wrz, rz, wrz, gbuf, accumulate, krows, krows, 0, stream);
    return rc ? fail(rc, "fast_backward_saved: unsupported Dpad %d", plan->Dpad) : launch_status("fast_bwd_dsl_kernel");
#endif
}

extern "C" int crossclr_backward_saved_xf(const crossclr_plan* plan, const void* xhat_xf, const void* stash, float temperature,
                                          float negative_weight, const float* rz, const float* wrz,
                                          const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream) {
    if (!plan || !xhat_xf || !stash || !rz || !wrz || !gbuf) return fail(CROSSCLR_E_ARG, "NULL argument");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_backward_saved_xf needs the register-resident path");
#else
    if (!plan->stash_bytes || !plan->xf_bytes) return fail(CROSSCLR_E_ARG, "this plan has no fragment-major saved backward (stash_bytes / xf_bytes == 0)");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    if (krows != kcols) return fail(CROSSCLR_E_ARG, "the local block's row and column negative scales are the same array");
    Geo g;
    int rc = make_geo(plan, 1, plan->rank, -1, temperature, negative_weight, &g);
    if (rc) return rc;
    rc = fast_backward_saved(plan, g, xhat_xf, stash, rz, wrz, rz, wrz, gbuf, accumulate, krows, krows, 3, stream);
    return rc ? fail(rc, "fast_backward_saved (xf): unsupported Dpad %d", plan->Dpad) : launch_status("fast_bwd_dsl_kernel (xf)");
#endif
}

extern "C" int crossclr_backward_saved_xfp(const crossclr_plan* plan, const void* xhat_xf, const void* stash, float temperature,
                                           float negative_weight, const float* rz, const float* wrz,
                                           const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream) {
    if (!plan || !xhat_xf || !stash || !rz || !wrz || !gbuf) return fail(CROSSCLR_E_ARG, "NULL argument");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_backward_saved_xfp needs the register-resident path");
#else
    if (!plan->stash_bytes || !plan->xf_bytes) return fail(CROSSCLR_E_ARG, "this plan has no fragment-major saved backward (stash_bytes / xf_bytes == 0)");
    if (plan->stash_bytes >= ((size_t)1 << 32))
        return fail(CROSSCLR_E_ARG, "crossclr_backward_saved_xfp addresses the saved exponentials with 32-bit offsets (stash of %zu bytes): use crossclr_backward_saved_xf", plan->stash_bytes);
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    if (krows != kcols) return fail(CROSSCLR_E_ARG, "the local block's row and column negative scales are the same array");
    Geo g;
    int rc = make_geo(plan, 1, plan->rank, -1, temperature, negative_weight, &g);
    if (rc) return rc;
    rc = fast_backward_saved(plan, g, xhat_xf, stash, rz, wrz, rz, wrz, gbuf, accumulate, krows, krows, 4, stream);
    return rc ? fail(rc, "fast_backward_saved (xfp): unsupported Dpad %d", plan->Dpad) : launch_status("fast_bwd_xfp_kernel");
#endif
}

extern "C" int crossclr_forward_pairs(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_all, int first_rank,
                                      int nranks, float temperature, float negative_weight,
                                      const crossclr_sample_weights* sw, float* part, int slot0, float* colsum_out,
                                      void* stream) {
    if (!plan || !xhat_rows || !xhat_all || !part || !colsum_out) return fail(CROSSCLR_E_ARG, "NULL argument");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_forward_pairs needs the register-resident path");
#else
    if (!plan->fast_path) return fail(CROSSCLR_E_ARG, "crossclr_forward_pairs needs the register-resident bf16 path");
    if (first_rank < 0 || first_rank >= plan->world || nranks < 1 || nranks > (plan->world - 1) / 2)
        return fail(CROSSCLR_E_ARG, "bad first_rank/nranks %d/%d for world %d", first_rank, nranks, plan->world);
    if (plan->fwd_slots <= 0 || slot0 < 0 || slot0 % plan->fwd_slots != 0 || slot0 / plan->fwd_slots >= kLaunchGroups)
        return fail(CROSSCLR_E_ARG, "slot0 must be L * plan->fwd_slots, L = 0..%d", kLaunchGroups - 1);
    for (int i = 0; i < nranks; ++i)
        if ((first_rank + i) % plan->world == plan->rank) return fail(CROSSCLR_E_ARG, "the pair range must not contain this rank");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = make_geo(plan, nranks, first_rank, -1, temperature, negative_weight, &g);
    if (rc) return rc;
    g.col_wrap = plan->world;
    float* out = part + (size_t)slot0 * 2 * plan->bpad;
    int* header = reinterpret_cast<int*>(part + ws_flag_off(plan)) + 4 * (slot0 / plan->fwd_slots);
    float* colpart = part + ws_paircol_off(plan);
    rc = fast_forward(plan, g, xhat_rows, xhat_all, out, colpart, header, false, krows, kcols, stream, true);
    if (rc) return fail(rc, "fast_forward: unsupported Dpad %d", plan->Dpad);
    const int nrb = 2 * plan->bpad / (32 * fast_fwd_tpr(plan->Dpad));
    const int ncols = nranks * 2 * plan->bpad;
    LAUNCH(colsum_reduce_kernel, dim3((ncols + 255) / 256), dim3(256), stream, (const float*)colpart, nrb, ncols, colsum_out);
    return launch_status("fast_fwd_kernel (pairs)");
#endif
}

extern "C" int crossclr_forward_add(const crossclr_plan* plan, float* part, int slot0, const float* vec, void* stream) {
    if (!plan || !part) return fail(CROSSCLR_E_ARG, "NULL argument");
    if (plan->fwd_slots <= 0 || slot0 < 0 || slot0 % plan->fwd_slots != 0 || slot0 / plan->fwd_slots >= kLaunchGroups)
        return fail(CROSSCLR_E_ARG, "slot0 must be L * plan->fwd_slots, L = 0..%d", kLaunchGroups - 1);
    const int n = 2 * plan->bpad;
    LAUNCH(fwd_add_kernel, dim3((n + 255) / 256), dim3(256), stream, vec, n, part + (size_t)slot0 * n,
           reinterpret_cast<int*>(part + ws_flag_off(plan)) + 4 * (slot0 / plan->fwd_slots));
    return launch_status("fwd_add_kernel");
}

extern "C" int crossclr_forward_finish(const crossclr_plan* plan, const float* part, int nslots,
                                       const float* diag_cos, float temperature, float negative_weight,
                                       float* logz, float* rz, float* wrz, double* loss_sum, void* stream) {
    return crossclr_forward_finish_w(plan, part, nslots, diag_cos, temperature, negative_weight, nullptr, logz, rz, wrz,
                                     loss_sum, stream);
}

extern "C" int crossclr_forward_finish_w(const crossclr_plan* plan, const float* part, int nslots,
                                         const float* diag_cos, float temperature, float negative_weight,
                                         const crossclr_sample_weights* sw, float* logz, float* rz, float* wrz,
                                         double* loss_sum, void* stream) {
    if (!plan || !part || !diag_cos || !logz || !rz || !wrz || !loss_sum || plan->fwd_slots <= 0 || nslots <= 0 ||
        nslots % plan->fwd_slots != 0 || nslots / plan->fwd_slots > kLaunchGroups)
        return fail(CROSSCLR_E_ARG, "NULL argument / nslots must be fwd_slots times the number of launch groups (1..%d)", kLaunchGroups);
    Geo g;
    int rc = make_geo(plan, 1, plan->rank, -1, temperature, negative_weight, &g);
    if (rc) return rc;
    const int nb = plan->loss_ws_doubles - 1;
    const int nlaunch = nslots / plan->fwd_slots;
    return crossclr_forward_finish_s(plan, part, nslots, diag_cos, temperature, negative_weight, sw, nullptr, logz, rz, wrz, loss_sum,
                                     stream);
}

static int forward_finish_impl(const crossclr_plan* plan, const float* part, int nslots,
                               const float* diag_cos, float temperature, float negative_weight,
                               const crossclr_sample_weights* sw, const float* shift_rows, float* logz, float* rz,
                               float* wrz, double* loss_sum, void* stream, int* ticket);
extern "C" int crossclr_forward_finish_s(const crossclr_plan* plan, const float* part, int nslots,
                                         const float* diag_cos, float temperature, float negative_weight,
                                         const crossclr_sample_weights* sw, const float* shift_rows, float* logz, float* rz,
                                         float* wrz, double* loss_sum, void* stream) {
    return forward_finish_impl(plan, part, nslots, diag_cos, temperature, negative_weight, sw, shift_rows, logz, rz, wrz, loss_sum, stream, nullptr);
}
// ticket != NULL (crossclr_step_forward; an int the step's first kernel has cleared): the finish kernel's last block forms the sum -- one launch
static int forward_finish_impl(const crossclr_plan* plan, const float* part, int nslots,
                               const float* diag_cos, float temperature, float negative_weight,
                               const crossclr_sample_weights* sw, const float* shift_rows, float* logz, float* rz,
                               float* wrz, double* loss_sum, void* stream, int* ticket) {
    if (!plan || !part || !diag_cos || !logz || !rz || !wrz || !loss_sum || plan->fwd_slots <= 0 || nslots <= 0 ||
        nslots % plan->fwd_slots != 0 || nslots / plan->fwd_slots > kLaunchGroups)
        return fail(CROSSCLR_E_ARG, "NULL argument / nslots must be fwd_slots times the number of launch groups (1..%d)", kLaunchGroups);
    Geo g;
    int rc = make_geo(plan, 1, plan->rank, -1, temperature, negative_weight, &g, shift_rows != nullptr);
    if (rc) return rc;
    if ((g.row_shift != 0) != (shift_rows != nullptr))
        return fail(CROSSCLR_E_ARG, "shift_rows must be given exactly when crossclr_needs_row_shift(temperature, negative_weight)");
    const int nb = plan->loss_ws_doubles - 1;
    const int nlaunch = nslots / plan->fwd_slots;
    LAUNCH(fwd_finish_kernel, dim3(nb), dim3(256), stream, part, nlaunch, plan->fwd_slots, g, diag_cos, 1.0f / temperature,
           negative_weight, logz, rz, wrz, loss_sum, part + ws_colpart_off(plan),
           reinterpret_cast<const int*>(part + ws_flag_off(plan)), sw ? sw->neg_scale_rows : nullptr,
           sw ? sw->loss_weight : nullptr, shift_rows, ticket, 1.0 / (2.0 * (double)plan->b * (double)plan->world));
    if (!ticket) LAUNCH(fwd_finish_reduce_kernel, dim3(1), dim3(64), stream, loss_sum, nb, 1.0 / (2.0 * (double)plan->b * (double)plan->world));
    return launch_status("fwd_finish_kernel");
}

// ---- two-pass soft-max for small temperatures (max |logit| > 128) ------------------------------------------------------
extern "C" int crossclr_needs_row_shift(float temperature, float negative_weight) {
    return (temperature > 0.f && isfinite(temperature) && isfinite(negative_weight) && needs_row_shift(temperature, negative_weight)) ? 1 : 0;
}

extern "C" int crossclr_forward_rowmax(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols, int col_ranks,
                                       int col_rank0, int skip_rank, float temperature, float negative_weight,
                                       const crossclr_sample_weights* sw, float* part, float* shift_rows, int accumulate,
                                       void* stream) {
    if (!plan || !xhat_rows || !xhat_cols || !part || !shift_rows) return fail(CROSSCLR_E_ARG, "NULL argument");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = make_geo(plan, col_ranks, col_rank0, skip_rank, temperature, negative_weight, &g, true);
    if (rc) return rc;
    const bool skipping = skip_rank >= col_rank0 && skip_rank < col_rank0 + col_ranks;
    const int n = 2 * plan->bpad;
    int nslots = plan->fwd_slots;
    const float* colpart = nullptr;
    if (col_ranks - (skipping ? 1 : 0) <= 0) nslots = 0;   // nothing to look at: only the self pair / the previous value
    else if (xhat_rows == xhat_cols && col_ranks == 1 && col_rank0 == plan->rank && skip_rank < 0 && krows == kcols &&
             !env_knobs().disable_symmetric) {   // the local block: upper triangle, column maxima for the mirrored tiles
        float* cp = part + ws_colpart_off(plan);
        int* header = reinterpret_cast<int*>(part + ws_flag_off(plan));   // (launch group 0's header: rewritten by the pass that follows)
        rc = plan->mode == CROSSCLR_MODE_FP32 ? forward_generic_sym<float>(plan, g, xhat_rows, part, kcols, cp, header, stream, nullptr, nullptr, true)
                                              : forward_generic_sym<bf16_t>(plan, g, xhat_rows, part, kcols, cp, header, stream, nullptr, nullptr, true);
        if (rc) return rc;
        colpart = cp;
    } else if ((rc = forward_generic(plan, g, xhat_rows, xhat_cols, part, kcols, nullptr, 1, stream))) return rc;
    LAUNCH(rowmax_combine_kernel, dim3((n + 255) / 256), dim3(256), stream, (const float*)part, nslots, n, krows, accumulate, shift_rows, colpart);
    return launch_status("rowmax_combine_kernel");
}

extern "C" int crossclr_forward_s(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols, int col_ranks,
                                  int col_rank0, int skip_rank, float temperature, float negative_weight,
                                  const crossclr_sample_weights* sw, const float* shift_rows, float* part, int slot0, void* stream) {
    if (!shift_rows)
        return crossclr_forward_w(plan, xhat_rows, xhat_cols, col_ranks, col_rank0, skip_rank, temperature, negative_weight, sw, part,
                                  slot0, stream);
    if (!plan || !xhat_rows || !xhat_cols || !part || slot0 < 0) return fail(CROSSCLR_E_ARG, "NULL/negative argument");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = make_geo(plan, col_ranks, col_rank0, skip_rank, temperature, negative_weight, &g, true);
    if (rc) return rc;
    if (plan->fwd_slots <= 0 || slot0 % plan->fwd_slots != 0 || slot0 / plan->fwd_slots >= kLaunchGroups)
        return fail(CROSSCLR_E_ARG, "slot0 must be L * plan->fwd_slots, L = 0..%d", kLaunchGroups - 1);
    float* out = part + (size_t)slot0 * 2 * plan->bpad;
    int* header = reinterpret_cast<int*>(part + ws_flag_off(plan)) + 4 * (slot0 / plan->fwd_slots);
    rc = device_zero_header(header, stream);
    if (rc) return rc;
    const bool skipping = skip_rank >= col_rank0 && skip_rank < col_rank0 + col_ranks;
    if (col_ranks - (skipping ? 1 : 0) <= 0) return device_zero(out, (size_t)plan->fwd_slots * 2 * plan->bpad * sizeof(float), stream);
    if (xhat_rows == xhat_cols && col_ranks == 1 && col_rank0 == plan->rank && skip_rank < 0 && krows == kcols &&
        !env_knobs().disable_symmetric) {   // the local block: upper triangle + column sums (two exponentials per element)
        float* colpart = part + ws_colpart_off(plan);
        return plan->mode == CROSSCLR_MODE_FP32
                   ? forward_generic_sym<float>(plan, g, xhat_rows, out, kcols, colpart, header, stream, nullptr, shift_rows)
                   : forward_generic_sym<bf16_t>(plan, g, xhat_rows, out, kcols, colpart, header, stream, nullptr, shift_rows);
    }
    return forward_generic(plan, g, xhat_rows, xhat_cols, out, kcols, shift_rows, 2, stream);
}

// the two-pass regime's save-for-backward pair (exact-fp32 plans, local block): U and Ut, twice the single-pass stash
// bf16 register-resident plans in the two-pass regime: the FULL matrix of bf16 records U[p][q] = exp2(x - shift[p]) in the rectangular layout
// of a one-rank remote block, followed by 2 bpad floats of zeros (crossclr_backward_saved_s passes them as the statistics of the side a
// launch must not weigh: W = U rz_p + U^T rz_q is formed as two launches of the saved backward, direct and transposed)
static size_t rect_bytes_s(const crossclr_plan* plan) {
#ifdef CROSSCLR_NO_FAST
    (void)plan; return 0;
#else
    if (!plan || plan->mode != CROSSCLR_MODE_BF16 || !plan->stash_bytes) return 0;
    if (plan->fast_path) return fast_stash_bytes_rect(plan->bpad, plan->Dpad, 1);
    return plan->Dpad > 1024 ? wide_stash_bytes_rect(plan->bpad, 1) : 0;      // wide plans: the generic second pass writes U AND Ut (below)
#endif
}
// wide bf16 plans (Dpad > 1024) have no transposed launch of the saved backward: their second pass writes Ut behind U (like a remote block's)
// and the backward is two DIRECT launches, over U with the rows' statistics and over Ut with the columns'
static bool wide_two_pass(const crossclr_plan* plan) { return plan && plan->mode == CROSSCLR_MODE_BF16 && !plan->fast_path && plan->Dpad > 1024; }
static size_t stash_bytes_s(const crossclr_plan* plan) {
    if (!plan || !plan->stash_bytes) return 0;
    if (plan->mode == CROSSCLR_MODE_BF16) {
        const size_t rb = rect_bytes_s(plan);
        return rb ? (wide_two_pass(plan) ? 2 : 1) * rb + (size_t)2 * plan->bpad * 4 : 0;
    }
    if (plan->fast_path || plan->mode != CROSSCLR_MODE_FP32) return 0;
    return 2 * plan->stash_bytes <= ((size_t)16 << 30) ? 2 * plan->stash_bytes : 0;
}
extern "C" size_t crossclr_stash_bytes_s(const crossclr_plan* plan) { return stash_bytes_s(plan); }

extern "C" int crossclr_forward_save_s(const crossclr_plan* plan, const void* xhat, float temperature, float negative_weight,
                                       const crossclr_sample_weights* sw, const float* shift, float* part, int slot0, void* stash,
                                       void* stream) {
    if (!plan || !xhat || !shift || !part || !stash || slot0 < 0) return fail(CROSSCLR_E_ARG, "NULL/negative argument");
    if (!stash_bytes_s(plan)) return fail(CROSSCLR_E_ARG, "this plan has no two-pass save-for-backward path (crossclr_stash_bytes_s == 0)");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    if (krows != kcols) return fail(CROSSCLR_E_ARG, "the local block's row and column negative scales are the same array");
    Geo g;
    int rc = make_geo(plan, 1, plan->rank, -1, temperature, negative_weight, &g, true);
    if (rc) return rc;
    if (plan->fwd_slots <= 0 || slot0 % plan->fwd_slots != 0 || slot0 / plan->fwd_slots >= kLaunchGroups)
        return fail(CROSSCLR_E_ARG, "slot0 must be L * plan->fwd_slots, L = 0..%d", kLaunchGroups - 1);
    float* out = part + (size_t)slot0 * 2 * plan->bpad;
    int* header = reinterpret_cast<int*>(part + ws_flag_off(plan)) + 4 * (slot0 / plan->fwd_slots);
    if (plan->mode == CROSSCLR_MODE_BF16) {   // full (non-symmetric) second pass: bf16 records + the zero statistics behind them
        rc = device_zero_header(header, stream);
        if (rc) return rc;
        rc = device_zero(static_cast<unsigned char*>(stash) + (wide_two_pass(plan) ? 2 : 1) * rect_bytes_s(plan), (size_t)2 * plan->bpad * 4, stream);
        if (rc) return rc;
        return forward_generic(plan, g, xhat, xhat, out, kcols, shift, 2, stream, static_cast<float*>(stash), wide_two_pass(plan) ? shift : nullptr);
    }
    if (!env_knobs().disable_symmetric)
        return forward_generic_sym<float>(plan, g, xhat, out, kcols, part + ws_colpart_off(plan), header, stream, static_cast<float*>(stash), shift);
    rc = device_zero_header(header, stream);
    if (rc) return rc;
    return forward_generic(plan, g, xhat, xhat, out, kcols, shift, 2, stream, static_cast<float*>(stash));
}

extern "C" int crossclr_backward_saved_s(const crossclr_plan* plan, const void* xhat, const void* stash, float temperature,
                                         float negative_weight, const float* rz, const float* wrz,
                                         const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream) {
    if (!plan || !xhat || !stash || !rz || !wrz || !gbuf) return fail(CROSSCLR_E_ARG, "NULL argument");
    if (!stash_bytes_s(plan)) return fail(CROSSCLR_E_ARG, "this plan has no two-pass save-for-backward path (crossclr_stash_bytes_s == 0)");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    if (krows != kcols) return fail(CROSSCLR_E_ARG, "the local block's row and column negative scales are the same array");
    Geo g;
    int rc = make_geo(plan, 1, plan->rank, -1, temperature, negative_weight, &g, true);
    if (rc) return rc;
#ifndef CROSSCLR_NO_FAST
    if (plan->mode == CROSSCLR_MODE_BF16) {
        // W[p][q] = U[p][q] rz_p + U[q][p] rz_q: the direct launch weighs with the ROW statistics only (column side: zeros), the transposed
        // launch with the statistics of the rows it contracts over (output side: zeros) and accumulates -- two 8 B^2 D launches of the saved
        // backward instead of the 16 B^2 D (1 + ...) recompute of the generic kernel
        if (wide_two_pass(plan)) {   // U rz_p from the first array, Ut rz_q from the second: two direct launches of the D-slice kernel in column parts
            const unsigned char* U = static_cast<const unsigned char*>(stash);
            const size_t rb = rect_bytes_s(plan);
            const float* zw = reinterpret_cast<const float*>(U + 2 * rb);
            rc = fast_backward_saved(plan, g, xhat, U, rz, wrz, zw, zw, gbuf, accumulate, krows, krows, 1, stream);
            if (rc) return fail(rc, "fast_backward_saved (two-pass, wide, rows' side): unsupported Dpad %d", plan->Dpad);
            rc = fast_backward_saved(plan, g, xhat, U + rb, zw, zw, rz, wrz, gbuf, 1, krows, krows, 1, stream);
            return rc ? fail(rc, "fast_backward_saved (two-pass, wide, columns' side): unsupported Dpad %d", plan->Dpad)
                      : launch_status("fast_bwd_dsl_kernel (two-pass pair, wide)");
        }
        const float* zeros = reinterpret_cast<const float*>(static_cast<const unsigned char*>(stash) + rect_bytes_s(plan));
        rc = fast_backward_saved(plan, g, xhat, stash, rz, wrz, zeros, zeros, gbuf, accumulate, krows, krows, 1, stream);
        if (rc) return fail(rc, "fast_backward_saved (two-pass, direct): unsupported Dpad %d", plan->Dpad);
        Geo gt = g;
        gt.col_ranks = 1; gt.skip_rank = 0; gt.col_rank0 = plan->rank; gt.col_wrap = 0; gt.row_rank = plan->rank;
        rc = fast_backward_saved(plan, gt, xhat, stash, zeros, zeros, rz, wrz, gbuf, 1, krows, krows, 2, stream);
        return rc ? fail(rc, "fast_backward_saved (two-pass, transposed): unsupported Dpad %d", plan->Dpad) : launch_status("fast_bwd_dsl_kernel (two-pass pair)");
    }
#endif
    const int NQ = 2 * plan->bpad / 32;
    const int tps = (NQ + plan->bwd_slices - 1) / plan->bwd_slices;
    const unsigned rb = 2 * plan->bpad / 64, nz = (unsigned)plan->bwd_slices;
    dim3 block(256);
#define CROSSCLR_LS32R(DC)                                                                                                             \
    do {                                                                                                                               \
        if (krows) LAUNCH((bwd_saved32_kernel<DC, true, true>), dim3(rb, plan->Dpad / DC, nz), block, stream, (const float*)xhat,      \
                          (const float*)stash, g, rz, wrz, gbuf, accumulate, tps, krows, kNoF, kNoF, kNoF);                                               \
        else LAUNCH((bwd_saved32_kernel<DC, false, true>), dim3(rb, plan->Dpad / DC, nz), block, stream, (const float*)xhat,           \
                    (const float*)stash, g, rz, wrz, gbuf, accumulate, tps, krows, kNoF, kNoF, kNoF);                                                     \
    } while (0)
    if (plan->Dpad % 256 == 0) CROSSCLR_LS32R(256);
    else if (plan->Dpad % 128 == 0) CROSSCLR_LS32R(128);
    else CROSSCLR_LS32R(64);
#undef CROSSCLR_LS32R
    return launch_status("bwd_saved32_kernel (two-pass)");
}

// ------------------------------------------------------------------------------------------------
template <typename T>
static int backward_generic(const crossclr_plan* p, const Geo& g, const void* rows, const void* cols,
                            const float* rz_rows, const float* wrz_rows, const float* rz_cols,
                            const float* wrz_cols, float* gbuf, int accumulate, const float* krows, const float* kcols,
                            const float* shift_rows, const float* shift_cols, void* stream) {
    dim3 block(256);
    const int rb = 2 * p->bpad / 64;
    const bool skipping = g.skip_rank >= g.col_rank0 && g.skip_rank < g.col_rank0 + g.col_ranks;
    const int ntiles = (g.col_ranks - (skipping ? 1 : 0)) * 2 * p->bpad / 64;   // usable column tiles
    const int tps = (ntiles + p->bwd_slices - 1) / p->bwd_slices;
    const unsigned nz = (unsigned)p->bwd_slices;
    // the generic backward slices D by DC and gbuf by plan->bwd_slices column slices; a plan made for the register-resident
    // kernels has the slice count of THEIR tiling, which is a valid (just not tuned) slice count here too
#define CROSSCLR_LB2(DC, SW, RM) LAUNCH((bwd_kernel<T, DC, SW, RM>), dim3(rb, p->Dpad / DC, nz), block, stream, (const T*)rows, (const T*)cols, g, \
                                        rz_rows, wrz_rows, rz_cols, wrz_cols, gbuf, accumulate, tps, krows, kcols, shift_rows, shift_cols)
#define CROSSCLR_LB(DC)                                          \
    do {                                                          \
        if (shift_rows) { if (krows) CROSSCLR_LB2(DC, true, true); else CROSSCLR_LB2(DC, false, true); }   \
        else { if (krows) CROSSCLR_LB2(DC, true, false); else CROSSCLR_LB2(DC, false, false); }            \
    } while (0)
    if (p->Dpad % 512 == 0 && !env_knobs().bwd_dc256) CROSSCLR_LB(512);
    else if (p->Dpad % 256 == 0) CROSSCLR_LB(256);
    else if (p->Dpad % 128 == 0) CROSSCLR_LB(128);
    else CROSSCLR_LB(64);
#undef CROSSCLR_LB
#undef CROSSCLR_LB2
    return launch_status("bwd_kernel");
}

extern "C" int crossclr_backward(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols,
                                 int col_ranks, int col_rank0, int skip_rank, float temperature,
                                 float negative_weight, const float* rz_rows, const float* wrz_rows,
                                 const float* rz_cols, const float* wrz_cols, float* gbuf, int accumulate,
                                 void* stream) {
    return crossclr_backward_w(plan, xhat_rows, xhat_cols, col_ranks, col_rank0, skip_rank, temperature, negative_weight,
                               rz_rows, wrz_rows, rz_cols, wrz_cols, nullptr, gbuf, accumulate, stream);
}

extern "C" int crossclr_backward_w(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols,
                                   int col_ranks, int col_rank0, int skip_rank, float temperature,
                                   float negative_weight, const float* rz_rows, const float* wrz_rows,
                                   const float* rz_cols, const float* wrz_cols, const crossclr_sample_weights* sw,
                                   float* gbuf, int accumulate, void* stream) {
    if (!plan || !xhat_rows || !xhat_cols || !rz_rows || !wrz_rows || !rz_cols || !wrz_cols || !gbuf)
        return fail(CROSSCLR_E_ARG, "NULL argument");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = make_geo(plan, col_ranks, col_rank0, skip_rank, temperature, negative_weight, &g);
    if (rc) return rc;
#ifndef CROSSCLR_NO_FAST
    if (plan->fast_bwd) {
        rc = plan->fast_bwd == 2
                 ? fast_backward16(plan, g, xhat_rows, xhat_cols, rz_rows, wrz_rows, rz_cols, wrz_cols, gbuf, accumulate, krows, kcols, stream)
                 : fast_backward(plan, g, xhat_rows, xhat_cols, rz_rows, wrz_rows, rz_cols, wrz_cols, gbuf, accumulate, krows, kcols, stream);
        return rc ? fail(rc, "fast backward: unsupported Dpad %d", plan->Dpad) : launch_status("fast_bwd_kernel");
    }
#endif
    if (plan->mode == CROSSCLR_MODE_FP32)
        return backward_generic<float>(plan, g, xhat_rows, xhat_cols, rz_rows, wrz_rows, rz_cols, wrz_cols, gbuf, accumulate, krows, kcols, nullptr, nullptr, stream);
    return backward_generic<bf16_t>(plan, g, xhat_rows, xhat_cols, rz_rows, wrz_rows, rz_cols, wrz_cols, gbuf, accumulate, krows, kcols, nullptr, nullptr, stream);
}

extern "C" int crossclr_backward_s(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_cols, int col_ranks,
                                   int col_rank0, int skip_rank, float temperature, float negative_weight,
                                   const float* rz_rows, const float* wrz_rows, const float* rz_cols, const float* wrz_cols,
                                   const crossclr_sample_weights* sw, const float* shift_rows, const float* shift_cols,
                                   float* gbuf, int accumulate, void* stream) {
    if (!shift_rows && !shift_cols)
        return crossclr_backward_w(plan, xhat_rows, xhat_cols, col_ranks, col_rank0, skip_rank, temperature, negative_weight, rz_rows,
                                   wrz_rows, rz_cols, wrz_cols, sw, gbuf, accumulate, stream);
    if (!plan || !xhat_rows || !xhat_cols || !rz_rows || !wrz_rows || !rz_cols || !wrz_cols || !gbuf || !shift_rows || !shift_cols)
        return fail(CROSSCLR_E_ARG, "NULL argument");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = make_geo(plan, col_ranks, col_rank0, skip_rank, temperature, negative_weight, &g, true);
    if (rc) return rc;
    if (plan->mode == CROSSCLR_MODE_FP32)
        return backward_generic<float>(plan, g, xhat_rows, xhat_cols, rz_rows, wrz_rows, rz_cols, wrz_cols, gbuf, accumulate, krows, kcols, shift_rows, shift_cols, stream);
    return backward_generic<bf16_t>(plan, g, xhat_rows, xhat_cols, rz_rows, wrz_rows, rz_cols, wrz_cols, gbuf, accumulate, krows, kcols, shift_rows, shift_cols, stream);
}


// ---- rectangular blocks of the sharded step with saved exponentials ------------------------------------------------------
// `first_rank`, `nranks`: column ranks first_rank .. first_rank+nranks-1 (mod plan->world) of the WHOLE gathered operand.
static int rect_geo(const crossclr_plan* plan, int first_rank, int nranks, float temperature, float negative_weight, Geo* g,
                    bool allow_row_shift = false) {
    if (first_rank < 0 || first_rank >= plan->world || nranks < 1 || nranks >= plan->world)
        return fail(CROSSCLR_E_ARG, "bad first_rank/nranks %d/%d for world %d", first_rank, nranks, plan->world);
    for (int i = 0; i < nranks; ++i)
        if ((first_rank + i) % plan->world == plan->rank) return fail(CROSSCLR_E_ARG, "the rank range must not contain this rank");
    int rc = make_geo(plan, nranks, first_rank, -1, temperature, negative_weight, g, allow_row_shift);
    if (rc) return rc;
    g->col_wrap = plan->world;
    return CROSSCLR_OK;
}

extern "C" size_t crossclr_rect_stash_bytes(const crossclr_plan* plan, int nranks) {
#ifdef CROSSCLR_NO_FAST
    return 0;
#else
    if (!plan || !plan->stash_bytes || nranks < 1) return 0;
    if (!plan->fast_path && plan->mode == CROSSCLR_MODE_FP32) {   // exact-fp32 plans: fp32 fragments of the rectangular block, up to 16 GiB
        const size_t sb = (size_t)(2 * plan->bpad / 32) * (size_t)(2 * plan->bpad / 32) * (size_t)nranks * 4096;
        return sb <= ((size_t)16 << 30) && plan->operand_bytes * (size_t)plan->world < ((size_t)1 << 32) ? sb : 0;
    }
    if (!plan->fast_path && plan->mode == CROSSCLR_MODE_BF16 && plan->Dpad > 1024)   // wide bf16 plans: bf16 records of the generic forward
        return plan->operand_bytes * (size_t)plan->world < ((size_t)1 << 32) ? wide_stash_bytes_rect(plan->bpad, nranks) : 0;
    if (!plan->fast_path) return 0;
    return fast_stash_bytes_rect(plan->bpad, plan->Dpad, nranks);
#endif
}

extern "C" int crossclr_forward_rect_save(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_all, int first_rank,
                                          int nranks, int with_colsums, float temperature, float negative_weight,
                                          const crossclr_sample_weights* sw, float* part, int slot0, float* colsum_out,
                                          void* stash, void* stream) {
    if (!plan || !xhat_rows || !xhat_all || !part || !stash || (with_colsums && !colsum_out)) return fail(CROSSCLR_E_ARG, "NULL argument");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_forward_rect_save needs the register-resident path");
#else
    if (plan->stash_bytes && !plan->fast_path && (plan->mode == CROSSCLR_MODE_FP32 || plan->Dpad > 1024)) {
        // exact-fp32 plans: the generic forward over the rank range, leaving its fp32 fragments behind ([row group][fragments of the range]);
        // wide bf16 plans (Dpad > 1024): the same launch leaves bf16 records ([row group][tile of the range]: fast_bwd_dsl_kernel<..., MODE 1>)
        if (with_colsums) return fail(CROSSCLR_E_ARG, "plans on the generic forward have no pair scheme (with_colsums must be 0)");
        if (!crossclr_rect_stash_bytes(plan, nranks)) return fail(CROSSCLR_E_ARG, "rectangular stash too large for this plan");
        if (plan->fwd_slots <= 0 || slot0 < 0 || slot0 % plan->fwd_slots != 0 || slot0 / plan->fwd_slots >= kLaunchGroups)
            return fail(CROSSCLR_E_ARG, "slot0 must be L * plan->fwd_slots, L = 0..%d", kLaunchGroups - 1);
        const float *kr32, *kc32;
        if (int rk = unpack_k(sw, &kr32, &kc32)) return rk;
        Geo g32;
        int rc32 = rect_geo(plan, first_rank, nranks, temperature, negative_weight, &g32);
        if (rc32) return rc32;
        int* header32 = reinterpret_cast<int*>(part + ws_flag_off(plan)) + 4 * (slot0 / plan->fwd_slots);
        rc32 = device_zero_header(header32, stream);
        if (rc32) return rc32;
        return forward_generic(plan, g32, xhat_rows, xhat_all, part + (size_t)slot0 * 2 * plan->bpad, kc32, nullptr, 0, stream, static_cast<float*>(stash));
    }
    if (!plan->stash_bytes || !plan->fast_path) return fail(CROSSCLR_E_ARG, "this plan has no save-for-backward path for remote blocks");
    if (with_colsums && nranks > (plan->world - 1) / 2) return fail(CROSSCLR_E_ARG, "a pair range holds at most (world-1)/2 ranks");
    if (plan->fwd_slots <= 0 || slot0 < 0 || slot0 % plan->fwd_slots != 0 || slot0 / plan->fwd_slots >= kLaunchGroups)
        return fail(CROSSCLR_E_ARG, "slot0 must be L * plan->fwd_slots, L = 0..%d", kLaunchGroups - 1);
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = rect_geo(plan, first_rank, nranks, temperature, negative_weight, &g);
    if (rc) return rc;
    float* out = part + (size_t)slot0 * 2 * plan->bpad;
    int* header = reinterpret_cast<int*>(part + ws_flag_off(plan)) + 4 * (slot0 / plan->fwd_slots);
    float* colpart = part + ws_paircol_off(plan);
    const FwdWork wk = fast_forward_work(plan, nranks, -1, false, with_colsums != 0);
    rc = fast_forward_pipe(plan, g, wk, xhat_rows, xhat_all, out, colpart, header, with_colsums ? 3 : 2, krows, kcols, stash, stream);
    if (rc) return fail(rc, "fast_forward_pipe: unsupported Dpad %d", plan->Dpad);
    if (with_colsums) {
        const int nrb = 2 * plan->bpad / (32 * fast_fwd_tpr(plan->Dpad));
        const int ncols = nranks * 2 * plan->bpad;
        LAUNCH(colsum_reduce_kernel, dim3((ncols + 255) / 256), dim3(256), stream, (const float*)colpart, nrb, ncols, colsum_out);
    }
    return launch_status("fast_fwd_pipe_kernel (rect, save)");
#endif
}

extern "C" int crossclr_backward_rect_saved(const crossclr_plan* plan, const void* xhat_all, const void* stash, int first_rank,
                                            int nranks, float temperature, float negative_weight, const float* rz_rows,
                                            const float* wrz_rows, const float* rz_all, const float* wrz_all,
                                            const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream) {
    if (!plan || !xhat_all || !stash || !rz_rows || !wrz_rows || !rz_all || !wrz_all || !gbuf) return fail(CROSSCLR_E_ARG, "NULL argument");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_backward_rect_saved needs the register-resident path");
#else
    if (plan->stash_bytes && !plan->fast_path && plan->mode == CROSSCLR_MODE_FP32) {   // exact-fp32 plans: bwd_saved32_kernel<..., RECT>
        const float *kr32, *kc32;
        if (int rk = unpack_k(sw, &kr32, &kc32)) return rk;
        Geo g32;
        int rc32 = rect_geo(plan, first_rank, nranks, temperature, negative_weight, &g32);
        if (rc32) return rc32;
        const int NQ = nranks * (2 * plan->bpad / 32);
        const int tps32 = (NQ + plan->bwd_slices - 1) / plan->bwd_slices;
        const unsigned rb = 2 * plan->bpad / 64, nz = (unsigned)plan->bwd_slices;
        dim3 block(256);
#define CROSSCLR_LS32X(DC)                                                                                                                    \
    do {                                                                                                                                      \
        if (kr32) LAUNCH((bwd_saved32_kernel<DC, true, false, true>), dim3(rb, plan->Dpad / DC, nz), block, stream, (const float*)xhat_all,   \
                         (const float*)stash, g32, rz_rows, wrz_rows, gbuf, accumulate, tps32, kr32, rz_all, wrz_all, kc32);                  \
        else LAUNCH((bwd_saved32_kernel<DC, false, false, true>), dim3(rb, plan->Dpad / DC, nz), block, stream, (const float*)xhat_all,       \
                    (const float*)stash, g32, rz_rows, wrz_rows, gbuf, accumulate, tps32, kr32, rz_all, wrz_all, kc32);                       \
    } while (0)
        if (plan->Dpad % 256 == 0) CROSSCLR_LS32X(256);
        else if (plan->Dpad % 128 == 0) CROSSCLR_LS32X(128);
        else CROSSCLR_LS32X(64);
#undef CROSSCLR_LS32X
        return launch_status("bwd_saved32_kernel (rect)");
    }
    if (!plan->stash_bytes || !(plan->fast_path || (plan->mode == CROSSCLR_MODE_BF16 && plan->Dpad > 1024)))
        return fail(CROSSCLR_E_ARG, "this plan has no save-for-backward path for remote blocks");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = rect_geo(plan, first_rank, nranks, temperature, negative_weight, &g);
    if (rc) return rc;
    rc = fast_backward_saved(plan, g, xhat_all, stash, rz_rows, wrz_rows, rz_all, wrz_all, gbuf, accumulate, krows, kcols, 1, stream);
    return rc ? fail(rc, "fast_backward_saved: unsupported Dpad %d", plan->Dpad) : launch_status("fast_bwd_dsl_kernel (rect)");
#endif
}

// ---- the two-pass regime's rectangular blocks (exact-fp32 plans): U and Ut of this rank's rows against other ranks' columns ----------
// bf16 register-resident plans: two arrays of bf16 records (U, Ut) in the rectangular layout + (world + 1) x 2 bpad floats of zeros (the statistics
// of the side a launch must not weigh: W = U rz_p + Ut rz_q is formed as two rectangular launches of the saved backward)
static size_t rect_zero_floats(const crossclr_plan* plan) { return (size_t)(plan->world + 1) * 2 * plan->bpad; }
extern "C" size_t crossclr_rect_stash_bytes_s(const crossclr_plan* plan, int nranks) {
    if (!plan) return 0;
    const size_t one = crossclr_rect_stash_bytes(plan, nranks);
    if (plan->mode == CROSSCLR_MODE_BF16)
        return (one && rect_bytes_s(plan)) ? 2 * one + rect_zero_floats(plan) * 4 : 0;
    if (plan->fast_path || plan->mode != CROSSCLR_MODE_FP32) return 0;
    return one && 2 * one <= ((size_t)32 << 30) ? 2 * one : 0;
}

extern "C" int crossclr_forward_rect_save_s(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_all, int first_rank,
                                            int nranks, float temperature, float negative_weight, const crossclr_sample_weights* sw,
                                            const float* shift_rows, const float* shift_all, float* part, int slot0, void* stash,
                                            void* stream) {
    if (!plan || !xhat_rows || !xhat_all || !shift_rows || !shift_all || !part || !stash) return fail(CROSSCLR_E_ARG, "NULL argument");
    if (!crossclr_rect_stash_bytes_s(plan, nranks))
        return fail(CROSSCLR_E_ARG, "this plan has no two-pass save-for-backward path for remote blocks (crossclr_rect_stash_bytes_s == 0)");
    if (plan->fwd_slots <= 0 || slot0 < 0 || slot0 % plan->fwd_slots != 0 || slot0 / plan->fwd_slots >= kLaunchGroups)
        return fail(CROSSCLR_E_ARG, "slot0 must be L * plan->fwd_slots, L = 0..%d", kLaunchGroups - 1);
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = rect_geo(plan, first_rank, nranks, temperature, negative_weight, &g, true);
    if (rc) return rc;
    int* header = reinterpret_cast<int*>(part + ws_flag_off(plan)) + 4 * (slot0 / plan->fwd_slots);
    rc = device_zero_header(header, stream);
    if (rc) return rc;
    if (plan->mode == CROSSCLR_MODE_BF16) {
        rc = device_zero(static_cast<unsigned char*>(stash) + 2 * crossclr_rect_stash_bytes(plan, nranks), rect_zero_floats(plan) * 4, stream);
        if (rc) return rc;
    }
    return forward_generic(plan, g, xhat_rows, xhat_all, part + (size_t)slot0 * 2 * plan->bpad, kcols, shift_rows, 2, stream,
                           static_cast<float*>(stash), shift_all);
}

extern "C" int crossclr_backward_rect_saved_s(const crossclr_plan* plan, const void* xhat_all, const void* stash, int first_rank,
                                              int nranks, float temperature, float negative_weight, const float* rz_rows,
                                              const float* wrz_rows, const float* rz_all, const float* wrz_all,
                                              const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream) {
    if (!plan || !xhat_all || !stash || !rz_rows || !wrz_rows || !rz_all || !wrz_all || !gbuf) return fail(CROSSCLR_E_ARG, "NULL argument");
    if (!crossclr_rect_stash_bytes_s(plan, nranks))
        return fail(CROSSCLR_E_ARG, "this plan has no two-pass save-for-backward path for remote blocks (crossclr_rect_stash_bytes_s == 0)");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = rect_geo(plan, first_rank, nranks, temperature, negative_weight, &g, true);
    if (rc) return rc;
#ifndef CROSSCLR_NO_FAST
    if (plan->mode == CROSSCLR_MODE_BF16) {
        // W[p][q] = U[p][q] rz_p + Ut[p][q] rz_q: the launch over U weighs with the ROW statistics only (column side: zeros), the launch over Ut
        // with the COLUMN statistics only (row side: zeros) and accumulates -- two rectangular launches of the saved backward, no recompute
        const size_t one = crossclr_rect_stash_bytes(plan, nranks);
        const unsigned char* U = static_cast<const unsigned char*>(stash);
        const float* zeros = reinterpret_cast<const float*>(U + 2 * one);       // [world][2 bpad] for the column side, [2 bpad] of it for the rows
        rc = fast_backward_saved(plan, g, xhat_all, U, rz_rows, wrz_rows, zeros, zeros, gbuf, accumulate, krows, kcols, 1, stream);
        if (rc) return fail(rc, "fast_backward_saved (two-pass, rows' side): unsupported Dpad %d", plan->Dpad);
        rc = fast_backward_saved(plan, g, xhat_all, U + one, zeros, zeros, rz_all, wrz_all, gbuf, 1, krows, kcols, 1, stream);
        return rc ? fail(rc, "fast_backward_saved (two-pass, columns' side): unsupported Dpad %d", plan->Dpad)
                  : launch_status("fast_bwd_dsl_kernel (rect, two-pass pair)");
    }
#endif
    const int NQ = nranks * (2 * plan->bpad / 32);
    const int tps = (NQ + plan->bwd_slices - 1) / plan->bwd_slices;
    const unsigned rb = 2 * plan->bpad / 64, nz = (unsigned)plan->bwd_slices;
    dim3 block(256);
#define CROSSCLR_LS32XS(DC)                                                                                                                  \
    do {                                                                                                                                     \
        if (krows) LAUNCH((bwd_saved32_kernel<DC, true, true, true>), dim3(rb, plan->Dpad / DC, nz), block, stream, (const float*)xhat_all,  \
                          (const float*)stash, g, rz_rows, wrz_rows, gbuf, accumulate, tps, krows, rz_all, wrz_all, kcols);                  \
        else LAUNCH((bwd_saved32_kernel<DC, false, true, true>), dim3(rb, plan->Dpad / DC, nz), block, stream, (const float*)xhat_all,       \
                    (const float*)stash, g, rz_rows, wrz_rows, gbuf, accumulate, tps, krows, rz_all, wrz_all, kcols);                        \
    } while (0)
    if (plan->Dpad % 256 == 0) CROSSCLR_LS32XS(256);
    else if (plan->Dpad % 128 == 0) CROSSCLR_LS32XS(128);
    else CROSSCLR_LS32XS(64);
#undef CROSSCLR_LS32XS
    return launch_status("bwd_saved32_kernel (rect, two-pass)");
}


// The fragment-major copy of `nranks` consecutive packed operands (the slices of a gathered operand a rank received): what the pair
// kernel's rectangular launches read their column tiles from.
extern "C" int crossclr_pack_xf_from_packed(const crossclr_plan* plan, const void* xhat_packed, int nranks, void* xhat_xf, void* stream) {
    if (!plan || !xhat_packed || !xhat_xf || nranks < 1) return fail(CROSSCLR_E_ARG, "bad arguments");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_pack_xf_from_packed needs the register-resident path");
#else
    if (!plan->xf_bytes) return fail(CROSSCLR_E_ARG, "this plan has no fragment-major operand (xf_bytes == 0)");
    const size_t tiles = (size_t)nranks * 2 * plan->bpad / 32;
    if (tiles > 0x7fffffffull) return fail(CROSSCLR_E_ARG, "too many rows");
    LAUNCH(xf_from_packed_kernel, dim3((unsigned)tiles, (unsigned)((plan->Dpad + 1023) / 1024)), dim3(256), stream, (const bf16_t*)xhat_packed,
           (unsigned char*)xhat_xf, plan->Dpad);
    return launch_status("xf_from_packed_kernel");
#endif
}

// crossclr_backward_rect_saved on the fragment-major copy of the gathered operand, with the pair kernel (two tiles per barrier interval).
extern "C" int crossclr_backward_rect_saved_xfp(const crossclr_plan* plan, const void* xf_all, const void* stash, int first_rank,
                                                int nranks, float temperature, float negative_weight, const float* rz_rows,
                                                const float* wrz_rows, const float* rz_all, const float* wrz_all,
                                                const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream) {
    if (!plan || !xf_all || !stash || !rz_rows || !wrz_rows || !rz_all || !wrz_all || !gbuf) return fail(CROSSCLR_E_ARG, "NULL argument");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_backward_rect_saved_xfp needs the register-resident path");
#else
    if (!plan->stash_bytes || !plan->fast_path || !plan->xf_bytes) return fail(CROSSCLR_E_ARG, "this plan has no fragment-major saved backward for remote blocks");
    if ((size_t)plan->world * plan->operand_bytes >= ((size_t)1 << 32) || fast_stash_bytes_rect(plan->bpad, plan->Dpad, nranks) >= ((size_t)1 << 32))
        return fail(CROSSCLR_E_ARG, "crossclr_backward_rect_saved_xfp uses 32-bit offsets (operand / stash of 4 GiB or more): use crossclr_backward_rect_saved");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = rect_geo(plan, first_rank, nranks, temperature, negative_weight, &g);
    if (rc) return rc;
    rc = fast_backward_saved(plan, g, xf_all, stash, rz_rows, wrz_rows, rz_all, wrz_all, gbuf, accumulate, krows, kcols, 5, stream);
    return rc ? fail(rc, "fast_backward_saved (xfp, rect): unsupported Dpad %d", plan->Dpad) : launch_status("fast_bwd_xfp_kernel (rect)");
#endif
}

// crossclr_backward_rect_saved_t on this rank's LOCAL fragment-major operand (what crossclr_normalize_xf wrote), with the pair kernel.
extern "C" int crossclr_backward_rect_saved_t_xfp(const crossclr_plan* plan, const void* xf_rows, const void* stash, int first_rank,
                                                  int nranks, int which, float temperature, float negative_weight, const float* rz_rows,
                                                  const float* wrz_rows, const float* rz_all, const float* wrz_all,
                                                  const crossclr_sample_weights* sw, float* gpartner, void* stream) {
    if (!plan || !xf_rows || !stash || !rz_rows || !wrz_rows || !rz_all || !wrz_all || !gpartner) return fail(CROSSCLR_E_ARG, "NULL argument");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_backward_rect_saved_t_xfp needs the register-resident path");
#else
    if (!plan->stash_bytes || !plan->fast_path || !plan->xf_bytes) return fail(CROSSCLR_E_ARG, "this plan has no fragment-major saved backward for remote blocks");
    if (which < 0 || which >= nranks) return fail(CROSSCLR_E_ARG, "which must be 0 .. nranks-1");
    if (fast_stash_bytes_rect(plan->bpad, plan->Dpad, nranks) >= ((size_t)1 << 32))
        return fail(CROSSCLR_E_ARG, "crossclr_backward_rect_saved_t_xfp uses 32-bit offsets (stash of 4 GiB or more): use crossclr_backward_rect_saved_t");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = rect_geo(plan, first_rank, nranks, temperature, negative_weight, &g);     // (validates the range; scales)
    if (rc) return rc;
    const int partner = (first_rank + which) % plan->world;
    const size_t n2 = (size_t)2 * plan->bpad;
    g.col_ranks = nranks;        // rank segments per row of the rectangular stash
    g.skip_rank = which;         // the partner's segment inside a stash row
    g.col_rank0 = plan->rank; g.col_wrap = 0;
    g.row_rank = partner;
    rc = fast_backward_saved(plan, g, xf_rows, stash, rz_all + partner * n2, wrz_all + partner * n2, rz_rows, wrz_rows, gpartner, 0,
                             kcols ? kcols + partner * n2 : nullptr, krows, 6, stream);
    return rc ? fail(rc, "fast_backward_saved (xfp, transposed): unsupported Dpad %d", plan->Dpad) : launch_status("fast_bwd_xfp_kernel (rect, transposed)");
#endif
}

// The transpose of one saved rectangular block: what block (this rank x partner) contributes to the PARTNER's gradient buffer.
extern "C" int crossclr_backward_rect_saved_t(const crossclr_plan* plan, const void* xhat_rows, const void* stash, int first_rank,
                                              int nranks, int which, float temperature, float negative_weight, const float* rz_rows,
                                              const float* wrz_rows, const float* rz_all, const float* wrz_all,
                                              const crossclr_sample_weights* sw, float* gpartner, void* stream) {
    if (!plan || !xhat_rows || !stash || !rz_rows || !wrz_rows || !rz_all || !wrz_all || !gpartner) return fail(CROSSCLR_E_ARG, "NULL argument");
#ifdef CROSSCLR_NO_FAST
    return fail(CROSSCLR_E_ARG, "crossclr_backward_rect_saved_t needs the register-resident path");
#else
    if (!plan->stash_bytes || !plan->fast_path) return fail(CROSSCLR_E_ARG, "this plan has no save-for-backward path for remote blocks");
    if (which < 0 || which >= nranks) return fail(CROSSCLR_E_ARG, "which must be 0 .. nranks-1");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = rect_geo(plan, first_rank, nranks, temperature, negative_weight, &g);     // (validates the range; scales)
    if (rc) return rc;
    const int partner = (first_rank + which) % plan->world;
    const size_t n2 = (size_t)2 * plan->bpad;
    // the kernel sees: rows = the partner's 2 bpad rows (statistics: its segment of the gathered arrays), columns = this rank's own rows
    g.col_ranks = nranks;        // rank segments per row of the rectangular stash
    g.skip_rank = which;         // the partner's segment inside a stash row
    g.col_rank0 = plan->rank; g.col_wrap = 0;
    g.row_rank = partner;
    rc = fast_backward_saved(plan, g, xhat_rows, stash, rz_all + partner * n2, wrz_all + partner * n2, rz_rows, wrz_rows, gpartner, 0,
                             kcols ? kcols + partner * n2 : nullptr, krows, 2, stream);
    return rc ? fail(rc, "fast_backward_saved: unsupported Dpad %d", plan->Dpad) : launch_status("fast_bwd_dsl_kernel (rect, transposed)");
#endif
}

// the recomputing backward over a (wrapping) rank range of the whole gathered operand: the blocks OTHER ranks evaluated in the
// pair scheme, whose exponentials this rank therefore does not hold
extern "C" int crossclr_backward_ranks(const crossclr_plan* plan, const void* xhat_rows, const void* xhat_all, int first_rank,
                                       int nranks, float temperature, float negative_weight, const float* rz_rows,
                                       const float* wrz_rows, const float* rz_all, const float* wrz_all,
                                       const crossclr_sample_weights* sw, float* gbuf, int accumulate, void* stream) {
    if (!plan || !xhat_rows || !xhat_all || !rz_rows || !wrz_rows || !rz_all || !wrz_all || !gbuf) return fail(CROSSCLR_E_ARG, "NULL argument");
    const float *krows, *kcols;
    if (int rk = unpack_k(sw, &krows, &kcols)) return rk;
    Geo g;
    int rc = rect_geo(plan, first_rank, nranks, temperature, negative_weight, &g);
    if (rc) return rc;
#ifndef CROSSCLR_NO_FAST
    if (plan->fast_bwd) {
        rc = plan->fast_bwd == 2
                 ? fast_backward16(plan, g, xhat_rows, xhat_all, rz_rows, wrz_rows, rz_all, wrz_all, gbuf, accumulate, krows, kcols, stream)
                 : fast_backward(plan, g, xhat_rows, xhat_all, rz_rows, wrz_rows, rz_all, wrz_all, gbuf, accumulate, krows, kcols, stream);
        return rc ? fail(rc, "fast backward: unsupported Dpad %d", plan->Dpad) : launch_status("fast_bwd_kernel (ranks)");
    }
#endif
    if (plan->mode == CROSSCLR_MODE_FP32)
        return backward_generic<float>(plan, g, xhat_rows, xhat_all, rz_rows, wrz_rows, rz_all, wrz_all, gbuf, accumulate, krows, kcols, nullptr, nullptr, stream);
    return backward_generic<bf16_t>(plan, g, xhat_rows, xhat_all, rz_rows, wrz_rows, rz_all, wrz_all, gbuf, accumulate, krows, kcols, nullptr, nullptr, stream);
}

template <typename TIN>
static int backward_finish_t(const crossclr_plan* p, const float* gbuf, const void* v, const void* t, long ldv, long ldt,
                             const float* inv_norm, float temperature, const double* grad_out, void* gv, void* gt,
                             long ldgv, long ldgt, const float* lw, void* stream, int prenormalized = 0) {
    Geo g; memset(&g, 0, sizeof(g));
    g.b = p->b; g.bpad = p->bpad; g.D = p->D; g.Dpad = p->Dpad;
    dim3 grid((2 * p->b + 3) / 4), block(256);
    if (p->D <= 256 * kRowCache) {      // row pairs: each raw row read once
#define CROSSCLR_FINISH_PAIR(KC)                                                                                                        \
    LAUNCH((bwd_finish_pair_kernel<TIN, KC>), dim3((p->b + 3) / 4), block, stream, gbuf, p->bwd_slices, (const TIN*)v, (const TIN*)t, ldv, ldt, \
           g, inv_norm, 1.0f / temperature, p->b * p->world, grad_out, (TIN*)gv, (TIN*)gt, ldgv, ldgt, lw, prenormalized)
        switch ((p->D + 255) / 256) {      // (one instantiation per number of 256-element stretches a lane caches)
            case 1: CROSSCLR_FINISH_PAIR(1); break;
            case 2: CROSSCLR_FINISH_PAIR(2); break;
            case 3: CROSSCLR_FINISH_PAIR(3); break;
            default: CROSSCLR_FINISH_PAIR(4); break;
        }
#undef CROSSCLR_FINISH_PAIR
        return launch_status("bwd_finish_pair_kernel");
    }
    LAUNCH((bwd_finish_kernel<TIN>), grid, block, stream, gbuf, p->bwd_slices, (const TIN*)v, (const TIN*)t, ldv, ldt, g, inv_norm,
           1.0f / temperature, p->b * p->world, grad_out, (TIN*)gv, (TIN*)gt, ldgv, ldgt, lw, prenormalized);
    return launch_status("bwd_finish_kernel");
}

extern "C" int crossclr_backward_finish(const crossclr_plan* plan, const float* gbuf, const void* video,
                                        const void* text, long ld_video, long ld_text, int in_dtype,
                                        const float* inv_norm, float temperature, const double* grad_out,
                                        void* grad_video, void* grad_text, long ld_gvideo, long ld_gtext,
                                        void* stream) {
    return crossclr_backward_finish_w(plan, gbuf, video, text, ld_video, ld_text, in_dtype, inv_norm, temperature, nullptr,
                                      grad_out, grad_video, grad_text, ld_gvideo, ld_gtext, stream);
}

extern "C" int crossclr_backward_finish_w(const crossclr_plan* plan, const float* gbuf, const void* video,
                                          const void* text, long ld_video, long ld_text, int in_dtype,
                                          const float* inv_norm, float temperature, const crossclr_sample_weights* sw,
                                          const double* grad_out, void* grad_video, void* grad_text, long ld_gvideo,
                                          long ld_gtext, void* stream) {
    return crossclr_backward_finish_p(plan, gbuf, video, text, ld_video, ld_text, in_dtype, inv_norm, temperature, sw, grad_out,
                                      grad_video, grad_text, ld_gvideo, ld_gtext, 0, stream);
}

extern "C" int crossclr_backward_finish_p(const crossclr_plan* plan, const float* gbuf, const void* video,
                                          const void* text, long ld_video, long ld_text, int in_dtype,
                                          const float* inv_norm, float temperature, const crossclr_sample_weights* sw,
                                          const double* grad_out, void* grad_video, void* grad_text, long ld_gvideo,
                                          long ld_gtext, int prenormalized, void* stream) {
    const float* lw = sw ? sw->loss_weight : nullptr;
    if (!plan || !gbuf || !video || !text || !inv_norm || !grad_out || !grad_video || !grad_text)
        return fail(CROSSCLR_E_ARG, "NULL argument");
    if (!(temperature > 0.f)) return fail(CROSSCLR_E_ARG, "temperature must be > 0");
    switch (in_dtype) {
        case CROSSCLR_IN_F32: return backward_finish_t<float>(plan, gbuf, video, text, ld_video, ld_text, inv_norm, temperature, grad_out, grad_video, grad_text, ld_gvideo, ld_gtext, lw, stream, prenormalized);
        case CROSSCLR_IN_F64: return backward_finish_t<double>(plan, gbuf, video, text, ld_video, ld_text, inv_norm, temperature, grad_out, grad_video, grad_text, ld_gvideo, ld_gtext, lw, stream, prenormalized);
        case CROSSCLR_IN_F16: return backward_finish_t<in_f16>(plan, gbuf, video, text, ld_video, ld_text, inv_norm, temperature, grad_out, grad_video, grad_text, ld_gvideo, ld_gtext, lw, stream, prenormalized);
        case CROSSCLR_IN_BF16: return backward_finish_t<in_bf16>(plan, gbuf, video, text, ld_video, ld_text, inv_norm, temperature, grad_out, grad_video, grad_text, ld_gvideo, ld_gtext, lw, stream, prenormalized);
    }
    return fail(CROSSCLR_E_ARG, "bad in_dtype %d", in_dtype);
}

// ------------------------------------------------------------------------------------------------
// score statistics of the inter-modal block: max-margin ranking loss (trainer/loss.py:17-41) and retrieval ranks
static int score_geo(const crossclr_plan* p, float margin, Geo* g) {
    if (!p) return fail(CROSSCLR_E_ARG, "plan is NULL");
    if (p->world != 1) return fail(CROSSCLR_E_ARG, "score statistics are single-device (plan->world must be 1)");
    if (!isfinite(margin)) return fail(CROSSCLR_E_ARG, "margin must be finite");
    memset(g, 0, sizeof(*g));
    g->b = p->b; g->bpad = p->bpad; g->D = p->D; g->Dpad = p->Dpad;
    g->col_ranks = 1; g->col_rank0 = 0; g->row_rank = 0; g->skip_rank = -1; g->col_wrap = 0;
    g->c_inter = 1.f; g->c_intra = 1.f; g->m2 = margin;
    return CROSSCLR_OK;
}
template <typename T>
static void score_launch(const crossclr_plan* plan, const Geo& g, const void* x, float* out, float* cnt, const float* diag, int mode,
                         void* stream, float* colpart = nullptr, unsigned char* hinge_mask = nullptr) {
    if (mode == 3 && colpart) {   // one pass: rows of modality 0 against the column tiles of modality 1, column statistics for the rest
        const int half_tiles = plan->bpad / 128, nsplit = plan->fwd_slots;
        const int tps = (half_tiles + nsplit - 1) / nsplit;
        // (the kernel's `header` argument is free in this mode: it carries the optional hinge mask of crossclr_score_rows_save)
        LAUNCH((fwd_sums_kernel<T, false, 3, false, true>), dim3(plan->bpad / 128, nsplit), dim3(256), stream, (const T*)x, (const T*)x, g, tps, out,
               (const float*)nullptr, diag, cnt, reinterpret_cast<int*>(hinge_mask), colpart);
        return;
    }
    const int ntiles = 2 * plan->bpad / 128;
    const int nsplit = mode == 4 ? 1 : plan->fwd_slots;
    const int tps = (ntiles + nsplit - 1) / nsplit;
    dim3 grid(2 * plan->bpad / 128, nsplit), block(256);
    if (mode == 4) LAUNCH((fwd_sums_kernel<T, false, 4>), grid, block, stream, (const T*)x, (const T*)x, g, tps, out, (const float*)nullptr, diag, cnt, (int*)nullptr, (float*)nullptr);
    else LAUNCH((fwd_sums_kernel<T, false, 3>), grid, block, stream, (const T*)x, (const T*)x, g, tps, out, (const float*)nullptr, diag, cnt, (int*)nullptr, (float*)nullptr);
}

extern "C" int crossclr_score_diag(const crossclr_plan* plan, const void* xhat, float* diag, void* stream) {
    if (!plan || !xhat || !diag) return fail(CROSSCLR_E_ARG, "NULL argument");
    Geo g;
    if (int rc = score_geo(plan, 0.f, &g)) return rc;
    if (plan->mode == CROSSCLR_MODE_FP32) score_launch<float>(plan, g, xhat, diag, nullptr, nullptr, 4, stream);
    else score_launch<bf16_t>(plan, g, xhat, diag, nullptr, nullptr, 4, stream);
    return launch_status("fwd_sums_kernel (positive-pair scores)");
}

static int score_rows_impl(const crossclr_plan* plan, const void* xhat, const float* diag, float margin, float* part,
                           float* hinge, float* active, double* loss_sum, unsigned char* hinge_mask, void* stream);
extern "C" int crossclr_score_rows(const crossclr_plan* plan, const void* xhat, const float* diag, float margin, float* part,
                                   float* hinge, float* active, double* loss_sum, void* stream) {
    return score_rows_impl(plan, xhat, diag, margin, part, hinge, active, loss_sum, nullptr, stream);
}
extern "C" size_t crossclr_maxmargin_mask_bytes(const crossclr_plan* plan) {
    if (!plan || plan->world != 1) return 0;
    return (size_t)plan->bpad * (size_t)plan->bpad;
}
extern "C" int crossclr_score_rows_save(const crossclr_plan* plan, const void* xhat, const float* diag, float margin, float* part,
                                        float* hinge, float* active, double* loss_sum, void* hinge_mask, void* stream) {
    if (!hinge_mask) return fail(CROSSCLR_E_ARG, "NULL argument");
    if (env_knobs().disable_symmetric) return fail(CROSSCLR_E_ARG, "CROSSCLR_DISABLE_SYMMETRIC: the hinge mask is written by the one-pass evaluation only");
    return score_rows_impl(plan, xhat, diag, margin, part, hinge, active, loss_sum, static_cast<unsigned char*>(hinge_mask), stream);
}
static int score_rows_impl(const crossclr_plan* plan, const void* xhat, const float* diag, float margin, float* part,
                           float* hinge, float* active, double* loss_sum, unsigned char* hinge_mask, void* stream) {
    if (!plan || !xhat || !diag || !part || !hinge || !active || !loss_sum) return fail(CROSSCLR_E_ARG, "NULL argument");
    Geo g;
    if (int rc = score_geo(plan, margin, &g)) return rc;
    if (plan->fwd_slots <= 0) return fail(CROSSCLR_E_ARG, "bad plan");
    float* cnt = part + (size_t)plan->fwd_slots * 2 * plan->bpad;     // launch group 1 of the forward workspace
    float* colpart = env_knobs().disable_symmetric ? nullptr : part + ws_colpart_off(plan);   // one pass for both directions
    if (plan->mode == CROSSCLR_MODE_FP32) score_launch<float>(plan, g, xhat, part, cnt, diag, 3, stream, colpart, hinge_mask);
    else score_launch<bf16_t>(plan, g, xhat, part, cnt, diag, 3, stream, colpart, hinge_mask);
    if (int rc = launch_status("fwd_sums_kernel (score rows)")) return rc;
    const int nb = plan->loss_ws_doubles - 1;
    LAUNCH(score_finish_kernel, dim3(nb), dim3(256), stream, (const float*)part, (const float*)cnt, plan->fwd_slots, plan->bpad, plan->b,
           hinge, active, loss_sum, (const float*)colpart);
    // loss_sum[0] = sum of the hinges, loss_sum[1] = the reference's mean: / (B * B)   (trainer/loss.py:41)
    LAUNCH(fwd_finish_reduce_kernel, dim3(1), dim3(64), stream, loss_sum, nb, 1.0 / ((double)plan->b * (double)plan->b));
    return launch_status("score_finish_kernel");
}

template <typename T>
static int maxmargin_backward_t(const crossclr_plan* p, const Geo& g, const void* x, const float* diag, float* gbuf, void* stream) {
    dim3 block(256);
    const unsigned rb = 2 * p->bpad / 64, nz = (unsigned)p->bwd_slices;
    const int ntiles = 2 * p->bpad / 64;
    const int tps = (ntiles + p->bwd_slices - 1) / p->bwd_slices;
#define CROSSCLR_LMM(DC) LAUNCH((bwd_kernel<T, DC, false, false, 1>), dim3(rb, p->Dpad / DC, nz), block, stream, (const T*)x, (const T*)x, g, \
                                diag, diag, diag, diag, gbuf, 0, tps, (const float*)nullptr, (const float*)nullptr, diag, diag)
    if (p->Dpad % 256 == 0) CROSSCLR_LMM(256);
    else if (p->Dpad % 128 == 0) CROSSCLR_LMM(128);
    else CROSSCLR_LMM(64);
#undef CROSSCLR_LMM
    return launch_status("bwd_kernel (max-margin)");
}

extern "C" int crossclr_maxmargin_backward(const crossclr_plan* plan, const void* xhat, const float* diag, float margin, float* gbuf,
                                           void* stream) {
    if (!plan || !xhat || !diag || !gbuf) return fail(CROSSCLR_E_ARG, "NULL argument");
    Geo g;
    if (int rc = score_geo(plan, margin, &g)) return rc;
    if (plan->mode == CROSSCLR_MODE_FP32) return maxmargin_backward_t<float>(plan, g, xhat, diag, gbuf, stream);
    return maxmargin_backward_t<bf16_t>(plan, g, xhat, diag, gbuf, stream);
}

template <typename T>
static int maxmargin_backward_saved_t(const crossclr_plan* p, const Geo& g, const void* x, const unsigned char* mask, float* gbuf, void* stream) {
    dim3 block(256);
    const unsigned rb = 2 * p->bpad / 64, nz = (unsigned)p->bwd_slices;
    const int ntiles = 2 * p->bpad / 64;
    const int tps = (ntiles + p->bwd_slices - 1) / p->bwd_slices;
#define CROSSCLR_LMS(DC) LAUNCH((maxmargin_saved_kernel<T, DC>), dim3(rb, p->Dpad / DC, nz), block, stream, (const T*)x, g, mask, gbuf, tps)
    if (p->Dpad % 256 == 0) CROSSCLR_LMS(256);
    else if (p->Dpad % 128 == 0) CROSSCLR_LMS(128);
    else CROSSCLR_LMS(64);
#undef CROSSCLR_LMS
    return launch_status("maxmargin_saved_kernel");
}
extern "C" int crossclr_maxmargin_backward_saved(const crossclr_plan* plan, const void* xhat, const void* hinge_mask, float* gbuf, void* stream) {
    if (!plan || !xhat || !hinge_mask || !gbuf) return fail(CROSSCLR_E_ARG, "NULL argument");
    Geo g;
    if (int rc = score_geo(plan, 0.f, &g)) return rc;
    if (plan->mode == CROSSCLR_MODE_FP32) return maxmargin_backward_saved_t<float>(plan, g, xhat, static_cast<const unsigned char*>(hinge_mask), gbuf, stream);
    return maxmargin_backward_saved_t<bf16_t>(plan, g, xhat, static_cast<const unsigned char*>(hinge_mask), gbuf, stream);
}

extern "C" int crossclr_maxmargin_backward_finish(const crossclr_plan* plan, const float* gbuf, const void* im, const void* s,
                                                  long ld_im, long ld_s, int in_dtype, const float* ones, const float* active,
                                                  const double* grad_out, void* grad_im, void* grad_s, long ld_gim, long ld_gs,
                                                  void* stream) {
    if (!plan || !ones || !active) return fail(CROSSCLR_E_ARG, "NULL argument");
    // bwd_finish_kernel with unit rows as given: out = (sum_slices * inv_tau / (2B) - partner * inv_tau / B * (lw_im + lw_s) / 2) * grad_out;
    // inv_tau = 2 / B and lw = the active-hinge counts make that (sum_slices - (a_im + a_s) partner) / B^2   (loss.py:33-41)
    crossclr_sample_weights sw = {nullptr, nullptr, active};
    return crossclr_backward_finish_p(plan, gbuf, im, s, ld_im, ld_s, in_dtype, ones, 0.5f * (float)plan->b, &sw, grad_out, grad_im, grad_s,
                                      ld_gim, ld_gs, 1, stream);
}

// ------------------------------------------------------------------------------------------------
// influential-sample statistics
template <typename TIN>
static int infl_colsum_t(const void* xv, const void* xt, long ldv, long ldt, int b, int Din, float* inv_norm, float* partial,
                         double* colsum, void* stream) {
#define CROSSCLR_LI(KC) LAUNCH((infl_colsum_kernel<TIN, KC>), dim3(kInflBlocks, 2), dim3(256), stream, (const TIN*)xv, (const TIN*)xt, ldv, ldt, b, Din, inv_norm, partial)
    if (Din <= 256) CROSSCLR_LI(1);
    else if (Din <= 512) CROSSCLR_LI(2);
    else if (Din <= 1024) CROSSCLR_LI(4);
    else if (Din <= 2048) CROSSCLR_LI(8);
    else CROSSCLR_LI(16);
#undef CROSSCLR_LI
    LAUNCH(infl_colsum_finish_kernel, dim3((Din + 63) / 64, 2), dim3(256), stream, (const float*)partial, kInflBlocks, Din, colsum);
    return launch_status("infl_colsum_kernel");
}
template <typename TIN>
static int infl_conn_t(const void* xv, const void* xt, long ldv, long ldt, int b, int Din, const float* inv_norm,
                       const double* colsum, int Bglobal, double* conn, void* stream) {
    LAUNCH((infl_conn_kernel<TIN>), dim3((b + 3) / 4, 2), dim3(256), stream, (const TIN*)xv, (const TIN*)xt, ldv, ldt, b, Din,
           inv_norm, colsum, Bglobal, conn);
    return launch_status("infl_conn_kernel");
}
static int infl_args(const void* xv, const void* xt, long ldv, long ldt, int b, int Din) {
    if (!xv || !xt) return fail(CROSSCLR_E_ARG, "NULL argument");
    if (b < 1 || Din < 1 || Din > CROSSCLR_INFL_MAX_DIN) return fail(CROSSCLR_E_ARG, "need b >= 1 and 1 <= Din <= %d", CROSSCLR_INFL_MAX_DIN);
    if (ldv < Din || ldt < Din) return fail(CROSSCLR_E_ARG, "row stride smaller than Din");
    return CROSSCLR_OK;
}

extern "C" int crossclr_influence_colsum(const void* x_video, const void* x_text, long ld_video, long ld_text, int in_dtype,
                                         int b, int Din, float* inv_norm, float* partial_ws, double* colsum, void* stream) {
    static_assert(kInflBlocks == CROSSCLR_INFL_BLOCKS && 256 * kInflMaxCols == CROSSCLR_INFL_MAX_DIN, "header out of sync");
    if (int rc = infl_args(x_video, x_text, ld_video, ld_text, b, Din)) return rc;
    if (!inv_norm || !partial_ws || !colsum) return fail(CROSSCLR_E_ARG, "NULL argument");
    switch (in_dtype) {
        case CROSSCLR_IN_F32: return infl_colsum_t<float>(x_video, x_text, ld_video, ld_text, b, Din, inv_norm, partial_ws, colsum, stream);
        case CROSSCLR_IN_F64: return infl_colsum_t<double>(x_video, x_text, ld_video, ld_text, b, Din, inv_norm, partial_ws, colsum, stream);
        case CROSSCLR_IN_F16: return infl_colsum_t<in_f16>(x_video, x_text, ld_video, ld_text, b, Din, inv_norm, partial_ws, colsum, stream);
        case CROSSCLR_IN_BF16: return infl_colsum_t<in_bf16>(x_video, x_text, ld_video, ld_text, b, Din, inv_norm, partial_ws, colsum, stream);
    }
    return fail(CROSSCLR_E_ARG, "bad in_dtype %d", in_dtype);
}

extern "C" int crossclr_influence_conn(const void* x_video, const void* x_text, long ld_video, long ld_text, int in_dtype,
                                       int b, int Din, const float* inv_norm, const double* colsum_total, int B_global,
                                       double* conn, void* stream) {
    if (int rc = infl_args(x_video, x_text, ld_video, ld_text, b, Din)) return rc;
    if (!inv_norm || !colsum_total || !conn || B_global < b) return fail(CROSSCLR_E_ARG, "NULL argument / B_global < b");
    switch (in_dtype) {
        case CROSSCLR_IN_F32: return infl_conn_t<float>(x_video, x_text, ld_video, ld_text, b, Din, inv_norm, colsum_total, B_global, conn, stream);
        case CROSSCLR_IN_F64: return infl_conn_t<double>(x_video, x_text, ld_video, ld_text, b, Din, inv_norm, colsum_total, B_global, conn, stream);
        case CROSSCLR_IN_F16: return infl_conn_t<in_f16>(x_video, x_text, ld_video, ld_text, b, Din, inv_norm, colsum_total, B_global, conn, stream);
        case CROSSCLR_IN_BF16: return infl_conn_t<in_bf16>(x_video, x_text, ld_video, ld_text, b, Din, inv_norm, colsum_total, B_global, conn, stream);
    }
    return fail(CROSSCLR_E_ARG, "bad in_dtype %d", in_dtype);
}

extern "C" int crossclr_influence_finish(const crossclr_plan* plan, const double* conn_all, float score_threshold,
                                         float temperature_weights, float* neg_scale, float* loss_weight, void* stream) {
    if (!plan || !conn_all || !neg_scale || !loss_weight) return fail(CROSSCLR_E_ARG, "NULL argument");
    if (!(temperature_weights > 0.f)) return fail(CROSSCLR_E_ARG, "temperature_weights must be > 0");
    LAUNCH(infl_finish_kernel, dim3(2), dim3(1024), stream, conn_all, plan->world, plan->rank, plan->b, plan->bpad,
           (double)score_threshold, (double)temperature_weights, neg_scale, loss_weight);
    return launch_status("infl_finish_kernel");
}

// The sustained rate of the bf16 matrix pipe on THIS device, measured in the run that quotes it (bench.py: `roofline.sustained_mfma`):
// 256 blocks x 4 waves (one per SIMD), 16 independent accumulator chains of v_mfma_f32_32x32x16_bf16 per wave, nothing else in the
// loop; operands pseudo-random (toggling data: what the package power limit allows a real kernel) or all zero (`zero_operands`).
// One launch executes blocks * 4 * iters * 16 MFMAs of 32768 flop.
namespace crossclr {
#ifndef CROSSCLR_EMU
__global__ void __launch_bounds__(256, 1) mfma_sustained_kernel(float* out, int iters, unsigned seed, int zero_operands) {
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        x = x * 1664525u + 1013904223u;
        a[j] = zero_operands ? (__bf16)0.f : (__bf16)((float)(x >> 8) * (1.f / 16777216.f) - 0.5f);
        x = x * 1664525u + 1013904223u;
        b[j] = zero_operands ? (__bf16)0.f : (__bf16)((float)(x >> 8) * (1.f / 16777216.f) - 0.5f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = mfma_32x32x16_bf16(a, b, acc[i]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
#endif
}  // namespace crossclr

// ------------------------------------------------------------------------------------------------
// The whole step behind two calls (include/crossclr.h, ABI 6 / 7): the kernel-selection policy that used to live in the Python module.
// Composition of the entry points above.  crossclr_step_plan decides everything -- it is the only one of the three that reads the
// environment -- and writes the decision into the layout; forward and backward act on the layout they are handed (sealed with a check word).
static size_t step_max_stash_bytes() {
    const char* e = getenv("CROSSCLR_MAX_STASH_GB");
    double gb = 8.0;
    if (e) {
        char* end = nullptr;
        const double v = strtod(e, &end);
        if (end != e && isfinite(v)) gb = v < 0.0 ? 0.0 : (v > 1024.0 ? 1024.0 : v);      // (garbage: the default; negative: no stash)
    }
    return (size_t)(gb * (double)((size_t)1 << 30));
}
// Padded widths at which the fragment-major saved backward beats the LDS-staged one (profiles/r03_xf_widths.txt: wins from 512 up, ties at
// 384, loses at 256 / 128), with the row floor below which crossclr_normalize_xf's second copy is not paid back; CROSSCLR_XF_WIDTHS="128,256"
// (or "") overrides the widths and drops the floor (tuning / tests)
static bool step_use_xf(const crossclr_plan* p) {
    const char* e = getenv("CROSSCLR_XF_WIDTHS");
    if (!e) {
        static const int widths[] = {512, 768, 1024, 1152, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 8192};
        bool in = false;
        for (int w : widths) in = in || p->Dpad == w;
        return in && p->bpad >= (p->Dpad <= 512 ? 2048 : 4096);
    }
    for (const char* c = e; *c;) {
        char* end;
        const long w = strtol(c, &end, 10);
        if (end == c) { ++c; continue; }
        if (w == p->Dpad) return true;
        c = end;
    }
    return false;
}
static size_t step_align(size_t x) { return (x + 255) / 256 * 256; }
// FNV-1a over the plan and every layout field in front of `check`: a layout the library did not write for this plan is refused
static unsigned step_check(const crossclr_plan* plan, const crossclr_step_layout* L) {
    unsigned h = 2166136261u;
    auto mix = [&](const void* p, size_t n) {
        const unsigned char* c = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < n; ++i) h = (h ^ c[i]) * 16777619u;
    };
    // (field by field: struct padding is not part of the value)
    const int pi[] = {plan->b, plan->D, plan->world, plan->rank, plan->mode, plan->bpad, plan->Dpad, plan->fast_path, plan->fast_bwd,
                      plan->fwd_blocks, plan->fwd_slots, plan->bwd_slices, plan->loss_ws_doubles};
    const size_t pz[] = {plan->fwd_ws_floats, plan->operand_bytes, plan->gbuf_bytes, plan->stash_bytes, plan->xf_bytes};
    const size_t lz[] = {L->total_bytes, L->persistent_bytes, L->transient_bytes, L->backward_scratch_bytes, L->xhat, L->inv_norm, L->diag,
                         L->logz, L->rz, L->wrz, L->part, L->shift, L->xf, L->stash, L->gbuf, L->ticket, L->stash_bytes, L->xf_bytes};
    const int li[] = {(int)L->flags, L->two_pass, L->saved, L->backward_kernel};
    mix(pi, sizeof(pi)); mix(pz, sizeof(pz)); mix(lz, sizeof(lz)); mix(li, sizeof(li));
    mix(&L->temperature, sizeof(float)); mix(&L->negative_weight, sizeof(float));
    return h ? h : 1u;
}

extern "C" int crossclr_step_plan(const crossclr_plan* plan, float temperature, float negative_weight, unsigned flags,
                                  size_t workspace_bytes, crossclr_step_layout* L) {
    if (!plan || !L) return fail(CROSSCLR_E_ARG, "NULL argument");
    if (plan->world != 1) return fail(CROSSCLR_E_ARG, "crossclr_step_* is the single-device step (plan->world == 1); sharded runs compose the fine-grained entry points around their collectives");
    if (!(temperature > 0.f) || !isfinite(temperature) || !isfinite(negative_weight)) return fail(CROSSCLR_E_ARG, "temperature must be > 0 and finite, negative_weight finite");
    const bool two_pass = needs_row_shift(temperature, negative_weight);
    const bool fwd_only = (flags & CROSSCLR_STEP_FORWARD_ONLY) != 0;
    const bool may_save = !(flags & (CROSSCLR_STEP_NO_SAVE | CROSSCLR_STEP_FORWARD_ONLY));
    const bool eager = (flags & CROSSCLR_STEP_EAGER) && !fwd_only;
    const size_t stash_full = two_pass ? crossclr_stash_bytes_s(plan) : plan->stash_bytes;
    const size_t max_stash = step_max_stash_bytes();
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool saved = attempt == 0 && may_save && stash_full > 0 && stash_full <= max_stash;
        if (attempt == 0 && !saved) continue;
        // which saved backward: the fragment-major copy is only laid out (and written by the forward's first kernel) when a kernel that reads it is allowed
        int backward_kernel = saved ? 1 : 0;
        if (saved && !two_pass && plan->xf_bytes > 0 && step_use_xf(plan)) {
            const char* e = getenv("CROSSCLR_XFP");
            const bool xfp_ok = plan->stash_bytes < ((size_t)1 << 32) && !(e && e[0] == '0') && !(flags & CROSSCLR_STEP_NO_XFP);
            const bool xf1_ok = plan->Dpad <= 1024 && !(flags & CROSSCLR_STEP_NO_XF);     // (wide plans: the pair kernel or the LDS-staged one)
            backward_kernel = xfp_ok ? 3 : (xf1_ok ? 2 : 1);
        }
        const bool xf = backward_kernel >= 2;
        memset(L, 0, sizeof(*L));
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += step_align(bytes); return o; };
        const size_t n2 = (size_t)2 * plan->bpad;
        L->shift = L->xf = L->stash = L->gbuf = CROSSCLR_STEP_NONE;
        // region 1 (persistent): what the backward call reads
        if (eager) {                    // the finish kernel alone: 1 / ||x|| and the gradient slices
            L->inv_norm = take(4 * n2);
            L->gbuf = take(plan->gbuf_bytes);
        } else if (!fwd_only) {         // the gradient product (from saved exponentials, or recomputed: also what a backward without `transient` does)
            L->xhat = take(plan->operand_bytes);
            L->inv_norm = take(4 * n2);
            L->rz = take(4 * n2);
            L->wrz = take(4 * n2);
            if (two_pass) L->shift = take(4 * n2);
        }
        L->persistent_bytes = off;
        // region 2 (transient)
        if (eager || fwd_only) {
            L->xhat = take(plan->operand_bytes);
            if (fwd_only) L->inv_norm = take(4 * n2);
            L->rz = take(4 * n2);
            L->wrz = take(4 * n2);
            if (two_pass) L->shift = take(4 * n2);
        }
        L->diag = take(4 * (size_t)plan->bpad);
        L->logz = take(4 * n2);
        L->part = take(4 * plan->fwd_ws_floats);
        L->ticket = take(4);
        if (xf) L->xf = take(plan->xf_bytes);
        if (saved) L->stash = take(stash_full);
        L->xf_bytes = xf ? plan->xf_bytes : 0;
        L->stash_bytes = saved ? stash_full : 0;
        L->total_bytes = off;
        L->transient_bytes = off - L->persistent_bytes;
        L->backward_scratch_bytes = (fwd_only || eager) ? 0 : plan->gbuf_bytes;
        L->temperature = temperature;
        L->negative_weight = negative_weight;
        L->flags = flags;
        L->two_pass = two_pass ? 1 : 0;
        L->saved = saved ? 1 : 0;
        L->backward_kernel = backward_kernel;
        L->check = step_check(plan, L);
        if (workspace_bytes == 0 || off <= workspace_bytes) return CROSSCLR_OK;
    }
    return fail(CROSSCLR_E_WORKSPACE, "workspace of %zu bytes is below the recomputing layout's %zu", workspace_bytes, L->total_bytes);
}

namespace {
struct StepBufs {
    crossclr_step_layout L;
    unsigned char *persistent, *transient;
    unsigned char* at(size_t off) const {
        if (off == CROSSCLR_STEP_NONE) return nullptr;
        if (off < L.persistent_bytes) return persistent + off;
        return transient ? transient + (off - L.persistent_bytes) : nullptr;
    }
    void* xhat() const { return at(L.xhat); }
    float* f(size_t off) const { return reinterpret_cast<float*>(at(off)); }
    void* v(size_t off) const { return at(off); }
};
}  // namespace
static int step_bind(const crossclr_plan* plan, const crossclr_step_layout* layout, void* persistent, void* transient, bool transient_needed,
                     StepBufs* B) {
    if (!plan || !layout) return fail(CROSSCLR_E_ARG, "NULL argument");
    if (layout->check == 0 || layout->check != step_check(plan, layout))
        return fail(CROSSCLR_E_ARG, "this crossclr_step_layout was not written by crossclr_step_plan for this plan (or was modified since)");
    if (layout->persistent_bytes > 0 && !persistent) return fail(CROSSCLR_E_ARG, "persistent is NULL (layout.persistent_bytes = %zu)", layout->persistent_bytes);
    if (transient_needed && layout->transient_bytes > 0 && !transient) return fail(CROSSCLR_E_ARG, "transient is NULL (layout.transient_bytes = %zu)", layout->transient_bytes);
    B->L = *layout;
    B->persistent = static_cast<unsigned char*>(persistent);
    B->transient = static_cast<unsigned char*>(transient);
    return CROSSCLR_OK;
}
static int step_gradient_product(const crossclr_plan* plan, const StepBufs& B, const float* k, float* gbuf, bool from_saved, void* stream);

extern "C" int crossclr_step_forward(const crossclr_plan* plan, const crossclr_step_layout* layout, const void* video, const void* text,
                                     long ld_video, long ld_text, int in_dtype, const crossclr_sample_weights* sw,
                                     void* persistent, void* transient, double* loss_ws, void* stream) {
    if (!loss_ws || !video || !text) return fail(CROSSCLR_E_ARG, "NULL argument");
    StepBufs B;
    if (int rc = step_bind(plan, layout, persistent, transient, true, &B)) return rc;
    const crossclr_step_layout& L = B.L;
    const float temperature = L.temperature, negative_weight = L.negative_weight;
    const unsigned flags = L.flags;
    const float* k = sw ? sw->neg_scale_rows : nullptr;
    const float* lw = sw ? sw->loss_weight : nullptr;
    const crossclr_sample_weights sw_k = {k, k, nullptr}, sw_klw = {k, k, lw};
    const crossclr_sample_weights* pk = k ? &sw_k : nullptr;
    const crossclr_sample_weights* pklw = (k || lw) ? &sw_klw : nullptr;
    const bool pre = (flags & CROSSCLR_STEP_PRENORMALIZED) != 0;
    int rc;
    // loss.py:79-80 (+ the packed operand, 1 / ||x||, the positive pairs' cosines; with a fragment-major saved backward to follow: its operand copy)
    int* ticket = reinterpret_cast<int*>(B.at(L.ticket));      // cleared by this first kernel, taken by the finish kernel's blocks
    if (L.xf != CROSSCLR_STEP_NONE)
        rc = pre ? normalize_xf_any<false>(plan, video, text, ld_video, ld_text, in_dtype, B.xhat(), B.v(L.xf), B.f(L.inv_norm), B.f(L.diag), stream, ticket)
                 : normalize_xf_any<true>(plan, video, text, ld_video, ld_text, in_dtype, B.xhat(), B.v(L.xf), B.f(L.inv_norm), B.f(L.diag), stream, ticket);
    else
        rc = pre ? normalize_any<false>(plan, video, text, ld_video, ld_text, in_dtype, B.xhat(), B.f(L.inv_norm), B.f(L.diag), stream, ticket)
                 : normalize_any<true>(plan, video, text, ld_video, ld_text, in_dtype, B.xhat(), B.f(L.inv_norm), B.f(L.diag), stream, ticket);
    if (rc) return rc;
    if (L.two_pass) {      // loss.py:60's float64 soft-max takes the row maximum; so do these: row maxima, then sums relative to them
        rc = crossclr_forward_rowmax(plan, B.xhat(), B.xhat(), 1, plan->rank, -1, temperature, negative_weight, pk, B.f(L.part), B.f(L.shift), 0, stream);
        if (rc) return rc;
        rc = L.saved ? crossclr_forward_save_s(plan, B.xhat(), temperature, negative_weight, pk, B.f(L.shift), B.f(L.part), 0, B.v(L.stash), stream)
                     : crossclr_forward_s(plan, B.xhat(), B.xhat(), 1, plan->rank, -1, temperature, negative_weight, pk, B.f(L.shift), B.f(L.part), 0, stream);
        if (rc) return rc;
        rc = forward_finish_impl(plan, B.f(L.part), plan->fwd_slots, B.f(L.diag), temperature, negative_weight, pklw, B.f(L.shift),
                                 B.f(L.logz), B.f(L.rz), B.f(L.wrz), loss_ws, stream, ticket);
        if (rc || L.gbuf == CROSSCLR_STEP_NONE) return rc;
        return step_gradient_product(plan, B, k, B.f(L.gbuf), L.saved != 0, stream);
    }
    // loss.py:83-100, 59-60: soft-max denominators of the local block (and, saving, its exponentials)
    rc = L.saved ? crossclr_forward_save(plan, B.xhat(), temperature, negative_weight, pk, B.f(L.part), 0, B.v(L.stash), stream)
                 : crossclr_forward_w(plan, B.xhat(), B.xhat(), 1, plan->rank, -1, temperature, negative_weight, pk, B.f(L.part), 0, stream);
    if (rc) return rc;
    // loss.py:60 (-log), :111-114
    rc = forward_finish_impl(plan, B.f(L.part), plan->fwd_slots, B.f(L.diag), temperature, negative_weight, pklw, nullptr, B.f(L.logz), B.f(L.rz),
                             B.f(L.wrz), loss_ws, stream, ticket);
    if (rc || L.gbuf == CROSSCLR_STEP_NONE) return rc;
    return step_gradient_product(plan, B, k, B.f(L.gbuf), L.saved != 0, stream);      // CROSSCLR_STEP_EAGER
}

// autograd of loss.py:83-112: gbuf = d(loss)/d(unit rows), unscaled, in column slices (independent of grad_out).
// from_saved = false: the recomputing product (a step that did not save, or a backward whose caller has released the transient region)
static int step_gradient_product(const crossclr_plan* plan, const StepBufs& B, const float* k, float* gbuf, bool from_saved, void* stream) {
    const crossclr_step_layout& L = B.L;
    const float temperature = L.temperature, negative_weight = L.negative_weight;
    const crossclr_sample_weights sw_k = {k, k, nullptr};
    const crossclr_sample_weights* pk = k ? &sw_k : nullptr;
    float *rz = B.f(L.rz), *wrz = B.f(L.wrz);
    if (L.two_pass)
        return from_saved ? crossclr_backward_saved_s(plan, B.xhat(), B.v(L.stash), temperature, negative_weight, rz, wrz, pk, gbuf, 0, stream)
                          : crossclr_backward_s(plan, B.xhat(), B.xhat(), 1, plan->rank, -1, temperature, negative_weight, rz, wrz, rz, wrz, pk,
                                                B.f(L.shift), B.f(L.shift), gbuf, 0, stream);
    if (from_saved) {
        switch (L.backward_kernel) {
            case 3: return crossclr_backward_saved_xfp(plan, B.v(L.xf), B.v(L.stash), temperature, negative_weight, rz, wrz, pk, gbuf, 0, stream);
            case 2: return crossclr_backward_saved_xf(plan, B.v(L.xf), B.v(L.stash), temperature, negative_weight, rz, wrz, pk, gbuf, 0, stream);
            default: return crossclr_backward_saved(plan, B.xhat(), B.v(L.stash), temperature, negative_weight, rz, wrz, pk, gbuf, 0, stream);
        }
    }
    return crossclr_backward_w(plan, B.xhat(), B.xhat(), 1, plan->rank, -1, temperature, negative_weight, rz, wrz, rz, wrz, pk, gbuf, 0, stream);
}

extern "C" int crossclr_step_backward(const crossclr_plan* plan, const crossclr_step_layout* layout, const void* video, const void* text,
                                      long ld_video, long ld_text, int in_dtype, const crossclr_sample_weights* sw,
                                      void* persistent, void* transient, void* scratch, const double* grad_out,
                                      void* grad_video, void* grad_text, long ld_gvideo, long ld_gtext, void* stream) {
    if (!video || !text || !grad_out || !grad_video || !grad_text) return fail(CROSSCLR_E_ARG, "NULL argument");
    StepBufs B;
    if (int rc = step_bind(plan, layout, persistent, transient, false, &B)) return rc;
    const crossclr_step_layout& L = B.L;
    if (L.flags & CROSSCLR_STEP_FORWARD_ONLY) return fail(CROSSCLR_E_ARG, "the forward of this step was declared CROSSCLR_STEP_FORWARD_ONLY");
    const float* k = sw ? sw->neg_scale_rows : nullptr;
    const float* lw = sw ? sw->loss_weight : nullptr;
    const crossclr_sample_weights sw_lw = {nullptr, nullptr, lw};
    float* gbuf;
    if (L.gbuf != CROSSCLR_STEP_NONE) {
        gbuf = B.f(L.gbuf);                 // CROSSCLR_STEP_EAGER: the forward call enqueued the gradient product already
    } else {
        if (!scratch) return fail(CROSSCLR_E_ARG, "scratch is NULL (layout.backward_scratch_bytes bytes)");
        gbuf = static_cast<float*>(scratch);
        // (transient == NULL: released after an earlier backward through this step -- everything the recomputing product reads is persistent)
        if (int rc = step_gradient_product(plan, B, k, gbuf, L.saved != 0 && transient != nullptr, stream)) return rc;
    }
    // autograd of loss.py:79-80 + the positive-pair term, x grad_out, in the input dtype
    return crossclr_backward_finish_p(plan, gbuf, video, text, ld_video, ld_text, in_dtype, B.f(L.inv_norm), L.temperature, lw ? &sw_lw : nullptr, grad_out,
                                      grad_video, grad_text, ld_gvideo, ld_gtext, (L.flags & CROSSCLR_STEP_PRENORMALIZED) ? 1 : 0, stream);
}

// ------------------------------------------------------------------------------------------------
// Second-order terms (include/crossclr.h, ABI 7; crossclr_kernels_hvp.h): what autograd's double backward through loss.py:79-114 computes,
// composed of the exact-fp32 first-order entry points above and the two passes of hvp_kernel.
namespace {
struct HvpLayout {
    size_t xhat, vhat, inv_norm, diag, logz, rz, wrz, shift, drz, dwrz, part, loss_ws, gbuf1, dzpart, gbuf2, dgo_rows, total;
    int nz;
};
}  // namespace
static int hvp_layout(const crossclr_plan* plan, HvpLayout* H) {
    if (!plan) return fail(CROSSCLR_E_ARG, "plan is NULL");
    if (plan->world != 1) return fail(CROSSCLR_E_ARG, "crossclr_second_order is single-device (plan->world == 1)");
    if (plan->mode != CROSSCLR_MODE_FP32 || plan->fast_path) return fail(CROSSCLR_E_ARG, "crossclr_second_order takes a CROSSCLR_MODE_FP32 plan (second-order terms are formed from exact-fp32 products)");
    const size_t n2 = (size_t)2 * plan->bpad;
    const int rb = 2 * plan->bpad / 64, ntiles = 2 * plan->bpad / 64;
    int nz = (1024 + rb - 1) / rb;
    if (nz > ntiles) nz = ntiles;
    if (nz > 16) nz = 16;
    if (nz < 1) nz = 1;
    H->nz = nz;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += step_align(bytes); return o; };
    H->xhat = take(plan->operand_bytes);
    H->vhat = take(plan->operand_bytes);
    H->inv_norm = take(4 * n2);
    H->diag = take(4 * (size_t)plan->bpad);
    H->logz = take(4 * n2);
    H->rz = take(4 * n2);
    H->wrz = take(4 * n2);
    H->shift = take(4 * n2);
    H->drz = take(4 * n2);
    H->dwrz = take(4 * n2);
    H->part = take(4 * plan->fwd_ws_floats);
    H->loss_ws = take(8 * (size_t)(plan->loss_ws_doubles > 2 ? plan->loss_ws_doubles : 2));
    H->gbuf1 = take(plan->gbuf_bytes);
    H->dzpart = take(4 * (size_t)nz * n2);
    H->gbuf2 = take(4 * (size_t)nz * n2 * plan->Dpad);
    H->dgo_rows = take(8 * n2);
    H->total = off;
    return CROSSCLR_OK;
}
extern "C" size_t crossclr_second_order_workspace_bytes(const crossclr_plan* plan) {
    HvpLayout H;
    return hvp_layout(plan, &H) ? 0 : H.total;
}

template <typename TIN>
static int second_order_impl(const crossclr_plan* plan, const HvpLayout& H, const Geo& g, const TIN* video, const TIN* text, long ldv, long ldt,
                             float temperature, float negative_weight, const float* k, const float* lw, int prenormalized, bool two_pass,
                             const TIN* uvideo, const TIN* utext, long lduv, long ldut, const double* grad_out, unsigned char* w,
                             TIN* hvideo, TIN* htext, long ldhv, long ldht, double* d_grad_out, void* stream) {
    auto F = [&](size_t off) { return reinterpret_cast<float*>(w + off); };
    const int n2 = 2 * plan->bpad;
    const float* shift = two_pass ? F(H.shift) : nullptr;
    // v = the tangent of the unit rows along u
    LAUNCH((hvp_tangent_kernel<TIN>), dim3((unsigned)((n2 + 3) / 4)), dim3(256), stream, video, text, ldv, ldt, uvideo, utext, lduv, ldut, g,
           F(H.inv_norm), prenormalized, F(H.vhat));
    if (int rc = launch_status("hvp_tangent_kernel")) return rc;
    const int rb = n2 / 64, ntiles = n2 / 64;
    const int tps = (ntiles + H.nz - 1) / H.nz;
    const float* X = F(H.xhat);
    const float* V = F(H.vhat);
    // pass 1: dZ, then the tangents of omega / Z
    if (k) LAUNCH((hvp_kernel<64, true, 1>), dim3(rb, 1, H.nz), dim3(256), stream, X, V, g, F(H.rz), F(H.wrz), kNoF, kNoF, k, shift, F(H.dzpart), tps);
    else LAUNCH((hvp_kernel<64, false, 1>), dim3(rb, 1, H.nz), dim3(256), stream, X, V, g, F(H.rz), F(H.wrz), kNoF, kNoF, kNoF, shift, F(H.dzpart), tps);
    if (int rc = launch_status("hvp_kernel (row sums)")) return rc;
    LAUNCH(hvp_stats_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), stream, F(H.dzpart), H.nz, n2, F(H.rz), lw, negative_weight, F(H.drz), F(H.dwrz));
    if (int rc = launch_status("hvp_stats_kernel")) return rc;
    // pass 2: dG in column slices
#define CROSSCLR_LH(DC, NS)                                                                                                                          \
    do {                                                                                                                                             \
        if (k) LAUNCH((hvp_kernel<DC, true, 2, NS>), dim3(rb, plan->Dpad / (DC * NS), H.nz), dim3(256), stream, X, V, g, F(H.rz), F(H.wrz), F(H.drz), \
                      F(H.dwrz), k, shift, F(H.gbuf2), tps);                                                                                         \
        else LAUNCH((hvp_kernel<DC, false, 2, NS>), dim3(rb, plan->Dpad / (DC * NS), H.nz), dim3(256), stream, X, V, g, F(H.rz), F(H.wrz), F(H.drz),  \
                    F(H.dwrz), kNoF, shift, F(H.gbuf2), tps);                                                                                        \
    } while (0)
    // (two 256-column slices per block where they divide the row: S and T of a tile are evaluated once for 512 output columns)
    if (plan->Dpad % 512 == 0) CROSSCLR_LH(256, 2);
    else if (plan->Dpad % 256 == 0) CROSSCLR_LH(256, 1);
    else if (plan->Dpad % 128 == 0) CROSSCLR_LH(128, 1);
    else CROSSCLR_LH(64, 1);
#undef CROSSCLR_LH
    if (int rc = launch_status("hvp_kernel (product)")) return rc;
    // the row-local chain: normalisation, positive pairs, x grad_out; and <u, dL/d(rows)>
    double* dgo_rows = reinterpret_cast<double*>(w + H.dgo_rows);
    LAUNCH((hvp_finish_kernel<TIN>), dim3((unsigned)((2 * plan->b + 3) / 4)), dim3(256), stream, F(H.gbuf1), plan->bwd_slices, F(H.gbuf2), H.nz, video, text,
           ldv, ldt, uvideo, utext, lduv, ldut, g, F(H.inv_norm), 1.f / temperature, plan->b * plan->world, grad_out, lw, prenormalized, hvideo, htext,
           ldhv, ldht, dgo_rows);
    if (int rc = launch_status("hvp_finish_kernel")) return rc;
    LAUNCH(hvp_reduce_kernel, dim3(1), dim3(64), stream, dgo_rows, 2 * plan->b, d_grad_out);
    return launch_status("hvp_reduce_kernel");
}

extern "C" int crossclr_second_order(const crossclr_plan* plan, const void* video, const void* text, long ld_video, long ld_text, int in_dtype,
                                     float temperature, float negative_weight, const crossclr_sample_weights* sw, int prenormalized,
                                     const void* u_video, const void* u_text, long ld_uvideo, long ld_utext, const double* grad_out,
                                     void* workspace, size_t workspace_bytes, void* h_video, void* h_text, long ld_hvideo, long ld_htext,
                                     double* d_grad_out, void* stream) {
    if (!video || !text || !u_video || !u_text || !grad_out || !workspace || !h_video || !h_text || !d_grad_out) return fail(CROSSCLR_E_ARG, "NULL argument");
    HvpLayout H;
    if (int rc = hvp_layout(plan, &H)) return rc;
    if (workspace_bytes < H.total) return fail(CROSSCLR_E_WORKSPACE, "workspace of %zu bytes, crossclr_second_order_workspace_bytes says %zu", workspace_bytes, H.total);
    if (prenormalized != 0 && prenormalized != 1) return fail(CROSSCLR_E_ARG, "prenormalized must be 0 or 1");
    if (ld_video < plan->D || ld_text < plan->D || ld_uvideo < plan->D || ld_utext < plan->D || ld_hvideo < plan->D || ld_htext < plan->D)
        return fail(CROSSCLR_E_ARG, "row stride smaller than D");
    unsigned char* w = static_cast<unsigned char*>(workspace);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(w + off); };
    const float* k = sw ? sw->neg_scale_rows : nullptr;
    const float* lw = sw ? sw->loss_weight : nullptr;
    const crossclr_sample_weights sw_k = {k, k, nullptr}, sw_klw = {k, k, lw};
    const crossclr_sample_weights* pk = k ? &sw_k : nullptr;
    const crossclr_sample_weights* pklw = (k || lw) ? &sw_klw : nullptr;
    const bool two_pass = needs_row_shift(temperature, negative_weight);
    Geo g;
    if (int rc = make_geo(plan, 1, plan->rank, -1, temperature, negative_weight, &g, true)) return rc;
    // the first-order pieces, exact fp32: unit rows, soft-max statistics, the gradient product (loss.py:79-112 and its autograd)
    int rc = prenormalized ? normalize_any<false>(plan, video, text, ld_video, ld_text, in_dtype, w + H.xhat, F(H.inv_norm), F(H.diag), stream)
                           : normalize_any<true>(plan, video, text, ld_video, ld_text, in_dtype, w + H.xhat, F(H.inv_norm), F(H.diag), stream);
    if (rc) return rc;
    double* loss_ws = reinterpret_cast<double*>(w + H.loss_ws);
    if (two_pass) {
        rc = crossclr_forward_rowmax(plan, w + H.xhat, w + H.xhat, 1, plan->rank, -1, temperature, negative_weight, pk, F(H.part), F(H.shift), 0, stream);
        if (rc) return rc;
        rc = crossclr_forward_s(plan, w + H.xhat, w + H.xhat, 1, plan->rank, -1, temperature, negative_weight, pk, F(H.shift), F(H.part), 0, stream);
        if (rc) return rc;
        rc = forward_finish_impl(plan, F(H.part), plan->fwd_slots, F(H.diag), temperature, negative_weight, pklw, F(H.shift), F(H.logz), F(H.rz), F(H.wrz),
                                 loss_ws, stream, nullptr);
        if (rc) return rc;
        rc = crossclr_backward_s(plan, w + H.xhat, w + H.xhat, 1, plan->rank, -1, temperature, negative_weight, F(H.rz), F(H.wrz), F(H.rz), F(H.wrz), pk,
                                 F(H.shift), F(H.shift), F(H.gbuf1), 0, stream);
    } else {
        rc = crossclr_forward_w(plan, w + H.xhat, w + H.xhat, 1, plan->rank, -1, temperature, negative_weight, pk, F(H.part), 0, stream);
        if (rc) return rc;
        rc = forward_finish_impl(plan, F(H.part), plan->fwd_slots, F(H.diag), temperature, negative_weight, pklw, nullptr, F(H.logz), F(H.rz), F(H.wrz),
                                 loss_ws, stream, nullptr);
        if (rc) return rc;
        rc = crossclr_backward_w(plan, w + H.xhat, w + H.xhat, 1, plan->rank, -1, temperature, negative_weight, F(H.rz), F(H.wrz), F(H.rz), F(H.wrz), pk,
                                 F(H.gbuf1), 0, stream);
    }
    if (rc) return rc;
    switch (in_dtype) {
#define CROSSCLR_SO(TIN) second_order_impl<TIN>(plan, H, g, (const TIN*)video, (const TIN*)text, ld_video, ld_text, temperature, negative_weight, k, lw, \
                                                prenormalized, two_pass, (const TIN*)u_video, (const TIN*)u_text, ld_uvideo, ld_utext, grad_out, w,       \
                                                (TIN*)h_video, (TIN*)h_text, ld_hvideo, ld_htext, d_grad_out, stream)
        case CROSSCLR_IN_F32: return CROSSCLR_SO(float);
        case CROSSCLR_IN_F64: return CROSSCLR_SO(double);
        case CROSSCLR_IN_F16: return CROSSCLR_SO(in_f16);
        case CROSSCLR_IN_BF16: return CROSSCLR_SO(in_bf16);
#undef CROSSCLR_SO
        default: return fail(CROSSCLR_E_ARG, "bad in_dtype %d", in_dtype);
    }
}

extern "C" int crossclr_mfma_sustained(float* out, int blocks, int iters, unsigned seed, int zero_operands, void* stream) {
#ifndef CROSSCLR_EMU
    if (!out || blocks < 1 || blocks > 4096 || iters < 1) return fail(CROSSCLR_E_ARG, "bad mfma_sustained arguments");
    LAUNCH(mfma_sustained_kernel, dim3(blocks), dim3(256), stream, out, iters, seed, zero_operands);
    return launch_status("mfma_sustained_kernel");
#else
    (void)out; (void)blocks; (void)iters; (void)seed; (void)zero_operands; (void)stream;
    return fail(CROSSCLR_E_ARG, "crossclr_mfma_sustained measures the device: not available in the host emulation");
#endif
}

extern "C" int crossclr_selftest(int which, const void* in, void* out, void* stream) {
    if (which < 0 || which > 4 || !in || !out) return fail(CROSSCLR_E_ARG, "bad selftest arguments");
    LAUNCH(selftest_kernel, dim3(1), dim3(64), stream, which, in, out);
    return launch_status("selftest_kernel");
}

#ifdef CROSSCLR_TIMING
// (variant builds only; not declared in include/crossclr.h) the marks of the most recent pipelined launch: [blocks][8] uint64
extern "C" int crossclr_debug_timing(unsigned long long* host_out, int nblocks) {
    if (!host_out || nblocks < 1 || nblocks > 1024) return fail(CROSSCLR_E_ARG, "bad timing request");
    if (hipDeviceSynchronize() != hipSuccess) return fail(CROSSCLR_E_HIP, "synchronize failed");
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_timing), sizeof(unsigned long long) * 8 * (size_t)nblocks) != hipSuccess)
        return fail(CROSSCLR_E_HIP, "reading the timing marks failed");
    return CROSSCLR_OK;
}
#endif
