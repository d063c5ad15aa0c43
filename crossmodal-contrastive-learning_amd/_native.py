"""ctypes binding of the C-ABI in include/crossclr.h.

The product path loads exactly one library: `libcrossclr_hip.so` (hipcc, gfx950), built in-tree by
`build.py`.  If it is missing the import of the loss module still works but the first call raises
-- there is NO CPU fallback.  Tests may inject the host emulation build of the same sources with
`use_library_for_testing()`; nothing else calls that function.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# CROSSCLR_HIP_LIBRARY: kernel-tuning knob -- point the binding at another hipcc build of the same
# sources (tools/build_variant.py) for A/B timing.  It must still be a "hip-gfx950" library.
HIP_LIBRARY = os.environ.get("CROSSCLR_HIP_LIBRARY", os.path.join(_HERE, "libcrossclr_hip.so"))

MODE_FP32, MODE_BF16 = 0, 1
IN_F32, IN_F16, IN_BF16, IN_F64 = 0, 1, 2, 3
E_RANGE = -2
ABI_VERSION = 7
LAUNCH_GROUPS = 8      # CROSSCLR_LAUNCH_GROUPS of include/crossclr.h


class Plan(ctypes.Structure):
    _fields_ = [("b", ctypes.c_int), ("D", ctypes.c_int), ("world", ctypes.c_int), ("rank", ctypes.c_int),
                ("mode", ctypes.c_int), ("bpad", ctypes.c_int), ("Dpad", ctypes.c_int),
                ("fast_path", ctypes.c_int), ("fast_bwd", ctypes.c_int), ("fwd_blocks", ctypes.c_int), ("fwd_slots", ctypes.c_int),
                ("fwd_ws_floats", ctypes.c_size_t),
                ("bwd_slices", ctypes.c_int),
                ("loss_ws_doubles", ctypes.c_int),
                ("operand_bytes", ctypes.c_size_t), ("gbuf_bytes", ctypes.c_size_t), ("stash_bytes", ctypes.c_size_t),
                ("xf_bytes", ctypes.c_size_t)]


class SampleWeights(ctypes.Structure):
    """crossclr_sample_weights: device pointers (0 = all ones)."""
    _fields_ = [("neg_scale_rows", ctypes.c_void_p), ("neg_scale_cols", ctypes.c_void_p), ("loss_weight", ctypes.c_void_p)]


class StepLayout(ctypes.Structure):
    """crossclr_step_layout: what crossclr_step_plan decided for one step and where the pieces live in its two workspace regions
    (offsets into [persistent | transient]); handed unmodified to crossclr_step_forward and crossclr_step_backward."""
    _fields_ = [(n, ctypes.c_size_t) for n in ("total_bytes", "persistent_bytes", "transient_bytes", "backward_scratch_bytes")] + \
               [(n, ctypes.c_size_t) for n in ("xhat", "inv_norm", "diag", "logz", "rz", "wrz", "part", "shift", "xf", "stash", "gbuf", "ticket")] + \
               [("stash_bytes", ctypes.c_size_t), ("xf_bytes", ctypes.c_size_t),
                ("temperature", ctypes.c_float), ("negative_weight", ctypes.c_float), ("flags", ctypes.c_uint),
                ("two_pass", ctypes.c_int), ("saved", ctypes.c_int), ("backward_kernel", ctypes.c_int), ("check", ctypes.c_uint)]


STEP_NO_SAVE, STEP_FORWARD_ONLY, STEP_PRENORMALIZED, STEP_NO_XFP, STEP_NO_XF, STEP_EAGER = 1, 2, 4, 8, 16, 32
STEP_NONE = ctypes.c_size_t(-1).value


class CrossCLRNativeError(RuntimeError):
    pass


_P = ctypes.c_void_p
_SIGNATURES = {
    "crossclr_abi_version": (ctypes.c_int, []),
    "crossclr_last_error": (ctypes.c_char_p, []),
    "crossclr_backend": (ctypes.c_char_p, []),
    "crossclr_make_plan": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.POINTER(Plan)]),
    "crossclr_normalize": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                          _P, _P, _P, _P]),
    "crossclr_forward": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_float, ctypes.c_float, _P, ctypes.c_int, _P]),
    "crossclr_forward_finish": (ctypes.c_int, [ctypes.POINTER(Plan), _P, ctypes.c_int, _P, ctypes.c_float,
                                               ctypes.c_float, _P, _P, _P, _P, _P]),
    "crossclr_backward": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_float, ctypes.c_float, _P, _P, _P, _P, _P, ctypes.c_int, _P]),
    "crossclr_backward_finish": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, _P, ctypes.c_long, ctypes.c_long,
                                                ctypes.c_int, _P, ctypes.c_float, _P, _P, _P, ctypes.c_long,
                                                ctypes.c_long, _P]),
    "crossclr_selftest": (ctypes.c_int, [ctypes.c_int, _P, _P, _P]),
    # ABI version 2: the same four entry points with per-sample weights
    "crossclr_forward_w": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_float, ctypes.c_float, ctypes.POINTER(SampleWeights), _P, ctypes.c_int, _P]),
    "crossclr_forward_finish_w": (ctypes.c_int, [ctypes.POINTER(Plan), _P, ctypes.c_int, _P, ctypes.c_float,
                                                 ctypes.c_float, ctypes.POINTER(SampleWeights), _P, _P, _P, _P, _P]),
    "crossclr_backward_w": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_float, ctypes.c_float, _P, _P, _P, _P, ctypes.POINTER(SampleWeights),
                                           _P, ctypes.c_int, _P]),
    "crossclr_backward_finish_w": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, _P, ctypes.c_long, ctypes.c_long,
                                                  ctypes.c_int, _P, ctypes.c_float, ctypes.POINTER(SampleWeights), _P, _P, _P,
                                                  ctypes.c_long, ctypes.c_long, _P]),
    "crossclr_forward_pairs": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                              ctypes.c_float, ctypes.POINTER(SampleWeights), _P, ctypes.c_int, _P, _P]),
    # ABI version 3: save-for-backward pair
    "crossclr_forward_save": (ctypes.c_int, [ctypes.POINTER(Plan), _P, ctypes.c_float, ctypes.c_float,
                                             ctypes.POINTER(SampleWeights), _P, ctypes.c_int, _P, _P]),
    "crossclr_backward_saved": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_float, ctypes.c_float, _P, _P,
                                               ctypes.POINTER(SampleWeights), _P, ctypes.c_int, _P]),
    # ABI version 4: the fragment-major operand copy (column tiles of the saved backward straight into MFMA fragments)
    "crossclr_normalize_xf": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                             _P, _P, _P, _P, _P]),
    "crossclr_pack_xf": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int, _P, _P, _P, _P, _P]),
    "crossclr_backward_saved_xf": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_float, ctypes.c_float, _P, _P,
                                                  ctypes.POINTER(SampleWeights), _P, ctypes.c_int, _P]),
    "crossclr_backward_saved_xfp": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_float, ctypes.c_float, _P, _P,
                                                  ctypes.POINTER(SampleWeights), _P, ctypes.c_int, _P]),
    # ABI version 3: rectangular blocks with saved exponentials
    "crossclr_rect_stash_bytes": (ctypes.c_size_t, [ctypes.POINTER(Plan), ctypes.c_int]),
    "crossclr_forward_rect_save": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                  ctypes.c_float, ctypes.POINTER(SampleWeights), _P, ctypes.c_int, _P, _P, _P]),
    "crossclr_backward_rect_saved": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                    ctypes.c_float, _P, _P, _P, _P, ctypes.POINTER(SampleWeights), _P, ctypes.c_int, _P]),
    "crossclr_rect_stash_bytes_s": (ctypes.c_size_t, [ctypes.POINTER(Plan), ctypes.c_int]),
    "crossclr_forward_rect_save_s": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                                    ctypes.POINTER(SampleWeights), _P, _P, _P, ctypes.c_int, _P, _P]),
    "crossclr_backward_rect_saved_s": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                      ctypes.c_float, _P, _P, _P, _P, ctypes.POINTER(SampleWeights), _P, ctypes.c_int, _P]),
    "crossclr_backward_rect_saved_t": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                      ctypes.c_float, _P, _P, _P, _P, ctypes.POINTER(SampleWeights), _P, _P]),
    # ABI version 5: remote blocks on the fragment-major operand (pair kernel), the copy made from received slices
    "crossclr_pack_xf_from_packed": (ctypes.c_int, [ctypes.POINTER(Plan), _P, ctypes.c_int, _P, _P]),
    "crossclr_backward_rect_saved_xfp": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                    ctypes.c_float, _P, _P, _P, _P, ctypes.POINTER(SampleWeights), _P, ctypes.c_int, _P]),
    "crossclr_backward_rect_saved_t_xfp": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                      ctypes.c_float, _P, _P, _P, _P, ctypes.POINTER(SampleWeights), _P, _P]),
    "crossclr_backward_ranks": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                               ctypes.c_float, _P, _P, _P, _P, ctypes.POINTER(SampleWeights), _P, ctypes.c_int, _P]),
    # ABI version 3: unit-vector inputs (caller-side fusion)
    "crossclr_pack": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int, _P, _P, _P, _P]),
    "crossclr_backward_finish_p": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, _P, ctypes.c_long, ctypes.c_long,
                                                  ctypes.c_int, _P, ctypes.c_float, ctypes.POINTER(SampleWeights), _P, _P, _P,
                                                  ctypes.c_long, ctypes.c_long, ctypes.c_int, _P]),
    # ABI version 3: producer-side fusion (projection head + L2-norm + pack)
    "crossclr_project_pack": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             _P, _P, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P, _P, _P]),
    "crossclr_project_pack_wf": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                _P, _P, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P, _P, _P]),
    "crossclr_project_dw_ws_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "crossclr_project_dw": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _P, _P, ctypes.c_long, _P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, _P, _P, _P, ctypes.c_long, ctypes.c_long, _P, _P, _P]),
    "crossclr_project_backward_prep": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_long, ctypes.c_long, _P, _P, _P, _P,
                                                      ctypes.c_long, _P]),
    # ABI version 3: two-pass soft-max for small temperatures
    "crossclr_needs_row_shift": (ctypes.c_int, [ctypes.c_float, ctypes.c_float]),
    "crossclr_forward_rowmax": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_float, ctypes.c_float, ctypes.POINTER(SampleWeights), _P, _P, ctypes.c_int, _P]),
    "crossclr_forward_s": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_float, ctypes.c_float, ctypes.POINTER(SampleWeights), _P, _P, ctypes.c_int, _P]),
    "crossclr_forward_finish_s": (ctypes.c_int, [ctypes.POINTER(Plan), _P, ctypes.c_int, _P, ctypes.c_float,
                                                 ctypes.c_float, ctypes.POINTER(SampleWeights), _P, _P, _P, _P, _P, _P]),
    "crossclr_backward_s": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_float, ctypes.c_float, _P, _P, _P, _P, ctypes.POINTER(SampleWeights),
                                           _P, _P, _P, ctypes.c_int, _P]),
    "crossclr_stash_bytes_s": (ctypes.c_size_t, [ctypes.POINTER(Plan)]),
    "crossclr_forward_save_s": (ctypes.c_int, [ctypes.POINTER(Plan), _P, ctypes.c_float, ctypes.c_float, ctypes.POINTER(SampleWeights),
                                               _P, _P, ctypes.c_int, _P, _P]),
    "crossclr_backward_saved_s": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_float, ctypes.c_float, _P, _P,
                                                 ctypes.POINTER(SampleWeights), _P, ctypes.c_int, _P]),
    "crossclr_forward_add": (ctypes.c_int, [ctypes.POINTER(Plan), _P, ctypes.c_int, _P, _P]),
    "crossclr_influence_colsum": (ctypes.c_int, [_P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                 _P, _P, _P, _P]),
    "crossclr_influence_conn": (ctypes.c_int, [_P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               _P, _P, ctypes.c_int, _P, _P]),
    "crossclr_influence_finish": (ctypes.c_int, [ctypes.POINTER(Plan), _P, ctypes.c_float, ctypes.c_float, _P, _P, _P]),
    # ABI version 3: score statistics of the inter-modal block (max-margin ranking loss, retrieval ranks)
    "crossclr_score_diag": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, _P]),
    "crossclr_score_rows": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_float, _P, _P, _P, _P, _P]),
    "crossclr_maxmargin_backward": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_float, _P, _P]),
    "crossclr_maxmargin_mask_bytes": (ctypes.c_size_t, [ctypes.POINTER(Plan)]),
    "crossclr_score_rows_save": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_float, _P, _P, _P, _P, _P, _P]),
    "crossclr_maxmargin_backward_saved": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, _P, _P]),
    "crossclr_maxmargin_backward_finish": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                                          _P, _P, _P, _P, _P, ctypes.c_long, ctypes.c_long, _P]),
    # ABI version 6: the whole single-device step behind two calls (the kernel-selection policy lives in the library);
    # ABI version 7: the layout crossclr_step_plan wrote is handed to both calls, the workspace is a persistent and a transient region
    "crossclr_step_plan": (ctypes.c_int, [ctypes.POINTER(Plan), ctypes.c_float, ctypes.c_float, ctypes.c_uint, ctypes.c_size_t,
                                          ctypes.POINTER(StepLayout)]),
    "crossclr_step_forward": (ctypes.c_int, [ctypes.POINTER(Plan), ctypes.POINTER(StepLayout), _P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                             ctypes.POINTER(SampleWeights), _P, _P, _P, _P]),
    "crossclr_step_backward": (ctypes.c_int, [ctypes.POINTER(Plan), ctypes.POINTER(StepLayout), _P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                              ctypes.POINTER(SampleWeights), _P, _P, _P, _P, _P, _P, ctypes.c_long, ctypes.c_long, _P]),
    # ABI version 7: second-order terms (double backward of the loss) in closed form on the device
    "crossclr_second_order_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Plan)]),
    "crossclr_second_order": (ctypes.c_int, [ctypes.POINTER(Plan), _P, _P, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                             ctypes.POINTER(SampleWeights), ctypes.c_int, _P, _P, ctypes.c_long, ctypes.c_long, _P, _P, ctypes.c_size_t,
                                             _P, _P, ctypes.c_long, ctypes.c_long, _P, _P]),
    # ABI version 7 (reporting aid): the kernel template the process's most recent forward (0) / gradient-product (1) launch went to
    "crossclr_last_kernel": (ctypes.c_char_p, [ctypes.c_int]),
    # measurement aid (bench.py: the matrix pipe's sustained rate on this device, in the run that quotes it)
    "crossclr_mfma_sustained": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_int, _P]),
}
INFL_BLOCKS = 256
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib: Optional[ctypes.CDLL] = None
_lib_path: Optional[str] = None
_injected = False


def _bind(path: str) -> ctypes.CDLL:
    lib = ctypes.CDLL(path)
    # CROSSCLR_AB_OLD_ABI=1 (kernel A/B timing against a build of an older commit, tools/ only): bind what the library has
    tolerant = os.environ.get("CROSSCLR_AB_OLD_ABI") == "1"
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError here = the library does not export the ABI
        except AttributeError:
            if tolerant:
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if lib.crossclr_abi_version() != ABI_VERSION and not tolerant:
        raise CrossCLRNativeError(f"{path}: ABI version {lib.crossclr_abi_version()} != {ABI_VERSION}")
    return lib


def library() -> ctypes.CDLL:
    """The native library; raises (loudly) when the HIP extension has not been built."""
    global _lib, _lib_path
    if _lib is None:
        if not os.path.exists(HIP_LIBRARY):
            raise CrossCLRNativeError(
                f"{HIP_LIBRARY} not found: the CrossCLR HIP extension is not built. Run "
                "`python __graft_entry__.py` (or crossmodal-contrastive-learning_amd/build.py). "
                "There is no CPU fallback.")
        _lib = _bind(HIP_LIBRARY)
        _lib_path = HIP_LIBRARY
    return _lib


def use_library_for_testing(path: Optional[str]) -> None:
    """TESTS ONLY: route the binding to another build of the same C-ABI (the host emulation build
    under tests/emu/), or back to the HIP library with None."""
    global _lib, _lib_path, _injected
    if path is None:
        _lib, _lib_path, _injected = None, None, False
    else:
        _lib, _lib_path, _injected = _bind(path), path, True


def injected_for_testing() -> bool:
    """True while use_library_for_testing() has routed the binding elsewhere (the tests' build re-reads its tuning variables per call)."""
    return _injected


def backend() -> str:
    return library().crossclr_backend().decode()


def library_path() -> Optional[str]:
    library()
    return _lib_path


def check(rc: int) -> None:
    if rc != 0:
        msg = library().crossclr_last_error().decode()
        raise CrossCLRNativeError(f"crossclr native call failed ({rc}): {msg}")


def make_plan(b: int, D: int, world: int, rank: int, mode: int) -> Plan:
    p = Plan()
    check(library().crossclr_make_plan(b, D, world, rank, mode, ctypes.byref(p)))
    return p
