"""Per-stage HIP-event timing of the hot path (used by bench.py for the roofline object).

Every stage is one C-ABI call = one kernel launch on torch's current stream, so torch.cuda.Event
(which records on that stream) brackets exactly that kernel."""
from __future__ import annotations

import ctypes
from typing import Dict

import torch

from . import _native as nat
from . import loss as L


def stage_times(video: torch.Tensor, text: torch.Tensor, temperature: float, negative_w: float,
                compute_mode: str, iters: int = 10, warmup: int = 2, negative_scale=None, loss_weight=None, settle: int = 0
                ) -> Dict[str, float]:
    """Median milliseconds per launch of each single-GPU stage (normalize, forward, forward_finish,
    backward, backward_finish; forward_save / backward_saved when the plan has the save-for-backward pair -- those two
    are what a training step launches, `forward` / `backward` are the recomputing entry points);
    with sample weights the `_w` entry points are the ones timed."""
    lib = nat.library()
    _, ws = L._forward_impl(video, text, temperature, negative_w, compute_mode, None, negative_scale, loss_weight,
                            save_for_backward=True)
    sw_k, sw_all, sw_lw = L._sw(ws.k_rows, ws.k_rows, None), L._sw(ws.k_rows, ws.k_rows, ws.lw), L._sw(None, None, ws.lw)
    plan, pp = ws.plan, ctypes.byref(ws.plan)
    dev = video.device
    p = L._ptr
    stream = L._stream_for(video)
    part = torch.empty(plan.fwd_ws_floats, dtype=torch.float32, device=dev)
    gbuf = torch.empty(plan.gbuf_bytes // 4, dtype=torch.float32, device=dev)
    go = torch.ones(1, dtype=torch.float64, device=dev)
    gv, gt = torch.empty_like(video), torch.empty_like(text)
    t, w = ws.temperature, ws.negative_w

    stash = ws.stash   # None when the plan has no save-for-backward path
    xf = ws.xf         # None when the plan has no fragment-major operand (then the saved backward stages column tiles through LDS)
    xf_name = L._saved_backward_entry(ws, plan, dev) if (xf is not None and stash is not None) else "crossclr_backward_saved"
    xf_entry = getattr(lib, xf_name)
    stages = {
        "normalize": (lambda: lib.crossclr_normalize_xf(pp, p(video), p(text), video.stride(0), text.stride(0), ws.in_dtype,
                                                        p(ws.xhat), p(xf), p(ws.inv_norm), p(ws.diag), stream)) if xf is not None else
                     (lambda: lib.crossclr_normalize(pp, p(video), p(text), video.stride(0), text.stride(0), ws.in_dtype,
                                                     p(ws.xhat), p(ws.inv_norm), p(ws.diag), stream)),
        "normalize_plain": (lambda: lib.crossclr_normalize(pp, p(video), p(text), video.stride(0), text.stride(0), ws.in_dtype,
                                                           p(ws.xhat), p(ws.inv_norm), p(ws.diag), stream)) if xf is not None else None,
        "forward": lambda: lib.crossclr_forward_w(pp, p(ws.xhat), p(ws.xhat), 1, 0, -1, t, w, sw_k, p(part), 0, stream),
        "forward_finish": lambda: lib.crossclr_forward_finish_w(pp, p(part), plan.fwd_slots, p(ws.diag), t, w, sw_all,
                                                                p(ws.logz), p(ws.rz), p(ws.wrz), p(ws.loss_sum), stream),
        "backward": lambda: lib.crossclr_backward_w(pp, p(ws.xhat), p(ws.xhat), 1, 0, -1, t, w, p(ws.rz), p(ws.wrz),
                                                    p(ws.rz), p(ws.wrz), sw_k, p(gbuf), 0, stream),
        "forward_save": (lambda: lib.crossclr_forward_save(pp, p(ws.xhat), t, w, sw_k, p(part), 0, p(stash), stream)) if stash is not None else None,
        "backward_saved": ((lambda: xf_entry(pp, p(xf if xf_name != "crossclr_backward_saved" else ws.xhat), p(stash), t, w, p(ws.rz), p(ws.wrz), sw_k, p(gbuf), 0, stream))
                           if xf is not None else
                           (lambda: lib.crossclr_backward_saved(pp, p(ws.xhat), p(stash), t, w, p(ws.rz), p(ws.wrz), sw_k,
                                                                p(gbuf), 0, stream))) if stash is not None else None,
        # (the fragment-major kernel with one tile per barrier interval, beside the pair kernel the step runs)
        "backward_saved_xf1": (lambda: lib.crossclr_backward_saved_xf(pp, p(xf), p(stash), t, w, p(ws.rz), p(ws.wrz), sw_k, p(gbuf), 0, stream))
                              if (stash is not None and xf is not None and xf_name != "crossclr_backward_saved_xf" and plan.Dpad <= 1024) else None,
        "backward_saved_lds": (lambda: lib.crossclr_backward_saved(pp, p(ws.xhat), p(stash), t, w, p(ws.rz), p(ws.wrz), sw_k,
                                                                   p(gbuf), 0, stream)) if (stash is not None and xf is not None) else None,
        "backward_finish": lambda: lib.crossclr_backward_finish_w(pp, p(gbuf), p(video), p(text), video.stride(0),
                                                                  text.stride(0), ws.in_dtype, p(ws.inv_norm), t, sw_lw,
                                                                  p(go), p(gv), p(gt), gv.stride(0), gt.stride(0), stream),
    }
    # settle: untimed rounds of the two heavy kernels first (a freshly acquired MI355X runs its first ~20-50 steps ~9 % slower; bench.py has
    # its own settle steps, the tuning tools ask for some here)
    for _ in range(settle):
        for name in ("forward_save", "backward_saved"):
            if stages.get(name) is not None:
                nat.check(stages[name]())
    out = {}
    # the stages a training step runs first, the recomputing entry points (the long `backward`) last: they are what is compared
    order = ["normalize", "normalize_plain", "forward_save", "forward_finish", "backward_saved", "backward_saved_xf1", "backward_saved_lds",
             "backward_finish", "forward", "backward"]
    for name in order:
        fn = stages.get(name)
        if fn is None:
            continue
        for _ in range(warmup):
            nat.check(fn())
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
        for i in range(iters):
            e0[i].record()
            nat.check(fn())
            e1[i].record()
        torch.cuda.synchronize(dev)
        ms = sorted(a.elapsed_time(b) for a, b in zip(e0, e1))
        out[name] = ms[len(ms) // 2]          # median per launch (a mean over ten launches moved by 8 % with one slow launch)
    out["fast_path"] = float(plan.fast_path)
    out["saved_path"] = float(stash is not None)
    out["xf_path"] = float(xf is not None and xf_name != "crossclr_backward_saved")
    out["xfp_path"] = float(xf_name == "crossclr_backward_saved_xfp")
    # the stages a training step actually runs
    out["step_forward"] = out.get("forward_save", out["forward"])
    out["step_backward"] = out.get("backward_saved", out["backward"])
    return out
