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
                compute_mode: str, iters: int = 10, warmup: int = 2, negative_scale=None, loss_weight=None, settle: int = 0,
                in_step: int = 0, forward_only: bool = False) -> Dict[str, float]:
    """Median milliseconds per launch of each single-GPU stage (normalize, forward, forward_finish,
    backward, backward_finish; forward_save / backward_saved when the plan has the save-for-backward pair -- those two
    are what a training step launches, `forward` / `backward` are the recomputing entry points);
    with sample weights the `_w` entry points are the ones timed.
    in_step > 0: additionally `in_step` passes of the step's own kernel SEQUENCE (normalize, forward [+ save], forward_finish, [saved] backward,
    backward_finish; forward_only: the first three), events between the launches: `in_step_<stage>` = the median duration of each kernel IN ITS
    PLACE in the step.  Back-to-back launches of one MFMA kernel alone keep the package at its power limit without the row kernels' pauses and
    run a few per cent slower than the same kernel inside a step: the in-step figure is the one that adds up to the step time."""
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
    if in_step > 0:
        seq = [("normalize", stages["normalize"]),
               ("forward", stages["forward"] if (forward_only or stages.get("forward_save") is None) else stages["forward_save"]),
               ("forward_finish", stages["forward_finish"])]
        if not forward_only:
            seq += [("backward", stages["backward_saved"] if stages.get("backward_saved") is not None else stages["backward"]),
                    ("backward_finish", stages["backward_finish"])]
        for _ in range(3):
            for _, fn in seq:
                nat.check(fn())
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)] for _ in range(in_step)]
        for row in evs:
            row[0].record()
            for k, (_, fn) in enumerate(seq):
                nat.check(fn())
                row[k + 1].record()
        torch.cuda.synchronize(dev)
        for k, (name, _) in enumerate(seq):
            ms = sorted(row[k].elapsed_time(row[k + 1]) for row in evs)
            out["in_step_" + name] = ms[len(ms) // 2]
        tot = sorted(row[0].elapsed_time(row[-1]) for row in evs)
        out["in_step_total"] = tot[len(tot) // 2]
    out["fast_path"] = float(plan.fast_path)
    out["saved_path"] = float(stash is not None)
    out["xf_path"] = float(xf is not None and xf_name != "crossclr_backward_saved")
    out["xfp_path"] = float(xf_name == "crossclr_backward_saved_xfp")
    # the stages a training step actually runs
    out["step_forward"] = out.get("forward_save", out["forward"])
    out["step_backward"] = out.get("backward_saved", out["backward"])
    return out


def remote_block_times(video: torch.Tensor, text: torch.Tensor, temperature: float, negative_w: float, iters: int = 5, warmup: int = 2,
                       recompute: bool = True, mode: int = nat.MODE_FP32) -> Dict[str, float]:
    """Sharded run of an exact-fp32 plan (or, mode = MODE_BF16, of a wide bf16 plan), rank 0 of 2 driven through the C-ABI on one GPU (`video` / `text` = the rows of BOTH ranks): median
    milliseconds per launch of the block against the other rank -- the forward that saves its exponentials and the saved backward
    (single-pass regime: crossclr_forward_rect_save / crossclr_backward_rect_saved; two-pass regime: the `_s` pair, U and Ut), and with
    `recompute` the recomputing pair beside them (crossclr_forward[_s] / crossclr_backward[_s] over the same columns)."""
    lib, p = nat.library(), L._ptr
    B, D = video.shape
    world, b, dev, stream = 2, B // 2, video.device, L._stream_for(video)
    plans = [nat.make_plan(b, D, world, r, mode) for r in range(world)]
    pl, pp = plans[0], ctypes.byref(plans[0])
    f32 = dict(dtype=torch.float32, device=dev)
    xall = torch.empty(world * pl.operand_bytes, dtype=torch.uint8, device=dev)
    inv, diag = torch.empty(2 * pl.bpad, **f32), torch.empty(pl.bpad, **f32)
    for r in range(world):
        nat.check(lib.crossclr_normalize(ctypes.byref(plans[r]), p(video[r * b:]), p(text[r * b:]), video.stride(0), text.stride(0), nat.IN_F32,
                                         p(xall[r * pl.operand_bytes:]), p(inv), p(diag), stream))
    xr = xall[:pl.operand_bytes]
    part = torch.empty(pl.fwd_ws_floats, **f32)
    gbuf = torch.zeros(pl.gbuf_bytes // 4, **f32)
    rz, wrz = torch.rand(world, 2 * pl.bpad, **f32) * 1e-4, torch.rand(world, 2 * pl.bpad, **f32) * 1e-4
    T, w = float(temperature), float(negative_w)
    two_pass = bool(lib.crossclr_needs_row_shift(T, w))
    stages = {}
    if two_pass:
        shift = torch.empty(world, 2 * pl.bpad, **f32)
        for r in range(world):   # every rank's row maxima over all columns ("gathered")
            xs = xall[r * pl.operand_bytes:(r + 1) * pl.operand_bytes]
            nat.check(lib.crossclr_forward_rowmax(ctypes.byref(plans[r]), p(xs), p(xall), world, 0, -1, T, w, None, p(part), p(shift[r]), 0, stream))
        st = torch.empty(lib.crossclr_rect_stash_bytes_s(pp, 1), dtype=torch.uint8, device=dev)
        stages["forward_rect_save"] = lambda: lib.crossclr_forward_rect_save_s(pp, p(xr), p(xall), 1, 1, T, w, None, p(shift[0]), p(shift), p(part),
                                                                               pl.fwd_slots, p(st), stream)
        stages["backward_rect_saved"] = lambda: lib.crossclr_backward_rect_saved_s(pp, p(xall), p(st), 1, 1, T, w, p(rz[0]), p(wrz[0]), p(rz), p(wrz),
                                                                                   None, p(gbuf), 1, stream)
        if recompute:
            stages["forward_recompute_path"] = lambda: lib.crossclr_forward_s(pp, p(xr), p(xall), world, 0, 0, T, w, None, p(shift[0]), p(part),
                                                                              pl.fwd_slots, stream)
            stages["backward_recompute"] = lambda: lib.crossclr_backward_s(pp, p(xr), p(xall), world, 0, 0, T, w, p(rz[0]), p(wrz[0]), p(rz), p(wrz),
                                                                           None, p(shift[0]), p(shift), p(gbuf), 1, stream)
    else:
        st = torch.empty(lib.crossclr_rect_stash_bytes(pp, 1), dtype=torch.uint8, device=dev)
        stages["forward_rect_save"] = lambda: lib.crossclr_forward_rect_save(pp, p(xr), p(xall), 1, 1, 0, T, w, None, p(part), pl.fwd_slots, None,
                                                                             p(st), stream)
        stages["backward_rect_saved"] = lambda: lib.crossclr_backward_rect_saved(pp, p(xall), p(st), 1, 1, T, w, p(rz[0]), p(wrz[0]), p(rz), p(wrz),
                                                                                 None, p(gbuf), 1, stream)
        if recompute:
            stages["forward_recompute_path"] = lambda: lib.crossclr_forward(pp, p(xr), p(xall), world, 0, 0, T, w, p(part), pl.fwd_slots, stream)
            stages["backward_recompute"] = lambda: lib.crossclr_backward(pp, p(xr), p(xall), world, 0, 0, T, w, p(rz[0]), p(wrz[0]), p(rz), p(wrz),
                                                                         p(gbuf), 1, stream)
    if st.numel() == 0:
        raise RuntimeError("remote_block_times: this plan has no saved path for remote blocks")
    out: Dict[str, float] = {"two_pass": float(two_pass), "stash_bytes": float(st.numel())}
    for name, fn in stages.items():
        for _ in range(warmup):
            nat.check(fn())
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a, z in ev:
            a.record()
            nat.check(fn())
            z.record()
        torch.cuda.synchronize(dev)
        out[name] = sorted(a.elapsed_time(z) for a, z in ev)[iters // 2]
    return out
