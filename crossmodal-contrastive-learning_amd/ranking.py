"""Score statistics of the video x text similarity matrix: the reference's max-margin ranking loss and retrieval ranks.

Mirrors `trainer/loss.py:7-41` of amazon-science/crossmodal-contrastive-learning:

    criterion = MaxMargin_coot(use_cuda=True, margin=0.1)
    loss = criterion(im, s)          # [B, D], [B, D] -> 0-dim; scores = im @ s.T, no normalisation (loss.py:7-15, 30)

The reference class cannot be constructed (`super(ContrastiveLoss_coot, self)` at loss.py:24 names an undefined class); its
`forward` (loss.py:29-41) is what is implemented, with the constructor arguments and attributes the reference declares
(`use_cuda`, `margin`, `sim`).  The B x B score matrix, its two hinge matrices and the eye mask are never materialised: two
tiled passes over the inter-modal block through the C-ABI (`crossclr_score_diag`, `crossclr_score_rows`) for the forward, the
generic tiled backward with indicator weights (`crossclr_maxmargin_backward[_finish]`) for the gradients.

`retrieval_ranks` is the evaluation step that follows training in the CrossCLR / COOT pipelines (not in the reference
repository): rank of each sample's partner among all candidates of the other modality, both directions, from the same kernels
with margin 0 (count of candidates scoring strictly higher than the partner).
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict

import torch
from torch import nn

from . import _native as nat
from .loss import _IN_DTYPE, _device_of, _plan_for, _ptr, _refuse_double_backward, _resolve_mode, _row_major, _stream_for, _validate


def cosine_sim(emb1: torch.Tensor, emb2: torch.Tensor) -> torch.Tensor:
    """`trainer/loss.py:7-15`: the plain product emb1 @ emb2^T (a cosine only when the caller normalised the rows)."""
    return emb1.mm(emb2.t())


class _Scores:
    __slots__ = ("plan", "xhat", "ones", "diag", "hinge", "active", "loss_sum", "in_dtype", "margin", "mask")


def _score_rows(im: torch.Tensor, s: torch.Tensor, margin: float, compute_mode: str, normalize: bool, save_mask: bool = False) -> _Scores:
    lib = nat.library()
    b, D = im.shape
    dev = im.device
    mode = _resolve_mode(compute_mode, b, im.dtype)
    plan = _plan_for(b, D, 1, 0, mode)
    pp = ctypes.byref(plan)
    stream = _stream_for(im)
    f32 = dict(dtype=torch.float32, device=dev)
    sc = _Scores()
    sc.plan, sc.in_dtype, sc.margin = plan, _IN_DTYPE[im.dtype], float(margin)
    sc.xhat = torch.empty(plan.operand_bytes, dtype=torch.uint8, device=dev)
    sc.ones = torch.empty(2 * plan.bpad, **f32)          # inv_norm: all ones when the rows are used as given
    pair_cos = torch.empty(plan.bpad, **f32)
    lay_out = lib.crossclr_normalize if normalize else lib.crossclr_pack
    nat.check(lay_out(pp, _ptr(im), _ptr(s), im.stride(0), s.stride(0), sc.in_dtype, _ptr(sc.xhat), _ptr(sc.ones), _ptr(pair_cos), stream))
    sc.diag = torch.empty(2 * plan.bpad, **f32)
    nat.check(lib.crossclr_score_diag(pp, _ptr(sc.xhat), _ptr(sc.diag), stream))
    part = torch.empty(plan.fwd_ws_floats, **f32)
    sc.hinge, sc.active = torch.empty(2 * plan.bpad, **f32), torch.empty(2 * plan.bpad, **f32)
    sc.loss_sum = torch.empty(plan.loss_ws_doubles, dtype=torch.float64, device=dev)
    # With a backward to follow the same pass also leaves every pair's number of active hinges (one byte per pair, 64 MiB at B = 8192: the
    # reference's autograd keeps two B x B float hinge matrices for this): the backward is then one product with that mask instead of a
    # second evaluation of the scores.  CROSSCLR_MAXMARGIN_SAVE=0, or a library that declines (CROSSCLR_DISABLE_SYMMETRIC): the recomputing pair.
    sc.mask = None
    if save_mask and os.environ.get("CROSSCLR_MAXMARGIN_SAVE", "1") != "0":
        mask = torch.empty(lib.crossclr_maxmargin_mask_bytes(pp), dtype=torch.uint8, device=dev)
        if lib.crossclr_score_rows_save(pp, _ptr(sc.xhat), _ptr(sc.diag), sc.margin, _ptr(part), _ptr(sc.hinge), _ptr(sc.active),
                                        _ptr(sc.loss_sum), _ptr(mask), stream) == 0:
            sc.mask = mask
            return sc
    nat.check(lib.crossclr_score_rows(pp, _ptr(sc.xhat), _ptr(sc.diag), sc.margin, _ptr(part), _ptr(sc.hinge), _ptr(sc.active),
                                      _ptr(sc.loss_sum), stream))
    return sc


class _MaxMarginFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, s, margin, compute_mode):
        im_c, s_c = _row_major(im.detach()), _row_major(s.detach())
        with _device_of(im_c):
            sc = _score_rows(im_c, s_c, margin, compute_mode, normalize=False, save_mask=any(ctx.needs_input_grad[:2]))
        ctx.sc = sc
        ctx.save_for_backward(im_c, s_c)
        # (cost_s.sum() + cost_im.sum()).div(B * B) (loss.py:41), in the input dtype like the reference's eager ops
        return sc.loss_sum[1].to(im.dtype if im.dtype != torch.float64 else torch.float64)

    @staticmethod
    def backward(ctx, grad_out):
        _refuse_double_backward("MaxMargin_coot")
        im_c, s_c = ctx.saved_tensors
        sc, lib = ctx.sc, nat.library()
        plan = sc.plan
        pp = ctypes.byref(plan)
        dev = im_c.device
        with _device_of(im_c):
            stream = _stream_for(im_c)
            gbuf = torch.empty(plan.gbuf_bytes // 4, dtype=torch.float32, device=dev)
            if sc.mask is not None:
                nat.check(lib.crossclr_maxmargin_backward_saved(pp, _ptr(sc.xhat), _ptr(sc.mask), _ptr(gbuf), stream))
            else:
                nat.check(lib.crossclr_maxmargin_backward(pp, _ptr(sc.xhat), _ptr(sc.diag), sc.margin, _ptr(gbuf), stream))
            go = grad_out.detach().to(device=dev, dtype=torch.float64).reshape(1).contiguous()
            g_im, g_s = torch.empty_like(im_c), torch.empty_like(s_c)
            nat.check(lib.crossclr_maxmargin_backward_finish(pp, _ptr(gbuf), _ptr(im_c), _ptr(s_c), im_c.stride(0), s_c.stride(0),
                                                             sc.in_dtype, _ptr(sc.ones), _ptr(sc.active), _ptr(go), _ptr(g_im), _ptr(g_s),
                                                             g_im.stride(0), g_s.stride(0), stream))
        return (g_im if ctx.needs_input_grad[0] else None, g_s if ctx.needs_input_grad[1] else None, None, None)


def _check(im: torch.Tensor, s: torch.Tensor) -> None:
    _validate(im, s)
    if not im.is_cuda and nat.backend() != "emu-host":
        raise RuntimeError("the HIP path needs inputs on the GPU (got a CPU tensor); there is no CPU fallback")
    if im.shape[0] == 0:
        raise RuntimeError("empty batch")


def max_margin_loss(im: torch.Tensor, s: torch.Tensor, margin: float = 0.1, *, compute_mode: str = "fp32") -> torch.Tensor:
    """Functional form of `MaxMargin_coot.forward` (trainer/loss.py:29-41).  compute_mode="fp32" (default) = exact-fp32 products like the
    reference's `mm`; "bf16" is opt-in: the scores are products of the caller's UN-normalised rows, so there is no a-priori error bar."""
    _check(im, s)
    return _MaxMarginFunction.apply(im, s, float(margin), compute_mode)


class MaxMargin_coot(nn.Module):
    """Regular contrastive (max-margin ranking) loss between two groups of embeddings, inputs [batch, embed_dim]
    (`trainer/loss.py:17-41`; COOT, NeurIPS 2020).  Constructor arguments and attributes as the reference declares them
    (loss.py:23-27); `use_cuda` is kept for signature compatibility -- the inputs must be on the GPU either way."""

    def __init__(self, use_cuda: bool = True, margin: float = 0.1, *, compute_mode: str = "fp32"):
        super().__init__()
        self.margin = margin
        self.sim = cosine_sim
        self.use_cuda = use_cuda
        self.compute_mode = compute_mode

    def forward(self, im, s):
        return max_margin_loss(im, s, self.margin, compute_mode=self.compute_mode)

    def extra_repr(self):
        return f"margin={self.margin}, compute_mode={self.compute_mode!r}"


def retrieval_ranks(video_features: torch.Tensor, text_features: torch.Tensor, *, normalize: bool = True,
                    compute_mode: str = "fp32") -> Dict[str, torch.Tensor]:
    """Ranks of the partners in both retrieval directions over the B x B cosine matrix (normalize=False: plain products).

    Returns device tensors (nothing synchronises the host): `v2t_ranks[i]` = number of texts scoring strictly higher than text i
    for video i (0 = retrieved first), `t2v_ranks[j]` likewise; `v2t` / `t2v` = [R@1, R@5, R@10, median rank, mean rank] with
    ranks counted from 1 as in the CrossCLR / COOT evaluation tables.  compute_mode="fp32" (default) = exact-fp32 products."""
    _check(video_features, text_features)
    v, t = _row_major(video_features.detach()), _row_major(text_features.detach())
    with _device_of(v):
        sc = _score_rows(v, t, 0.0, compute_mode, normalize=normalize)
    b, bpad = sc.plan.b, sc.plan.bpad
    out = {"v2t_ranks": sc.active[:b].to(torch.int64), "t2v_ranks": sc.active[bpad:bpad + b].to(torch.int64)}
    for key in ("v2t", "t2v"):
        r = out[key + "_ranks"].to(torch.float64)
        out[key] = torch.stack([(r < 1).double().mean(), (r < 5).double().mean(), (r < 10).double().mean(),
                                r.median() + 1.0, r.mean() + 1.0])
    return out
