"""Producer-side fusion: the projection heads in front of the criterion.

`/root/reference/README.md:24-38` feeds `CrossCLR_onlyIntraModality` "features: [bsz, f_dim]" that the model's last linear layers
produced.  `ProjectedCrossCLR` owns those two layers and evaluates

    loss = criterion(F.linear(x_video, Wv, bv), F.linear(x_text, Wt, bt))

with the projection, the L2-normalisation of `trainer/loss.py:79-80` and the packing of the kernels' operand fused into ONE launch
(`crossclr_project_pack`): the projected features never travel to HBM and back, and the separate normalize launch disappears.
Backward: the loss kernels return the gradient w.r.t. the unit rows; `crossclr_project_backward_prep` applies the
normalise-backward; the projection's own gradients (dW = g^T x, dx = g W, db = column sums) are plain library GEMMs.
bf16 operands with fp32 accumulation (the BASELINE headline mode), embed_dim <= 512.
"""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from . import _native as nat
from . import loss as L


def _weights_bf16(w: torch.Tensor) -> "tuple[torch.Tensor, int]":
    """nn.Linear weight [D, Din] -> bf16 [D, ldw], columns zero-padded to a multiple of 64 (the kernel's K chunk)."""
    D, Din = w.shape
    ldw = (Din + 63) // 64 * 64
    out = torch.zeros(D, ldw, dtype=torch.bfloat16, device=w.device)
    out[:, :Din] = w.detach()
    return out, ldw


class _ProjectedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xv, xt, wv, bv, wt, bt, temperature, negative_w, group):
        xv_c, xt_c = L._row_major(xv.detach()), L._row_major(xt.detach())
        needs = any(ctx.needs_input_grad[:6])
        wvb, ldw = _weights_bf16(wv)
        wtb, ldw_t = _weights_bf16(wt)
        bvf = None if bv is None else bv.detach().float().contiguous()
        btf = None if bt is None else bt.detach().float().contiguous()
        with L._device_of(xv_c):
            loss, ws = L._forward_impl(xv_c, xt_c, temperature, negative_w, "bf16", group, None, None, save_for_backward=needs,
                                       project=(wvb, wtb, (ldw, ldw_t), bvf, btf, wv.shape[0]))
        ctx.ws = ws
        ctx.save_for_backward(xv_c, xt_c, wv.detach(), wt.detach())
        ctx.has_bias = (bv is not None, bt is not None)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        L._refuse_double_backward("ProjectedCrossCLR")
        xv, xt, wv, wt = ctx.saved_tensors
        ws, lib = ctx.ws, nat.library()
        plan = ws.plan
        b, D, bpad, Dpad = plan.b, plan.D, plan.bpad, plan.Dpad
        dev = xv.device
        with L._device_of(xv):
            # the unit rows in fp32 (the positive-pair term of the gradient is the PARTNER's unit row)
            packed = ws.xhat.view(torch.bfloat16).view(2, bpad, Dpad)
            yv, yt = packed[0, :b, :D].float(), packed[1, :b, :D].float()
            ws.in_dtype = nat.IN_F32
            # the finish kernel's "rows as given" are these unit rows: its inv_norm array must read all ones (as crossclr_pack
            # would have left it); the real 1 / ||y|| goes into the normalise-backward below
            inv_norm, ws.inv_norm = ws.inv_norm, torch.ones_like(ws.inv_norm)
            gv, gt = L._backward_impl(ws, yv, yt, grad_out)        # d(loss) / d(unit rows)   (ws.prenormalized is set)
            ws.inv_norm = inv_norm
            gyv, gyt = torch.empty_like(gv), torch.empty_like(gt)
            nat.check(lib.crossclr_project_backward_prep(ctypes.byref(plan), L._ptr(gv), L._ptr(gt), gv.stride(0), gt.stride(0),
                                                         L._ptr(ws.xhat), L._ptr(ws.inv_norm), L._ptr(gyv), L._ptr(gyt), gyv.stride(0),
                                                         L._stream_for(xv)))
            need = ctx.needs_input_grad
            xvf, xtf = xv.float(), xt.float()
            dxv = (gyv @ wv.float()).to(xv.dtype) if need[0] else None
            dxt = (gyt @ wt.float()).to(xt.dtype) if need[1] else None
            dwv = (gyv.t() @ xvf).to(wv.dtype) if need[2] else None
            dwt = (gyt.t() @ xtf).to(wt.dtype) if need[4] else None
            dbv = gyv.sum(0) if (need[3] and ctx.has_bias[0]) else None
            dbt = gyt.sum(0) if (need[5] and ctx.has_bias[1]) else None
        return dxv, dxt, dwv, dbv, dwt, dbt, None, None, None


def projected_crossclr_loss(x_video: torch.Tensor, x_text: torch.Tensor, w_video: torch.Tensor, b_video, w_text: torch.Tensor, b_text,
                            temperature: float = 0.03, negative_weight: float = 0.8, *, process_group=None) -> torch.Tensor:
    """`CrossCLR_onlyIntraModality(temperature, negative_weight)(F.linear(x_video, w_video, b_video), F.linear(x_text, w_text, b_text))`
    with the projection fused into the loss' first kernel.  Weights in `torch.nn.Linear` layout ([embed_dim, in_dim])."""
    if x_video.dim() != 2 or x_text.dim() != 2 or x_video.shape[0] != x_text.shape[0]:
        raise RuntimeError(f"expected two [batch, in_dim] inputs with the same batch size, got {tuple(x_video.shape)} and {tuple(x_text.shape)}")
    if w_video.dim() != 2 or w_text.dim() != 2 or w_video.shape[0] != w_text.shape[0]:
        raise RuntimeError("the two projections must map to the same embed_dim")
    if w_video.shape[1] != x_video.shape[1] or w_text.shape[1] != x_text.shape[1]:
        raise RuntimeError("weight / input in_dim mismatch")
    if x_video.dtype != x_text.dtype or x_video.dtype not in L._IN_DTYPE:
        raise RuntimeError(f"unsupported / mismatched input dtypes {x_video.dtype}, {x_text.dtype}")
    if not x_video.is_cuda and nat.backend() != "emu-host":
        raise RuntimeError("CrossCLR HIP path needs inputs on the GPU (got a CPU tensor); there is no CPU fallback")
    return _ProjectedFunction.apply(x_video, x_text, w_video, b_video, w_text, b_text, float(temperature), float(negative_weight),
                                    process_group)


class ProjectedCrossCLR(nn.Module):
    """Two linear projection heads + the CrossCLR criterion of `trainer/loss.py:44-114`, evaluated as one fused pipeline."""

    def __init__(self, in_dim_video: int, in_dim_text: int, embed_dim: int, temperature=0.03, negative_weight=0.8, bias: bool = True,
                 *, process_group=None):
        super().__init__()
        self.video_proj = nn.Linear(in_dim_video, embed_dim, bias=bias)
        self.text_proj = nn.Linear(in_dim_text, embed_dim, bias=bias)
        self.temperature = temperature
        self.negative_w = negative_weight
        self.process_group = process_group

    def forward(self, x_video, x_text):
        return projected_crossclr_loss(x_video, x_text, self.video_proj.weight, self.video_proj.bias, self.text_proj.weight,
                                       self.text_proj.bias, self.temperature, self.negative_w, process_group=self.process_group)
