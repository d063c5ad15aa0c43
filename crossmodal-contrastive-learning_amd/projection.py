"""Producer-side fusion: the projection heads in front of the criterion.

`/root/reference/README.md:24-38` feeds `CrossCLR_onlyIntraModality` "features: [bsz, f_dim]" that the model's last linear layers
produced.  `ProjectedCrossCLR` owns those two layers and evaluates

    loss = criterion(F.linear(x_video, Wv, bv), F.linear(x_text, Wt, bt))

with the projection, the L2-normalisation of `trainer/loss.py:79-80` and the packing of the kernels' operand fused into ONE launch
(`crossclr_project_pack`): the projected features never travel to HBM and back, and the separate normalize launch disappears.
Backward: the loss' finish kernel reads the packed bf16 unit rows and 1 / ||y|| and writes g_y = d(loss)/d(projected features) in bf16
(normalise-backward and positive-pair term fused: `crossclr_backward_finish_p(prenormalized = 2)`); the projection's own gradients
(dW = g_y^T x, dx = g_y W, db = column sums) are plain library GEMMs with bf16 operands and fp32 accumulation (hipBLASLt), against the
bf16 weights the forward multiplied with (cast every step: no cache to go stale; dx leaves the GEMM in fp32 for fp32 inputs).
bf16 operands with fp32 accumulation (the BASELINE headline mode), embed_dim <= 1024 (64 rows per block up to 512, 32 rows above).
"""
from __future__ import annotations

import torch
from torch import nn

from . import _native as nat
from . import loss as L


def _weights_bf16(w: torch.Tensor, Dpad: int = 0, row_major: "torch.Tensor | None" = None) -> "tuple[torch.Tensor, int]":
    """nn.Linear weight [D, Din] as the kernels want it, columns zero-padded to ldw = a multiple of 64 (the kernel's K chunk).
    Dpad = 0: bf16 [D, ldw] row-major (the backward's dx = g_y W GEMM; `crossclr_project_pack`).
    Dpad > 0: FRAGMENT-MAJOR for `crossclr_project_pack_wf` -- [Dpad / 32][ldw / 16][64][8], lane (l31, half) of record (d32, ks) holding
    W[32 d32 + l31][16 ks + 8 half .. + 7], rows beyond D zero: a wave's MFMA B fragment is one coalesced 1-KiB load
    (`row_major`: the Dpad = 0 form of the same weight, if the caller already has it -- saves the second cast).
    Cast on EVERY call (two small launches per weight and step): a cache keyed on the parameter's identity / `_version` cannot see updates
    through `.data` (fused optimisers, EMA, clipping) nor tell a new parameter from a freed one whose id and address were reused, and a
    stale copy would train silently on old weights (round-4 review)."""
    D, Din = w.shape
    ldw = (Din + 63) // 64 * 64
    if Dpad > 0:
        if row_major is not None and Dpad == D:
            pad = row_major
        elif Dpad != D or ldw != Din:
            pad = torch.zeros(Dpad, ldw, dtype=torch.bfloat16, device=w.device)
            pad[:D, :Din].copy_(w.detach())
        else:
            pad = w.detach().to(torch.bfloat16)
        out = pad.view(Dpad // 32, 32, ldw // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
    elif ldw == Din:
        out = w.detach().to(torch.bfloat16).contiguous()
    else:
        out = torch.zeros(D, ldw, dtype=torch.bfloat16, device=w.device)
        out[:, :Din].copy_(w.detach())
    return out, ldw


def _mm_f32_out(a_bf16: torch.Tensor, b_bf16: torch.Tensor, out_dtype) -> torch.Tensor:
    """a @ b with bf16 operands and fp32 accumulation, result in `out_dtype` WITHOUT passing through bf16 when it is wider (the library's
    bf16 GEMM with an fp32 output on the device; on the host emulation's CPU tensors: an fp32 product of the same bf16 values)."""
    if out_dtype == torch.bfloat16:
        return torch.mm(a_bf16, b_bf16)
    if a_bf16.is_cuda:
        return torch.mm(a_bf16, b_bf16, out_dtype=torch.float32).to(out_dtype)
    return torch.mm(a_bf16.float(), b_bf16.float()).to(out_dtype)


class _ProjectedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xv, xt, wv, bv, wt, bt, temperature, negative_w, group):
        xv_c, xt_c = L._row_major(xv.detach()), L._row_major(xt.detach())
        needs = any(ctx.needs_input_grad[:6])
        Dpad = L._plan_for(xv.shape[0], wv.shape[0], 1, 0, nat.MODE_BF16).Dpad
        # row-major bf16 copies (the backward's dx GEMM multiplies with the values the forward multiplied with: handed over through ctx,
        # not looked up again), and the fragment-major form of the same values for crossclr_project_pack_wf
        wv_rm, ldw = _weights_bf16(wv)
        wt_rm, ldw_t = _weights_bf16(wt)
        wvb, _ = _weights_bf16(wv, Dpad, wv_rm)
        wtb, _ = _weights_bf16(wt, Dpad, wt_rm)
        bvf = None if bv is None else bv.detach().float().contiguous()
        btf = None if bt is None else bt.detach().float().contiguous()
        with L._device_of(xv_c):
            loss, ws = L._forward_impl(xv_c, xt_c, temperature, negative_w, "bf16", group, None, None, save_for_backward=needs,
                                       project=(wvb, wtb, (ldw, ldw_t), bvf, btf, wv.shape[0]))
        ctx.ws = ws
        ctx.save_for_backward(xv_c, xt_c, wv_rm, wt_rm)
        ctx.w_meta = (tuple(wv.shape), wv.dtype, tuple(wt.shape), wt.dtype)
        ctx.bias_dtype = (None if bv is None else bv.dtype, None if bt is None else bt.dtype)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        L._refuse_double_backward("ProjectedCrossCLR")
        xv, xt, wv_rm, wt_rm = ctx.saved_tensors
        (wv_shape, wv_dtype, wt_shape, wt_dtype) = ctx.w_meta
        ws = ctx.ws
        plan = ws.plan
        b, D, bpad, Dpad = plan.b, plan.D, plan.bpad, plan.Dpad
        with L._device_of(xv):
            # The finish kernel reads the packed bf16 unit rows themselves -- own row for the normalise-backward, partner's row for the
            # positive-pair term -- with 1 / ||y|| from the forward, and writes g_y = d(loss)/d(projected features) in bf16
            # (prenormalized = 2): no fp32 copies of the unit rows, no separate normalise-backward pass.
            packed = ws.xhat.view(torch.bfloat16).view(2, bpad, Dpad)
            ws.in_dtype = nat.IN_BF16
            ws.prenormalized = 2
            gyv, gyt = L._backward_impl(ws, packed[0, :b, :D], packed[1, :b, :D], grad_out)
            # the projection's own gradients: bf16 operands, fp32 accumulation (plain library GEMMs -- hipBLASLt)
            need = ctx.needs_input_grad
            dxv = _mm_f32_out(gyv, wv_rm[:, :wv_shape[1]], xv.dtype) if need[0] else None     # (the bf16 values the forward multiplied with)
            dxt = _mm_f32_out(gyt, wt_rm[:, :wt_shape[1]], xt.dtype) if need[1] else None
            dwv = dwt = dbv = dbt = None
            if need[2] or need[4] or need[3] or need[5]:
                # dW = g_y^T x and db = column sums of g_y, both modalities: ONE split-K MFMA launch + a reduce (crossclr_project_dw) --
                # hipBLASLt's pick for this shape (K = batch = 8192, 128 output tiles, no split-K) takes 59 us per modality, this ~1/4
                lib = nat.library()
                dinv, dint = wv_shape[1], wt_shape[1]
                wsf = torch.empty(lib.crossclr_project_dw_ws_floats(b, D, dinv, dint), dtype=torch.float32, device=xv.device)
                dwv32 = torch.empty(D, dinv, dtype=torch.float32, device=xv.device)
                dwt32 = torch.empty(D, dint, dtype=torch.float32, device=xv.device)
                dbv32 = torch.empty(D, dtype=torch.float32, device=xv.device) if ctx.bias_dtype[0] is not None else None
                dbt32 = torch.empty(D, dtype=torch.float32, device=xv.device) if ctx.bias_dtype[1] is not None else None
                nat.check(lib.crossclr_project_dw(b, D, L._ptr(gyv), L._ptr(gyt), gyv.stride(0), L._ptr(xv), L._ptr(xt), xv.stride(0), xt.stride(0),
                                                  dinv, dint, L._IN_DTYPE[xv.dtype], L._ptr(wsf), L._ptr(dwv32), L._ptr(dwt32), dinv, dint,
                                                  L._ptr(dbv32), L._ptr(dbt32), L._stream_for(xv)))
                dwv = dwv32.to(wv_dtype) if need[2] else None
                dwt = dwt32.to(wt_dtype) if need[4] else None
                dbv = dbv32.to(ctx.bias_dtype[0]) if (need[3] and dbv32 is not None) else None
                dbt = dbt32.to(ctx.bias_dtype[1]) if (need[5] and dbt32 is not None) else None
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
