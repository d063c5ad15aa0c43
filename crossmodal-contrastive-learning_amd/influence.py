"""Influential-sample pruning and loss weighting on top of the fused CrossCLR kernels
(BASELINE config 5).

NOT part of the reference @ v1: `/root/reference/trainer/loss.py` has only the scalar `negative_weight`
(`:56,99-100`) and its README names the release "onlyIntraModality" (`README.md:19`).  The recipe
restates section 3.2/3.3 of the CrossCLR paper (ICCV 2021) -- see `oracle/influence_oracle.py` for the
dense statement the tests check this file and the kernels against; parity with the authors' code
is unpinned.

    conn[i] = mean_j xhat_i . xhat_j   (input-space features, self pair masked to 0)
            = (xhat_i . sum_j xhat_j - xhat_i . xhat_i) / B          -- O(B D), no B x B product
    keep[i] = conn[i] / max(conn) < score_threshold    (highly connected samples leave the negative set)
    omega   = B * rho / sum(rho),  rho = exp(conn / sum(conn) / temperature_weights)

The O(B D) statistics run in three small HBM-bound HIP kernels behind the C-ABI (`crossclr_influence_*` in
include/crossclr.h); when sharded, one [2, D] all-reduce and one [2, b] all-gather sit between them.  The
B x B work -- the loss with `negative_scale = keep` and `loss_weight = omega` -- runs in the weighted fused
kernels (`crossclr_*_w`).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _native as nat
from .loss import _IN_DTYPE, CrossCLR_onlyIntraModality, _device_of, _ptr, _row_major, _stream_for, crossclr_loss


class PackedPair(tuple):
    """(video[b], text[b]) views of one float32 [2][bpad] array in the kernels' statistics layout; `crossclr_loss`
    uses `.packed` directly instead of re-packing the two halves."""
    def __new__(cls, packed: torch.Tensor, b: int, bpad: int):
        self = super().__new__(cls, (packed[:b], packed[bpad:bpad + b]))
        self.packed, self.b = packed, b
        return self


def influential_sample_weights(input_vid: torch.Tensor, input_txt: torch.Tensor, score_threshold: float = 0.7,
                               temperature_weights: float = 0.0035, process_group=None) -> Tuple[PackedPair, PackedPair]:
    """(negative_scale, loss_weight) for `crossclr_loss` from the input-space features of the LOCAL rows
    (`process_group`: the statistics are those of the concatenated global batch)."""
    if input_vid.dim() != 2 or input_txt.dim() != 2 or input_vid.shape != input_txt.shape:
        raise RuntimeError("input_vid / input_txt must be [batch, features] tensors of the same shape")
    if input_vid.dtype != input_txt.dtype or input_vid.dtype not in _IN_DTYPE or input_vid.device != input_txt.device:
        raise RuntimeError("input_vid / input_txt must share one floating dtype and device")
    if not input_vid.is_cuda and nat.backend() != "emu-host":
        raise RuntimeError("influential_sample_weights needs inputs on the GPU; there is no CPU fallback")
    lib = nat.library()
    xv, xt = _row_major(input_vid.detach()), _row_major(input_txt.detach())
    with _device_of(xv):
        return _influence_impl(lib, xv, xt, score_threshold, temperature_weights, process_group)


def _influence_impl(lib, xv, xt, score_threshold, temperature_weights, process_group):
    import torch.distributed as dist
    b, din = xv.shape
    dev = xv.device
    world = dist.get_world_size(process_group) if process_group is not None else 1
    rank = dist.get_rank(process_group) if process_group is not None else 0
    plan = nat.make_plan(b, 1, world, rank, nat.MODE_FP32)   # only b / bpad / world / rank are used
    stream = _stream_for(xv)
    dt = _IN_DTYPE[xv.dtype]
    inv_norm = torch.empty(2 * b, dtype=torch.float32, device=dev)
    partial = torch.empty(2 * nat.INFL_BLOCKS * din, dtype=torch.float32, device=dev)
    colsum = torch.empty(2 * din, dtype=torch.float64, device=dev)
    conn = torch.empty(2 * b, dtype=torch.float64, device=dev)
    nat.check(lib.crossclr_influence_colsum(_ptr(xv), _ptr(xt), xv.stride(0), xt.stride(0), dt, b, din, _ptr(inv_norm),
                                            _ptr(partial), _ptr(colsum), stream))
    if world > 1:
        dist.all_reduce(colsum, group=process_group)
    nat.check(lib.crossclr_influence_conn(_ptr(xv), _ptr(xt), xv.stride(0), xt.stride(0), dt, b, din, _ptr(inv_norm),
                                          _ptr(colsum), b * world, _ptr(conn), stream))
    conn_all = conn
    if world > 1:
        conn_all = torch.empty(world * 2 * b, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(conn_all, conn, group=process_group)
    neg_scale = torch.empty(2 * plan.bpad, dtype=torch.float32, device=dev)
    loss_weight = torch.empty(2 * plan.bpad, dtype=torch.float32, device=dev)
    nat.check(lib.crossclr_influence_finish(ctypes.byref(plan), _ptr(conn_all), float(score_threshold),
                                            float(temperature_weights), _ptr(neg_scale), _ptr(loss_weight), stream))
    return PackedPair(neg_scale, b, plan.bpad), PackedPair(loss_weight, b, plan.bpad)


class CrossCLR(CrossCLR_onlyIntraModality):
    """CrossCLR with influential-sample pruning and weighting.  `forward(video_features, text_features)` alone is
    exactly `CrossCLR_onlyIntraModality`; passing the input-space features switches the weighting on."""

    def __init__(self, temperature=0.03, temperature_weights=0.0035, negative_weight=0.8, score_threshold=0.7,
                 logger=None, *, compute_mode: str = "auto", process_group=None):
        super().__init__(temperature, negative_weight, logger, compute_mode=compute_mode, process_group=process_group)
        self.temperature_weights = temperature_weights
        self.score_threshold = score_threshold

    def forward(self, video_features, text_features, input_vid: Optional[torch.Tensor] = None,
                input_txt: Optional[torch.Tensor] = None):
        if (input_vid is None) != (input_txt is None):
            raise RuntimeError("pass both input_vid and input_txt, or neither")
        if input_vid is None:
            return super().forward(video_features, text_features)
        if input_vid.shape[0] != video_features.shape[0]:
            raise RuntimeError("input_vid / input_txt must have one row per embedding row")
        scale, weight = influential_sample_weights(input_vid, input_txt, self.score_threshold, self.temperature_weights,
                                                   self.process_group)
        return crossclr_loss(video_features, text_features, self.temperature, self.negative_w,
                             compute_mode=self.compute_mode, process_group=self.process_group,
                             negative_scale=scale, loss_weight=weight)
