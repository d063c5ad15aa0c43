"""CPU oracle for the CrossCLR "only intra-modality" contrastive loss hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`crossmodal-contrastive-learning_amd/`) imports this file.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, and only as
the checker / the timed CPU baseline -- never as the thing shipped.

What it restates: `/root/reference/trainer/loss.py:44-114`
(`CrossCLR_onlyIntraModality.forward`) plus the autograd backward that PyTorch
derives from it.  The reference has no tests of its own, so
parity is pinned by golden vectors generated in the build container by importing
the reference itself (`tests/golden/make_golden.py`); `tests/test_oracle.py`
checks every function here against those vectors.

Three forms:

* `eager_loss`      -- op-for-op: the same ATen op sequence as the reference,
                       including the NumPy float64 identity masks that promote
                       everything after the masking step to float64.  Verified
                       bit-identical (loss and both gradients) to the reference
                       import in the build container.  This is the function
                       `bench.py` times as the CPU baseline ("port").
* `streaming_stats` / `streaming_loss_and_grads`
                    -- closed form in float64, blocked over rows so nothing
                       O(B^2) is ever resident; usable at B=65536.  Also returns
                       the per-row intermediates (logZ, diagonal logit) that the
                       HIP kernels expose, so a failing parity test can say
                       which stage is off.
* `sharded_loss_and_grads`
                    -- the multi-GPU semantics: rank r owns rows [r*b,(r+1)*b);
                       result must equal the single-process value on the
                       concatenated batch.  Used by the gloo tests.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

NORM_EPS = 1e-12  # F.normalize default eps, reference loss.py:79-80


# --------------------------------------------------------------------------- #
# (a) op-for-op eager form                                                      #
# --------------------------------------------------------------------------- #
def _off_diagonal_mask(n: int) -> torch.Tensor:
    """float64 (1 - I_n), built through NumPy exactly like loss.py:62-66."""
    return 1 - torch.from_numpy(np.eye(n))


def _neg_log_masked_softmax(logits: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """loss.py:59-60: -log(sum_j softmax(logits)[i, j] * mask[i, j])."""
    return -torch.log((F.softmax(logits, dim=1) * mask).sum(1))


def eager_loss(video: torch.Tensor, text: torch.Tensor, temperature: float = 0.03,
               negative_weight: float = 0.8) -> torch.Tensor:
    """Same op sequence as loss.py:76-114 (device placement calls dropped).

    Returns the 0-dim float64 loss with an autograd graph attached when the
    inputs require grad, exactly like the reference on a CPU tensor.
    """
    n = video.shape[0]                                            # :76
    vhat = F.normalize(video, dim=1)                              # :79
    that = F.normalize(text, dim=1)                               # :80
    inter_v = vhat @ that.t()                                     # :83
    inter_t = that @ vhat.t()                                     # :84
    intra_v = vhat @ vhat.t()                                     # :87
    intra_t = that @ that.t()                                     # :88
    inter_v /= temperature                                        # :90
    inter_t /= temperature                                        # :91
    intra_v /= temperature                                        # :92
    intra_t /= temperature                                        # :93
    keep = _off_diagonal_mask(vhat.shape[0])                      # :95
    neg_v = intra_v * keep                                        # :96  (-> float64, diag := 0.0)
    neg_t = intra_t * keep                                        # :97
    rows_v = torch.cat([inter_v, negative_weight * neg_v], dim=1)  # :99
    rows_t = torch.cat([inter_t, negative_weight * neg_t], dim=1)  # :100
    eye_v = torch.from_numpy(np.eye(n))                           # :102-103
    eye_t = torch.from_numpy(np.eye(n))                           # :104
    tgt_v = torch.cat([eye_v, torch.zeros_like(neg_v)], dim=1)    # :106,108
    tgt_t = torch.cat([eye_t, torch.zeros_like(neg_t)], dim=1)    # :107,109
    per_v = _neg_log_masked_softmax(rows_v, tgt_v)                # :111
    per_t = _neg_log_masked_softmax(rows_t, tgt_t)                # :112
    return (per_v.mean() + per_t.mean()) / 2                      # :114


def eager_loss_and_grads(video: torch.Tensor, text: torch.Tensor, temperature: float = 0.03,
                         negative_weight: float = 0.8
                         ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    v = video.detach().clone().requires_grad_(True)
    t = text.detach().clone().requires_grad_(True)
    loss = eager_loss(v, t, temperature, negative_weight)
    loss.backward()
    return loss.detach(), v.grad, t.grad


# --------------------------------------------------------------------------- #
# (b) streaming float64 closed form                       #
# --------------------------------------------------------------------------- #
def _unit_rows(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Row L2-normalisation the way loss.py:79 does it, carried out in the
    input dtype and then widened: returns (xhat float64, clamped norm float64)."""
    nrm = x.norm(dim=1, keepdim=True).clamp_min(NORM_EPS)
    return (x / nrm).double(), nrm.double().squeeze(1)


def streaming_stats(video: torch.Tensor, text: torch.Tensor, temperature: float = 0.03,
                    negative_weight: float = 0.8, block: int = 1024,
                    row_range: Optional[Tuple[int, int]] = None) -> Dict[str, torch.Tensor]:
    """Per-row statistics of the loss for rows in `row_range` against ALL columns.

        A[i,j]  = vhat_i . that_j / tau          Cv = vhat vhat^T / tau,  Ct likewise
        Zv[i]   = sum_j exp(A[i,j]) + sum_{j!=i} exp(w Cv[i,j]) + 1       (loss.py:96-100:
        Zt[i]   = sum_j exp(A[j,i]) + sum_{j!=i} exp(w Ct[i,j]) + 1        diag logit is 0, not -inf)
        loss    = mean_i(log Zv - A_ii)/2 + mean_i(log Zt - A_ii)/2       (loss.py:111-114)

    Returns float64 tensors: logZv, logZt, diag (A_ii) for the requested rows,
    `loss_sum` = sum over those rows of (logZv + logZt - 2*diag) and `loss` =
    loss_sum / (2*B) (only meaningful when row_range covers everything).
    """
    B = video.shape[0]
    lo, hi = (0, B) if row_range is None else row_range
    vhat, _ = _unit_rows(video)
    that, _ = _unit_rows(text)
    it = 1.0 / float(temperature)
    w = float(negative_weight)
    logZv = torch.empty(hi - lo, dtype=torch.float64)
    logZt = torch.empty(hi - lo, dtype=torch.float64)
    for r0 in range(lo, hi, block):
        r1 = min(hi, r0 + block)
        idx = torch.arange(r0, r1)
        rows = torch.arange(r1 - r0)
        for own, other, out in ((vhat, that, logZv), (that, vhat, logZt)):
            inter = (own[r0:r1] @ other.t()) * it               # [blk, B]
            intra = (own[r0:r1] @ own.t()) * (it * w)           # [blk, B]
            intra[rows, idx] = 0.0                               # masked diagonal -> logit 0
            out[r0 - lo:r1 - lo] = torch.logsumexp(torch.cat([inter, intra], dim=1), dim=1)
    diag = (vhat[lo:hi] * that[lo:hi]).sum(1) * it
    loss_sum = (logZv + logZt - 2.0 * diag).sum()
    return {"logZv": logZv, "logZt": logZt, "diag": diag, "loss_sum": loss_sum,
            "loss": loss_sum / (2.0 * B)}


def streaming_loss_and_grads(video: torch.Tensor, text: torch.Tensor, temperature: float = 0.03,
                             negative_weight: float = 0.8, block: int = 1024,
                             row_range: Optional[Tuple[int, int]] = None,
                             stats_all: Optional[Dict[str, torch.Tensor]] = None
                             ) -> Dict[str, torch.Tensor]:
    """Loss and d(loss)/d(inputs) in float64 from the closed form of SURVEY.md 3.5.

        GA[i,j] = (exp(A[i,j]) (1/Zv[i] + 1/Zt[j]) - 2 delta_ij) / (2B)
        Sv[i,j] = w exp(w Cv[i,j]) (1/Zv[i] + 1/Zv[j]) / (2B)   (diag 0), St likewise
        dL/dvhat = (GA that + Sv vhat)/tau      dL/dthat = (GA^T vhat + St that)/tau
        dL/dx_i  = (g_i - xhat_i (xhat_i . g_i)) / max(||x_i||, eps)

    Gradients are returned for rows in `row_range` only (exact derivative of the
    GLOBAL loss w.r.t. those rows) -- this is what each rank produces in the
    sharded scheme.  `stats_all` (full-batch logZv/logZt) may be passed in.
    """
    B = video.shape[0]
    lo, hi = (0, B) if row_range is None else row_range
    if stats_all is None:
        stats_all = streaming_stats(video, text, temperature, negative_weight, block)
    vhat, vnorm = _unit_rows(video)
    that, tnorm = _unit_rows(text)
    it = 1.0 / float(temperature)
    w = float(negative_weight)
    lzv, lzt = stats_all["logZv"], stats_all["logZt"]
    gv = torch.zeros(hi - lo, video.shape[1], dtype=torch.float64)
    gt = torch.zeros_like(gv)
    for r0 in range(lo, hi, block):
        r1 = min(hi, r0 + block)
        idx = torch.arange(r0, r1)
        rows = torch.arange(r1 - r0)
        # video rows: inter block A[r,:], intra block Cv[r,:]
        a = (vhat[r0:r1] @ that.t()) * it
        ga = torch.exp(a - lzv[r0:r1, None]) + torch.exp(a - lzt[None, :])
        cv = (vhat[r0:r1] @ vhat.t()) * (it * w)
        sv = torch.exp(cv - lzv[r0:r1, None]) + torch.exp(cv - lzv[None, :])
        sv[rows, idx] = 0.0
        gv[r0 - lo:r1 - lo] = ga @ that + w * (sv @ vhat)
        # text rows: inter block A[:,r]^T, intra block Ct[r,:]
        at = (that[r0:r1] @ vhat.t()) * it
        gat = torch.exp(at - lzt[r0:r1, None]) + torch.exp(at - lzv[None, :])
        ct = (that[r0:r1] @ that.t()) * (it * w)
        st = torch.exp(ct - lzt[r0:r1, None]) + torch.exp(ct - lzt[None, :])
        st[rows, idx] = 0.0
        gt[r0 - lo:r1 - lo] = gat @ vhat + w * (st @ that)
    scale = it / (2.0 * B)
    gv = gv * scale - that[lo:hi] * (it / B)       # analytic -2 delta_ij term
    gt = gt * scale - vhat[lo:hi] * (it / B)

    def _through_normalize(g, xhat, x, nrm):
        # normalize backward; rows with ||x|| < eps were divided by eps (no projection)
        proj = g - xhat * (xhat * g).sum(1, keepdim=True)
        tiny = (x.double().norm(dim=1) < NORM_EPS)[:, None]
        return torch.where(tiny, g, proj) / nrm[:, None]

    grad_v = _through_normalize(gv, vhat[lo:hi], video[lo:hi], vnorm[lo:hi])
    grad_t = _through_normalize(gt, that[lo:hi], text[lo:hi], tnorm[lo:hi])
    return {"loss": stats_all["loss"], "grad_v": grad_v, "grad_t": grad_t,
            "grad_vhat": gv, "grad_that": gt,
            "logZv": lzv[lo:hi], "logZt": lzt[lo:hi], "diag": stats_all["diag"][lo:hi]}


# --------------------------------------------------------------------------- #
# (c) sharded semantics                                        #
# --------------------------------------------------------------------------- #
def sharded_loss_and_grads(video: torch.Tensor, text: torch.Tensor, world: int, rank: int,
                           temperature: float = 0.03, negative_weight: float = 0.8,
                           block: int = 1024) -> Dict[str, torch.Tensor]:
    """What rank `rank` of `world` must return when the global batch (video,
    text) is row-sharded evenly: the GLOBAL loss and the exact gradient of the
    global loss w.r.t. its own rows."""
    B = video.shape[0]
    assert B % world == 0
    b = B // world
    return streaming_loss_and_grads(video, text, temperature, negative_weight, block,
                                    row_range=(rank * b, (rank + 1) * b))


# --------------------------------------------------------------------------- #
# bf16-operand model (what the bf16 MFMA path is expected to compute)          #
# --------------------------------------------------------------------------- #
def bf16_operand_model_loss(video: torch.Tensor, text: torch.Tensor, temperature: float = 0.03,
                            negative_weight: float = 0.8) -> torch.Tensor:
    """Closed-form loss with the normalised rows rounded to bf16 before the
    similarity products (products/accumulation exact in float64) and the
    diagonal logit kept in fp32 -- a numerical model of the bf16-compute /
    fp32-accumulate kernel, used to set honest tolerances for small cases."""
    vhat32 = F.normalize(video.float(), dim=1)
    that32 = F.normalize(text.float(), dim=1)
    vb = vhat32.bfloat16().double()
    tb = that32.bfloat16().double()
    it, w = 1.0 / temperature, negative_weight
    B = video.shape[0]
    a = vb @ tb.t() * it
    cv = vb @ vb.t() * (it * w)
    ct = tb @ tb.t() * (it * w)
    eye = torch.eye(B, dtype=torch.bool)
    cv[eye] = 0.0
    ct[eye] = 0.0
    lzv = torch.logsumexp(torch.cat([a, cv], 1), 1)
    lzt = torch.logsumexp(torch.cat([a.t(), ct], 1), 1)
    diag = (vhat32.double() * that32.double()).sum(1) * it
    return (lzv + lzt - 2 * diag).sum() / (2 * B)


def bf16_operand_model_loss_and_grads(video: torch.Tensor, text: torch.Tensor, temperature: float = 0.03,
                                      negative_weight: float = 0.8):
    """Loss AND input gradients of the bf16-operand model: the similarity products see the unit rows rounded to bf16, the
    gradient treats that rounding as the identity (straight-through) -- exactly what the kernels do: the backward's weights and
    its second operand are the rounded rows, the normalise-backward is exact.  The yardstick for bf16 gradients where the loss
    (and with it the float64 reference gradient) is nearly zero: aligned pairs."""
    v = video.double().clone().requires_grad_(True)
    t = text.double().clone().requires_grad_(True)
    vh, th = F.normalize(v, dim=1), F.normalize(t, dim=1)
    vb = vh + (vh.detach().float().bfloat16().double() - vh.detach())
    tb = th + (th.detach().float().bfloat16().double() - th.detach())
    it, w = 1.0 / temperature, negative_weight
    B = video.shape[0]
    a = vb @ tb.t() * it
    off = 1.0 - torch.eye(B, dtype=torch.float64)
    cv = (vb @ vb.t()) * (it * w) * off          # the masked self pair keeps logit 0 (trainer/loss.py:96-97)
    ct = (tb @ tb.t()) * (it * w) * off
    lzv = torch.logsumexp(torch.cat([a, cv], 1), 1)
    lzt = torch.logsumexp(torch.cat([a.t(), ct], 1), 1)
    diag = (vh * th).sum(1) * it                  # the positive pair's logit comes from the fp32 rows
    loss = (lzv + lzt - 2 * diag).sum() / (2 * B)
    loss.backward()
    return loss.detach(), v.grad, t.grad


def make_inputs(kind: str, B: int, D: int, seed: int, dtype: torch.dtype = torch.float32
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Deterministic synthetic inputs shared by goldens, tests and bench.

    randn   : v, t ~ N(0,1), v drawn first then t from one generator
    aligned : t = v + 0.3 n  (near-zero-loss regime; stress for bf16)
    cluster : 16 cluster centres + 0.1 noise per modality
    """
    g = torch.Generator().manual_seed(seed)
    if kind == "randn":
        v = torch.randn(B, D, generator=g)
        t = torch.randn(B, D, generator=g)
    elif kind == "aligned":
        v = torch.randn(B, D, generator=g)
        t = v + 0.3 * torch.randn(B, D, generator=g)
    elif kind == "cluster":
        c = torch.randn(16, D, generator=g)
        lab = torch.randint(0, 16, (B,), generator=g)
        v = c[lab] + 0.1 * torch.randn(B, D, generator=g)
        t = c[lab] + 0.1 * torch.randn(B, D, generator=g)
    else:
        raise ValueError(kind)
    return v.to(dtype), t.to(dtype)
