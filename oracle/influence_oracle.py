"""CPU oracle for the sample-weighted form of the CrossCLR loss (BASELINE config 5:
"influential-sample negative weighting").

TEST INFRASTRUCTURE ONLY -- same rules as crossclr_oracle.py: nothing in the product package imports
this file.

PARITY UNPINNED.  The reference @ v1 (`/root/reference/trainer/loss.py`) contains only the scalar
`negative_weight` (`:56,99-100`); its README calls the release "onlyIntraModality" (`README.md:19`).
The influential-sample recipe below restates the CrossCLR paper (Zolfaghari et al., ICCV 2021,
section 3.2/3.3) as recalled in SURVEY.md 8(f); there is no reference code or golden vector to
check it against.  What IS pinned: with keep == 1 and loss_weight == 1 every function here reduces
to the reference-checked functions of crossclr_oracle.py (tests/test_oracle.py asserts that).

Generalised loss (k = per-sample "negative scale" in [0, inf), omega = per-sample loss weight):

    Zv[i] = sum_j exp(A[i,j]) + sum_{j != i} kv[j] exp(w Cv[i,j]) + kv[i] * exp(0)
    Zt[i] = sum_j exp(A[j,i]) + sum_{j != i} kt[j] exp(w Ct[i,j]) + kt[i] * exp(0)
    loss  = ( sum_i ov[i] (log Zv[i] - A_ii) + sum_i ot[i] (log Zt[i] - A_ii) ) / (2 B)

k in {0,1} = a sample pruned from / kept in the intra-modal negative set (the masked diagonal, logit
0.0 in the reference `:96-97`, travels with its column); omega = B * rho / sum(rho) so that
omega == 1 is the reference's plain mean.  k and omega are constants w.r.t. the embeddings.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import crossclr_oracle as base


# --------------------------------------------------------------------------- #
# the influential-sample recipe (input-space connectivity -> keep mask, weights) #
# --------------------------------------------------------------------------- #
def influence_weights(input_vid: torch.Tensor, input_txt: torch.Tensor, score_threshold: float = 0.7,
                      temperature_weights: float = 0.0035) -> Dict[str, torch.Tensor]:
    """Dense float64 restatement of the paper's recipe, one modality at a time:

        s[i,j]  = xhat_i . xhat_j  on the INPUT-space features, diagonal masked to 0
        conn[i] = mean_j s[i,j]                              (connectivity of sample i)
        keep[i] = conn[i] / max_j conn[j] < score_threshold  (highly connected = "influential" samples
                                                              are removed from the negative set)
        rho[i]  = exp( (conn[i] / sum_j conn[j]) / temperature_weights )
        omega   = B * rho / sum(rho)                         (influential samples weigh more)
    """
    out = {}
    for name, x in (("v", input_vid), ("t", input_txt)):
        xh = F.normalize(x.double(), dim=1)
        n = xh.shape[0]
        s = xh @ xh.t()
        s = s * (1 - torch.eye(n, dtype=torch.float64))
        conn = s.mean(1)
        keep = (conn / conn.max() < score_threshold).double()
        rho = torch.exp((conn / conn.sum()) / temperature_weights)
        out["conn_" + name] = conn
        out["keep_" + name] = keep
        out["omega_" + name] = n * rho / rho.sum()
    return out


def influence_weights_streaming(input_vid: torch.Tensor, input_txt: torch.Tensor, score_threshold: float = 0.7,
                                temperature_weights: float = 0.0035) -> Dict[str, torch.Tensor]:
    """The same recipe without the B x B matrix (usable at B = 65536):
    mean_j s[i,j] with the diagonal masked = (xhat_i . sum_j xhat_j - xhat_i . xhat_i) / B.  Checked against
    `influence_weights` in tests/test_sample_weights_cpu.py."""
    out = {}
    for name, x in (("v", input_vid), ("t", input_txt)):
        xh = F.normalize(x.double(), dim=1)
        n = xh.shape[0]
        conn = (xh @ xh.sum(0) - (xh * xh).sum(1)) / n
        keep = (conn / conn.max() < score_threshold).double()
        rho = torch.exp((conn / conn.sum()) / temperature_weights)
        out["conn_" + name] = conn
        out["keep_" + name] = keep
        out["omega_" + name] = n * rho / rho.sum()
    return out


# --------------------------------------------------------------------------- #
# literal dense form (column selection + masked softmax), binary keep only       #
# --------------------------------------------------------------------------- #
def eager_pruned_loss(video, text, temperature, negative_weight, keep_v, keep_t, omega_v, omega_t) -> torch.Tensor:
    """The reference's op sequence (loss.py:76-114) with the two changes of the recipe: the intra-modal
    negative block keeps only the columns with keep == 1, and the per-row losses are averaged with weights."""
    n = video.shape[0]
    vhat, that = F.normalize(video.double(), dim=1), F.normalize(text.double(), dim=1)
    inter_v, inter_t = vhat @ that.t() / temperature, that @ vhat.t() / temperature
    off = 1 - torch.eye(n, dtype=torch.float64)
    neg_v = (vhat @ vhat.t() / temperature * off)[:, keep_v.bool()]
    neg_t = (that @ that.t() / temperature * off)[:, keep_t.bool()]
    rows_v = torch.cat([inter_v, negative_weight * neg_v], 1)
    rows_t = torch.cat([inter_t, negative_weight * neg_t], 1)
    eye = torch.eye(n, dtype=torch.float64)
    per_v = -torch.log((F.softmax(rows_v, 1) * torch.cat([eye, torch.zeros_like(neg_v)], 1)).sum(1))
    per_t = -torch.log((F.softmax(rows_t, 1) * torch.cat([eye, torch.zeros_like(neg_t)], 1)).sum(1))
    return ((per_v * omega_v).sum() + (per_t * omega_t).sum()) / (2 * n)


# --------------------------------------------------------------------------- #
# closed form, dense, autograd (any k >= 0)                                      #
# --------------------------------------------------------------------------- #
def dense_weighted_loss(video, text, temperature, negative_weight, k_v, k_t, omega_v, omega_t) -> torch.Tensor:
    n = video.shape[0]
    vhat, that = F.normalize(video.double(), dim=1), F.normalize(text.double(), dim=1)
    a = vhat @ that.t() / temperature
    off = 1 - torch.eye(n, dtype=torch.float64)
    cv = negative_weight * (vhat @ vhat.t() / temperature) * off
    ct = negative_weight * (that @ that.t() / temperature) * off
    m = max(1.0, abs(negative_weight)) / temperature
    zv = torch.exp(a - m).sum(1) + (torch.exp(cv - m) * k_v[None, :].double()).sum(1)
    zt = torch.exp(a.t() - m).sum(1) + (torch.exp(ct - m) * k_t[None, :].double()).sum(1)
    d = torch.diagonal(a)
    return ((omega_v.double() * (torch.log(zv) + m - d)).sum() + (omega_t.double() * (torch.log(zt) + m - d)).sum()) / (2 * n)


def dense_weighted_loss_and_grads(video, text, temperature, negative_weight, k_v, k_t, omega_v, omega_t):
    v = video.detach().double().clone().requires_grad_(True)
    t = text.detach().double().clone().requires_grad_(True)
    loss = dense_weighted_loss(v, t, temperature, negative_weight, k_v, k_t, omega_v, omega_t)
    loss.backward()
    return {"loss": loss.detach(), "grad_v": v.grad, "grad_t": t.grad}


# --------------------------------------------------------------------------- #
# closed form, streaming float64 (nothing O(B^2) resident; row_range for the sharded semantics)      #
# --------------------------------------------------------------------------- #
def streaming_weighted_loss_and_grads(video, text, temperature, negative_weight, k_v, k_t, omega_v, omega_t,
                                      block: int = 1024, row_range: Optional[Tuple[int, int]] = None,
                                      logz_all: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
                                      ) -> Dict[str, torch.Tensor]:
    """W for the gradient:  inter  E (ov_i/Zv_i + ot_j/Zt_j);  intra  w E (o_i k_j / Z_i + o_j k_i / Z_j), diag 0;
    positive-pair term  -(ov_i + ot_i)/(2 B tau) * partner."""
    B = video.shape[0]
    lo, hi = (0, B) if row_range is None else row_range
    vhat, vnorm = base._unit_rows(video)
    that, tnorm = base._unit_rows(text)
    it, w = 1.0 / float(temperature), float(negative_weight)
    kv, kt, ov, ot = (x.double() for x in (k_v, k_t, omega_v, omega_t))
    lzv = torch.empty(B, dtype=torch.float64)
    lzt = torch.empty(B, dtype=torch.float64)
    neg_inf = float("-inf")
    if logz_all is not None:      # (the full-batch denominators of an earlier call: the O(B^2 D) part)
        lzv, lzt = logz_all[0].double().clone(), logz_all[1].double().clone()
    for r0 in range(0, B if logz_all is None else 0, block):
        r1 = min(B, r0 + block)
        idx, rows = torch.arange(r0, r1), torch.arange(r1 - r0)
        for own, other, k, out in ((vhat, that, kv, lzv), (that, vhat, kt, lzt)):
            inter = (own[r0:r1] @ other.t()) * it
            intra = (own[r0:r1] @ own.t()) * (it * w)
            intra[rows, idx] = 0.0
            intra = intra + torch.where(k > 0, torch.log(k.clamp_min(1e-300)), torch.full_like(k, neg_inf))[None, :]
            out[r0:r1] = torch.logsumexp(torch.cat([inter, intra], 1), 1)
    diag = (vhat * that).sum(1) * it
    loss = ((ov * (lzv - diag)).sum() + (ot * (lzt - diag)).sum()) / (2.0 * B)
    gv = torch.zeros(hi - lo, video.shape[1], dtype=torch.float64)
    gt = torch.zeros_like(gv)
    for r0 in range(lo, hi, block):
        r1 = min(hi, r0 + block)
        idx, rows = torch.arange(r0, r1), torch.arange(r1 - r0)
        for own, other, lz_own, lz_oth, k, o_own, o_oth, out in ((vhat, that, lzv, lzt, kv, ov, ot, gv),
                                                                 (that, vhat, lzt, lzv, kt, ot, ov, gt)):
            a = (own[r0:r1] @ other.t()) * it
            ga = torch.exp(a - lz_own[r0:r1, None]) * o_own[r0:r1, None] + torch.exp(a - lz_oth[None, :]) * o_oth[None, :]
            c = (own[r0:r1] @ own.t()) * (it * w)
            s = (torch.exp(c - lz_own[r0:r1, None]) * (o_own[r0:r1, None] * k[None, :]) +
                 torch.exp(c - lz_own[None, :]) * (o_own[None, :] * k[r0:r1, None]))
            s[rows, idx] = 0.0
            out[r0 - lo:r1 - lo] = ga @ other + w * (s @ own)
    scale = it / (2.0 * B)
    pos = ((ov + ot) * scale)[lo:hi, None]
    gv = gv * scale - that[lo:hi] * pos
    gt = gt * scale - vhat[lo:hi] * pos

    def through_normalize(g, xhat, x, nrm):
        proj = g - xhat * (xhat * g).sum(1, keepdim=True)
        tiny = (x.double().norm(dim=1) < base.NORM_EPS)[:, None]
        return torch.where(tiny, g, proj) / nrm[:, None]

    return {"loss": loss, "logZv": lzv[lo:hi], "logZt": lzt[lo:hi], "logZv_all": lzv, "logZt_all": lzt,
            "grad_v": through_normalize(gv, vhat[lo:hi], video[lo:hi], vnorm[lo:hi]),
            "grad_t": through_normalize(gt, that[lo:hi], text[lo:hi], tnorm[lo:hi])}
