"""TEST INFRASTRUCTURE -- CPU restatement of the reference's max-margin ranking loss and of retrieval ranks.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package; the product path never does.

`max_margin_eager` follows `MaxMargin_coot.forward` of amazon-science/crossmodal-contrastive-learning op for op
(trainer/loss.py:29-41, with `cosine_sim` of loss.py:7-15).  The reference CLASS cannot be constructed (loss.py:24 names an
undefined class), but its `forward` is an ordinary function: tests/golden/make_golden_ranking.py calls it on a stand-in
`self` carrying the three attributes the constructor would have set, and pins this restatement bit for bit against it
(loss and both autograd gradients) -- parity pinned.

`retrieval_ranks_dense` (rank of each sample's partner among the other modality's candidates; the evaluation that follows
training in the CrossCLR / COOT pipelines) is NOT in the reference repository: parity unpinned, the definition is stated here.
"""
from typing import Dict

import torch
import torch.nn.functional as F


def max_margin_eager(im: torch.Tensor, s: torch.Tensor, margin: float = 0.1) -> torch.Tensor:
    scores = im.mm(s.t())                                          # loss.py:30 (cosine_sim, loss.py:15)
    diagonal = scores.diag().view(im.size(0), 1)                   # :31
    d1 = diagonal.expand_as(scores)                                # :32
    d2 = diagonal.t().expand_as(scores)                            # :33
    cost_s = (margin + scores - d1).clamp(min=0)                   # :34
    cost_im = (margin + scores - d2).clamp(min=0)                  # :35
    mask = torch.eye(scores.size(0)) > .5                          # :36
    cost_s = cost_s.masked_fill_(mask, 0)                          # :39
    cost_im = cost_im.masked_fill_(mask, 0)                        # :40
    return (cost_s.sum() + cost_im.sum()).div(im.shape[0] * s.shape[0])   # :41


def max_margin_loss_and_grads(im: torch.Tensor, s: torch.Tensor, margin: float = 0.1) -> Dict[str, torch.Tensor]:
    a = im.detach().clone().requires_grad_(True)
    b = s.detach().clone().requires_grad_(True)
    loss = max_margin_eager(a, b, margin)
    loss.backward()
    return {"loss": loss.detach(), "grad_im": a.grad, "grad_s": b.grad}


def max_margin_streaming(im: torch.Tensor, s: torch.Tensor, margin: float = 0.1) -> Dict[str, torch.Tensor]:
    """float64 closed form (what the kernels evaluate): hinge sums per row / column, gradients from indicator weights."""
    a, b = im.double(), s.double()
    B = a.shape[0]
    S = a @ b.t()
    d = S.diag()
    off = ~torch.eye(B, dtype=torch.bool)
    h1 = (margin + S - d[:, None]).clamp(min=0) * off        # rows: im_i against every s_j
    h2 = (margin + S - d[None, :]).clamp(min=0) * off        # columns: s_j against every im_i
    W = ((h1 > 0).double() + (h2 > 0).double())
    c = (h1 > 0).sum(1).double() + (h2 > 0).sum(0).double()  # active hinges that contain S_ii
    g_im = (W @ b - c[:, None] * b) / (B * B)
    g_s = (W.t() @ a - c[:, None] * a) / (B * B)
    return {"loss": (h1.sum() + h2.sum()) / (B * B), "grad_im": g_im, "grad_s": g_s,
            "hinge_im": h1.sum(1), "hinge_s": h2.sum(0), "active_im": (h1 > 0).sum(1), "active_s": (h2 > 0).sum(0)}


def retrieval_ranks_dense(video: torch.Tensor, text: torch.Tensor, normalize: bool = True) -> Dict[str, torch.Tensor]:
    """rank (from 0) of the partner: number of candidates of the other modality scoring STRICTLY higher; float64 scores."""
    v, t = video.double(), text.double()
    if normalize:
        v, t = F.normalize(v, dim=1), F.normalize(t, dim=1)
    S = v @ t.t()
    d = S.diag()
    out = {"v2t_ranks": (S > d[:, None]).sum(1), "t2v_ranks": (S > d[None, :]).sum(0), "scores": S}
    for key in ("v2t", "t2v"):
        r = out[key + "_ranks"].double()
        out[key] = torch.stack([(r < 1).double().mean(), (r < 5).double().mean(), (r < 10).double().mean(), r.median() + 1.0, r.mean() + 1.0])
    return out
