"""The criterion inside an ordinary training loop on the MI355X -- projection heads, an optimiser, `loss.backward()`, the import
line of the reference's README (`from trainer.loss import CrossCLR_onlyIntraModality`, README.md:24-38) -- against the same
loop driven by the CPU oracle's op-for-op restatement of the reference: the loss trajectories must agree step by step."""
import pytest
import torch

from oracle import crossclr_oracle as orc

pytestmark = pytest.mark.gpu


def _run(device, criterion, steps, B, Din, D, seed):
    g = torch.Generator().manual_seed(seed)
    xv, xt = torch.randn(B, Din, generator=g), torch.randn(B, Din, generator=g)
    xt = xt + 0.5 * xv                                   # some shared structure to learn
    torch.manual_seed(seed)
    heads = torch.nn.ModuleDict({"v": torch.nn.Linear(Din, D), "t": torch.nn.Linear(Din, D)}).to(device)
    opt = torch.optim.SGD(heads.parameters(), lr=0.2, momentum=0.5)   # (gentle: the two devices' GEMMs differ in the last bits)
    xv, xt = xv.to(device), xt.to(device)
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        loss = criterion(heads["v"](xv), heads["t"](xt))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses


@pytest.mark.parametrize("mode,B,D,tol", [("fp32", 512, 128, 5e-4), ("bf16", 2048, 256, 1e-2), ("fp32", 300, 96, 5e-4)])
def test_training_loop_follows_the_reference_trajectory(mode, B, D, tol):
    from trainer.loss import CrossCLR_onlyIntraModality          # the reference's import line, resolved to the MI355X path
    crit = CrossCLR_onlyIntraModality(temperature=0.05, negative_weight=0.8, compute_mode=mode).cuda()
    got = _run("cuda", crit, 6, B, 64, D, 3)
    want = _run("cpu", lambda v, t: orc.eager_loss(v, t, 0.05, 0.8), 6, B, 64, D, 3)
    assert got[-1] < got[0]                                       # it learns
    for a, b in zip(got, want):
        assert abs(a - b) <= tol * max(1.0, abs(b)), (got, want)


def test_max_margin_training_loop_follows_the_reference_trajectory():
    from trainer.loss import MaxMargin_coot
    from oracle import ranking_oracle as rk
    norm = lambda x: torch.nn.functional.normalize(x, dim=1)
    crit = MaxMargin_coot(use_cuda=True, margin=0.2, compute_mode="fp32")
    got = _run("cuda", lambda a, b: crit(norm(a), norm(b)), 6, 384, 64, 96, 5)
    want = _run("cpu", lambda a, b: rk.max_margin_eager(norm(a), norm(b), 0.2), 6, 384, 64, 96, 5)
    assert got[-1] < got[0]
    for a, b in zip(got, want):
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (got, want)
