"""`from trainer.loss import CrossCLR_onlyIntraModality` -- the import path a training loop written
against amazon-science/crossmodal-contrastive-learning already uses (README.md:24-28 there).
Put this repository ahead of the reference on sys.path and that line picks up the MI355X path."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
import crossclr_amd  # noqa: E402

CrossCLR_onlyIntraModality = crossclr_amd.CrossCLR_onlyIntraModality
crossclr_loss = crossclr_amd.crossclr_loss


# The other two top-level names of the reference module, so that `from trainer.loss import *`-style code keeps
# importing.  Neither is on the hot path: no kernels behind them.
def cosine_sim(emb1, emb2):
    """Similarity matrix emb1 @ emb2^T of two [n, d] embedding sets (reference `trainer/loss.py:7-15`)."""
    return emb1 @ emb2.t()


class MaxMargin_coot(crossclr_amd.loss.nn.Module):
    """Name kept for import compatibility only.  The reference class cannot be constructed either: its __init__
    (`trainer/loss.py:24`) names an undefined class and raises NameError -- the same error is raised here."""

    def __init__(self, use_cuda: bool = False, margin: float = 0.1):
        raise NameError("name 'ContrastiveLoss_coot' is not defined (MaxMargin_coot is unconstructible in the "
                        "reference, trainer/loss.py:24; it is not part of the MI355X hot path)")


__all__ = ["CrossCLR_onlyIntraModality", "crossclr_loss", "cosine_sim", "MaxMargin_coot"]
