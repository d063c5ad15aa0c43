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


# The other two top-level names of the reference module (trainer/loss.py:7-41).  The reference's MaxMargin_coot cannot be
# constructed (its __init__ names an undefined class, loss.py:24); the class exported here is the working implementation of
# its forward (loss.py:29-41) on the tiled score kernels -- same constructor arguments, same attributes.
cosine_sim = crossclr_amd.cosine_sim
MaxMargin_coot = crossclr_amd.MaxMargin_coot

__all__ = ["CrossCLR_onlyIntraModality", "crossclr_loss", "cosine_sim", "MaxMargin_coot"]
