"""MI355X-native CrossCLR contrastive-loss hot path (drop-in for
amazon-science/crossmodal-contrastive-learning `trainer/loss.py: CrossCLR_onlyIntraModality`).

The directory name carries a hyphen (it mirrors the reference repository's name), so it is imported
through the `crossclr_amd` alias module at the repository root:

    import crossclr_amd
    criterion = crossclr_amd.CrossCLR_onlyIntraModality(temperature=0.03, negative_weight=0.8)
"""
from . import _native
from .loss import AUTO_BF16_MIN_GLOBAL_BATCH, CrossCLR_onlyIntraModality, all_gather_with_grad, crossclr_loss
from .influence import CrossCLR, influential_sample_weights
from .ranking import MaxMargin_coot, cosine_sim, max_margin_loss, retrieval_ranks
from .projection import ProjectedCrossCLR, projected_crossclr_loss

__all__ = ["CrossCLR_onlyIntraModality", "CrossCLR", "crossclr_loss", "all_gather_with_grad", "influential_sample_weights",
           "MaxMargin_coot", "max_margin_loss", "cosine_sim", "retrieval_ranks", "ProjectedCrossCLR", "projected_crossclr_loss", "AUTO_BF16_MIN_GLOBAL_BATCH", "_native"]
