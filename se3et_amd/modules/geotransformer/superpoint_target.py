import torch
import torch.nn as nn

from ...training import select_targets


class SuperPointTargetGenerator(nn.Module):
    """Ground-truth superpoint correspondences for the fine-matching loss: those above `overlap_threshold`, at most `num_targets` of them
    (geotransformer/modules/geotransformer/superpoint_target.py:6-41; imported by experiments/se3ete.3dmatch/model.py:9-14, used at
    :56-58,129-131).  The random subset is drawn exactly as the reference draws it -- ONE `np.random.choice(arange(n), num_targets,
    replace=False)` on numpy's global generator, only when more than `num_targets` correspondences pass -- so a seeded reference run
    selects the same targets; `rng` (a numpy Generator / RandomState) replaces the global generator for reproducible tests.  The selected
    indices are moved to the device of the inputs (the reference hard-codes `.cuda()`)."""

    def __init__(self, num_targets, overlap_threshold):
        super().__init__()
        self.num_targets = num_targets
        self.overlap_threshold = overlap_threshold

    @torch.no_grad()
    def forward(self, gt_corr_indices, gt_corr_overlaps, rng=None):
        """(N, 2) int64, (N,) -> gt_ref_corr_indices, gt_src_corr_indices, gt_corr_overlaps of the selection."""
        return select_targets(gt_corr_indices, gt_corr_overlaps, self.num_targets, self.overlap_threshold, rng)
