import torch
import torch.nn as nn

from ... import functional as SF


class SuperPointMatching(nn.Module):
    def __init__(self, num_correspondences, dual_normalization=True):
        super().__init__()
        self.num_correspondences, self.dual_normalization = num_correspondences, dual_normalization

    @torch.no_grad()
    def forward(self, ref_feats, src_feats, ref_masks=None, src_masks=None, all_valid=False):
        """Top-k superpoint pairs by exp(-||f_r - f_s||^2) with dual normalisation; empty nodes dropped.
        Returns (ref_corr_indices, src_corr_indices, corr_scores).  `all_valid=True` (the caller knows that no node is
        empty) skips the two host-synchronising mask compactions."""
        dev = ref_feats.device
        m = src_feats.shape[0]
        if all_valid or (ref_masks is None and src_masks is None):
            scores = SF.superpoint_scores(ref_feats, src_feats, self.dual_normalization)
            k = min(self.num_correspondences, scores.numel())
            corr_scores, flat = scores.view(-1).topk(k=k, largest=True)
            return torch.div(flat, m, rounding_mode='floor'), flat % m, corr_scores
        ref_idx = torch.nonzero(ref_masks)[:, 0] if ref_masks is not None else torch.arange(ref_feats.shape[0], device=dev)
        src_idx = torch.nonzero(src_masks)[:, 0] if src_masks is not None else torch.arange(m, device=dev)
        scores = SF.superpoint_scores(ref_feats[ref_idx].contiguous(), src_feats[src_idx].contiguous(),
                                      self.dual_normalization)
        k = min(self.num_correspondences, scores.numel())
        corr_scores, flat = scores.view(-1).topk(k=k, largest=True)
        m = scores.shape[1]
        return ref_idx[torch.div(flat, m, rounding_mode='floor')], src_idx[flat % m], corr_scores
