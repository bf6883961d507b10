from typing import Optional

import torch
import torch.nn as nn

from ... import functional as SF


class LocalGlobalRegistration(nn.Module):
    """Local-to-global registration (geotransformer/modules/geotransformer/local_global_registration.py:11-235) kept on the
    device: mutual top-k correspondences per patch pair, one weighted-Procrustes hypothesis per patch pair with at least
    `correspondence_threshold` correspondences (segment reductions instead of the reference's host-side chunk loop and
    CPU SVD), the hypothesis with most inliers, then `num_refinement_steps` re-weighted global solves."""

    def __init__(self, k: int, acceptance_radius: float, mutual: bool = True, confidence_threshold: float = 0.05,
                 use_dustbin: bool = False, use_global_score: bool = False, correspondence_threshold: int = 3,
                 correspondence_limit: Optional[int] = None, num_refinement_steps: int = 5):
        super().__init__()
        if use_dustbin or use_global_score or correspondence_limit is not None or not mutual:
            raise NotImplementedError('LocalGlobalRegistration (HIP): SE3ET settings only (mutual, no dustbin/global score)')
        self.k, self.acceptance_radius, self.mutual = k, acceptance_radius, mutual
        self.confidence_threshold, self.correspondence_threshold = confidence_threshold, correspondence_threshold
        self.num_refinement_steps = num_refinement_steps

    def _rescore(self, ref_pts, src_pts, scores, T):
        res = torch.linalg.norm(ref_pts - SF.apply_transform(src_pts, T), dim=-1)
        return scores * (res < self.acceptance_radius).float()

    @torch.no_grad()
    def forward(self, ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, score_mat, global_scores):
        score_mat = torch.exp(score_mat)
        B = score_mat.shape[0]
        rs, ri = score_mat.topk(self.k, dim=2)
        ss, si = score_mat.topk(self.k, dim=1)
        corr = (torch.zeros_like(score_mat).scatter_(2, ri, rs) > self.confidence_threshold) & \
               (torch.zeros_like(score_mat).scatter_(1, si, ss) > self.confidence_threshold) & \
               (ref_knn_masks[:, :, None] & src_knn_masks[:, None, :])
        b_idx, r_idx, c_idx = torch.nonzero(corr, as_tuple=True)           # one host sync (row-major, as the reference)
        ref_c, src_c = ref_knn_points[b_idx, r_idx], src_knn_points[b_idx, c_idx]
        sc = score_mat[b_idx, r_idx, c_idx]
        counts = torch.bincount(b_idx, minlength=B)
        valid = counts >= self.correspondence_threshold
        if bool(valid.any()):
            Ts = SF.segment_procrustes(src_c, ref_c, sc, b_idx, B)          # (B, 4, 4); rows of invalid patches unused
            res = torch.linalg.norm(ref_c[None] - SF.apply_transform(src_c[None], Ts), dim=2)      # (B, total)
            inl = (res < self.acceptance_radius)
            votes = inl.sum(1).masked_fill(~valid, -1)
            # first maximum among the valid patches in patch order (reference: argmax over the kept chunks)
            best = int(torch.nonzero(votes == votes.max())[0, 0])
            cur = sc * inl[best].float()
        else:
            T = SF.weighted_procrustes(src_c, ref_c, sc)
            cur = self._rescore(ref_c, src_c, sc, T)
        T = SF.weighted_procrustes(src_c, ref_c, cur)
        for _ in range(self.num_refinement_steps - 1):
            cur = self._rescore(ref_c, src_c, sc, T)
            T = SF.weighted_procrustes(src_c, ref_c, cur)
        return ref_c, src_c, sc, T
