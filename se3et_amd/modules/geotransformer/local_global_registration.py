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

    @torch.no_grad()
    def forward(self, ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, score_mat, global_scores):
        score_mat = torch.exp(score_mat)
        B = score_mat.shape[0]
        corr = SF.mutual_topk_mask(score_mat, ref_knn_masks, src_knn_masks, self.k, self.confidence_threshold)
        b_idx, r_idx, c_idx = torch.nonzero(corr, as_tuple=True)           # one host sync (row-major, as the reference)
        ref_c, src_c = ref_knn_points[b_idx, r_idx].contiguous(), src_knn_points[b_idx, c_idx].contiguous()
        sc = score_mat[b_idx, r_idx, c_idx].contiguous()
        total = sc.shape[0]
        dev = sc.device
        counts = torch.zeros(B, dtype=torch.int64, device=dev).index_add_(0, b_idx, torch.ones_like(b_idx))   # (bincount syncs)
        offsets = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        offsets[1:] = torch.cumsum(counts, 0)
        whole = SF.to_device([0, total], torch.int64, dev)
        # local hypotheses: one weighted Procrustes per patch pair (patches below the threshold never win the vote)
        Ts = SF.weighted_procrustes(src_c, ref_c, sc, offsets)
        votes = SF.count_inliers(src_c, ref_c, Ts, self.acceptance_radius)
        votes = torch.where(counts >= self.correspondence_threshold, votes, torch.full_like(votes, -1))
        best = torch.argmax(votes)                       # first maximum in patch order, stays on the device
        any_valid = votes.max() >= 0
        T0 = Ts.index_select(0, best.view(1))[0]          # (indexing with a 0-dim device tensor would synchronise)
        # degenerate case (no patch pair with enough correspondences): start from all correspondences instead
        T_all = SF.weighted_procrustes(src_c, ref_c, sc, whole)[0]
        T_init = torch.where(any_valid, T0, T_all)
        # global refinement: weights = score * [residual under the previous estimate < radius], one launch per step
        T = SF.weighted_procrustes(src_c, ref_c, sc, whole, gate_transform=T_init, gate_radius=self.acceptance_radius)[0]
        for _ in range(self.num_refinement_steps - 1):
            T = SF.weighted_procrustes(src_c, ref_c, sc, whole, gate_transform=T, gate_radius=self.acceptance_radius)[0]
        return ref_c, src_c, sc, T
