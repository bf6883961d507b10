from typing import Optional

import torch

from ...training import node_correspondences


@torch.no_grad()
def get_node_correspondences(ref_nodes: torch.Tensor, src_nodes: torch.Tensor, ref_knn_points: torch.Tensor,
                             src_knn_points: torch.Tensor, transform: torch.Tensor, pos_radius: float,
                             ref_masks: Optional[torch.Tensor] = None, src_masks: Optional[torch.Tensor] = None,
                             ref_knn_masks: Optional[torch.Tensor] = None, src_knn_masks: Optional[torch.Tensor] = None):
    """Ground-truth superpoint (patch) correspondences under `transform` (geotransformer/modules/registration/matching.py:231-315;
    imported by experiments/se3ete.3dmatch/model.py:7, called at :110-121): patch pairs (M superpoints with K nearest points each,
    against N) holding at least one pair of points closer than `pos_radius`, overlap = mean of the two covered fractions.
    -> corr_indices (C, 2) int64 in row-major (ref, src) order, corr_overlaps (C,).  Absent masks mean "all valid"; they are created on the
    device of the inputs (the reference hard-codes `.cuda()`).  The node and patch distance matrices run on se3_pairwise_distance for
    float32 GPU tensors (modules/ops/pairwise_distance.py)."""
    dev = ref_nodes.device
    ones = lambda *shape: torch.ones(shape, dtype=torch.bool, device=dev)
    ref_masks = ones(ref_nodes.shape[0]) if ref_masks is None else ref_masks
    src_masks = ones(src_nodes.shape[0]) if src_masks is None else src_masks
    ref_knn_masks = ones(*ref_knn_points.shape[:2]) if ref_knn_masks is None else ref_knn_masks
    src_knn_masks = ones(*src_knn_points.shape[:2]) if src_knn_masks is None else src_knn_masks
    return node_correspondences(ref_nodes, src_nodes, ref_knn_points, src_knn_points, transform, pos_radius, ref_masks, src_masks,
                                ref_knn_masks, src_knn_masks)
