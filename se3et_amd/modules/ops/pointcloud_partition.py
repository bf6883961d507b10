import torch

from ... import ops as _ops
from .pairwise_distance import pairwise_distance


@torch.no_grad()
def get_point_to_node_indices(points, nodes, return_counts=False):
    """Index of the nearest node of every point (N,), optionally the number of points per node (M,)
    (geotransformer/modules/ops/pointcloud_partition.py:9-32; used by the reference's ground-truth helpers, not by the forward, which
    takes the fused `point_to_node_partition` below)."""
    nearest = pairwise_distance(points, nodes).argmin(dim=1)
    if return_counts:
        return nearest, torch.bincount(nearest, minlength=nodes.shape[0])
    return nearest


@torch.no_grad()
def point_to_node_partition(points, nodes, point_limit, return_count=False):
    """Assign every fine point to its nearest node and give every node the `point_limit` nearest of ITS OWN points
    (geotransformer/modules/ops/pointcloud_partition.py:60-107).  Returns (point_to_node (N,), [node_sizes (M,)],
    node_masks (M,), node_knn_indices (M, K) padded with N, node_knn_masks (M, K)).  One C call (two kernels,
    csrc/partition.hip) instead of the reference's (M, N) distance matrix + argmin + masked top-k.  K <= 128."""
    point_to_node, node_masks, knn, knn_masks = _ops.point_to_node_partition(points, nodes, point_limit)
    if return_count:
        sizes = torch.bincount(point_to_node, minlength=nodes.shape[0])
        return point_to_node, sizes, node_masks, knn, knn_masks
    return point_to_node, node_masks, knn, knn_masks
