import torch

from .pairwise_distance import pairwise_distance


@torch.no_grad()
def point_to_node_partition(points, nodes, point_limit, return_count=False):
    """Assign every fine point to its nearest node and give every node the `point_limit` nearest of ITS OWN points
    (geotransformer/modules/ops/pointcloud_partition.py:60-107).  Returns (point_to_node (N,), [node_sizes (M,)],
    node_masks (M,), node_knn_indices (M, K) padded with N, node_knn_masks (M, K))."""
    n, m = points.shape[0], nodes.shape[0]
    sq = pairwise_distance(nodes, points)                                  # (M, N)
    point_to_node = sq.argmin(0)
    node_masks = torch.zeros(m, dtype=torch.bool, device=points.device)
    node_masks[point_to_node] = True
    own = torch.zeros_like(sq, dtype=torch.bool)
    own[point_to_node, torch.arange(n, device=points.device)] = True
    sq.masked_fill_(~own, 1e12)
    knn = sq.topk(point_limit, dim=1, largest=False)[1]
    knn_masks = point_to_node[knn] == torch.arange(m, device=points.device)[:, None]
    knn.masked_fill_(~knn_masks, n)
    if return_count:
        sizes = torch.bincount(point_to_node, minlength=m)
        return point_to_node, sizes, node_masks, knn, knn_masks
    return point_to_node, node_masks, knn, knn_masks
