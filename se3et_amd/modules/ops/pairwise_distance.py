import torch


def pairwise_distance(x: torch.Tensor, y: torch.Tensor, normalized: bool = False, channel_first: bool = False):
    """Squared pairwise distances (*, N, M), clamped at 0 (geotransformer/modules/ops/pairwise_distance.py:4-30):
    `x2 - 2 x.y + y2`, or `2 - 2 x.y` for unit vectors.  The inner products are one library GEMM (rocBLAS through
    torch.matmul); the fused superpoint-matching kernel does not go through this function."""
    if channel_first:
        x, y = x.transpose(-1, -2), y.transpose(-1, -2)
    xy = torch.matmul(x, y.transpose(-1, -2))
    if normalized:
        sq = 2.0 - 2.0 * xy
    else:
        sq = (x * x).sum(-1).unsqueeze(-1) - 2 * xy + (y * y).sum(-1).unsqueeze(-2)
    return sq.clamp_(min=0.0)
