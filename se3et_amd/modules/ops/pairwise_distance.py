import torch


def pairwise_distance(x: torch.Tensor, y: torch.Tensor, normalized: bool = False, channel_first: bool = False):
    """Squared pairwise distances (*, N, M), clamped at 0 (geotransformer/modules/ops/pairwise_distance.py:4-30):
    `x2 - 2 x.y + y2`, or `2 - 2 x.y` for unit vectors.  float32 GPU tensors that need no gradient: the HIP kernel
    (csrc/pairwise_distance.hip, inner products on the f32 matrix cores); otherwise (the losses differentiate through it, CPU tensors)
    the reference's own formula through torch.  The fused superpoint-matching and partition kernels do not go through this function."""
    if channel_first:
        x, y = x.transpose(-1, -2), y.transpose(-1, -2)
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or y.requires_grad)
    if x.is_cuda and y.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32 and not needs_grad and x.dim() == y.dim() \
            and x.dim() >= 2 and x.shape[:-2] == y.shape[:-2]:
        from ... import ops as _ops
        return _ops.pairwise_distance(x, y, normalized)
    xy = torch.matmul(x, y.transpose(-1, -2))
    if normalized:
        sq = 2.0 - 2.0 * xy
    else:
        sq = (x * x).sum(-1).unsqueeze(-1) - 2 * xy + (y * y).sum(-1).unsqueeze(-2)
    return sq.clamp_(min=0.0)
