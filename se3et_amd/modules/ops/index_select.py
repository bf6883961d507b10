import torch


def index_select(data: torch.Tensor, index: torch.Tensor, dim: int) -> torch.Tensor:
    """`data` indexed along `dim` by an index tensor of any rank (geotransformer/modules/ops/index_select.py:4-33)."""
    out = data.index_select(dim, index.reshape(-1))
    if index.dim() != 1:
        out = out.reshape(data.shape[:dim] + index.shape + data.shape[dim + 1:])
    return out
