"""Decoder helpers the SE3ET backbone takes from geotransformer.modules.kpconv
(functional.py:6-22 nearest_upsample; modules.py:34-110 GroupNorm / UnaryBlock / LastUnaryBlock)."""
import torch
import torch.nn as nn

from ... import functional as SF


def nearest_upsample(x, upsample_indices):
    """Feature of the nearest coarse point (column 0 of the sorted neighbour table; padded index -> zeros)."""
    return SF.gather_rows_padded(x, upsample_indices[:, 0])


def _upsampled_linear(coarse, upsample_indices, skip, weight):
    """cat(nearest_upsample(coarse), skip) W^T as nearest_upsample(coarse W_a^T) + skip W_b^T: the gather commutes with the dense layer (a
    padded index gathers a zero row either way), so the product with the coarse features runs on the coarse rows (4x fewer) and neither the
    upsampled features nor the concatenation are written.  kpconv/modules.py:105-113 + e2pn backbone decoder (backbone.py)."""
    cu = coarse.shape[1]
    up = nearest_upsample(SF.linear(coarse, weight[:, :cu]), upsample_indices)
    return SF.linear(skip, weight[:, cu:]).add_(up)


class GroupNorm(nn.Module):
    def __init__(self, num_groups, num_channels):
        super().__init__()
        self.num_groups, self.num_channels = num_groups, num_channels
        self.norm = nn.GroupNorm(num_groups, num_channels)

    def forward(self, x, leaky_slope=None, x_bias=None):
        return SF.group_norm_rows(x, self.norm.weight, self.norm.bias, self.num_groups, self.norm.eps, leaky_slope, None, x_bias)


class UnaryBlock(nn.Module):
    def __init__(self, in_channels, out_channels, group_norm, has_relu=True, bias=True, layer_norm=False):
        super().__init__()
        if layer_norm:
            raise NotImplementedError('UnaryBlock (HIP): layer_norm=True is not used by SE3ET')
        self.mlp = nn.Linear(in_channels, out_channels, bias=bias)
        self.norm = GroupNorm(group_norm, out_channels)
        self.has_relu = has_relu

    def forward(self, x):
        return self.norm(SF.linear(x, self.mlp.weight), leaky_slope=0.1 if self.has_relu else None, x_bias=self.mlp.bias)

    def forward_upsampled(self, coarse, upsample_indices, skip):
        """forward(cat(nearest_upsample(coarse), skip)) without the concatenated tensor (inference)."""
        return self.norm(_upsampled_linear(coarse, upsample_indices, skip, self.mlp.weight), leaky_slope=0.1 if self.has_relu else None,
                         x_bias=self.mlp.bias)


class LastUnaryBlock(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.mlp = nn.Linear(in_channels, out_channels, bias=bias)

    def forward(self, x):
        return SF.linear(x, self.mlp.weight, self.mlp.bias)

    def forward_upsampled(self, coarse, upsample_indices, skip):
        y = _upsampled_linear(coarse, upsample_indices, skip, self.mlp.weight)
        return y if self.mlp.bias is None else y.add_(self.mlp.bias)
