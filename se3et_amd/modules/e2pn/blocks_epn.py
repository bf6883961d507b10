"""E2PN anchor-equivariant KPConv blocks on HIP kernels.

Mirror of the SE3ET-relevant classes of geotransformer/modules/e2pn/blocks_epn.py (KPConvInterSO3 :18-552,
UnaryBlockEPN :639-665, GroupNormEPN :684-701, KPConvInterSO3Block :703-743, SimpleBlockEPN :770-796,
ResnetBottleneckBlockEPN :798-852, InvOutBlockEPN :854-926, LiftBlockEPN :993-1004): same constructor arguments,
forward signatures and parameter / buffer names.  Only the configuration the SE3ET
experiments use is implemented (kanchor 6, quotient 4, 15 kernel points, non-separable rotate-by-permute conv,
linear influence, sum aggregation); anything else raises NotImplementedError.

Layout: features are (P, A, C) float32, points of ref and src stacked; neighbour tables (P, NN) int64 padded
with the support size.
"""
import math

import torch
import torch.nn as nn

from ... import functional as SF
from ... import tables


def _check_epn_config(config):
    ok = (config.kanchor == 6 and config.quotient_factor == 4 and config.num_kernel_points == 15
          and config.non_sep_conv and config.rot_by_permute and config.equiv_mode_kp
          and config.fixed_kernel_points == 'center' and config.KP_influence == 'linear'
          and config.aggregation_mode == 'sum' and not getattr(config, 'epn_kernel', False)
          and not getattr(config, 'ignore_steer_constraint', False))
    if not ok:
        raise NotImplementedError('the HIP E2PN path implements the SE3ET configuration only '
                                  '(kanchor=6, quotient_factor=4, 15 kernel points, non_sep_conv, rot_by_permute)')


class KPConvInterSO3(nn.Module):
    def __init__(self, kernel_size, kanchor, in_channels, out_channels, KP_extent, radius, KP_influence='linear',
                 aggregation_mode='sum', deformable=False, modulated=False, epn_kernel=False, equiv_mode_kp=False,
                 non_sep_conv=False, rot_by_permute=False, fixed_kernel_points='center', quotient_factor=1,
                 ignore_steer_constraint=False, gather_by_idxing=False):
        super().__init__()
        if not (kernel_size == 15 and kanchor == 6 and quotient_factor == 4 and non_sep_conv and rot_by_permute
                and equiv_mode_kp and fixed_kernel_points == 'center' and KP_influence == 'linear'
                and aggregation_mode == 'sum' and not deformable and not epn_kernel and not ignore_steer_constraint):
            raise NotImplementedError('KPConvInterSO3 (HIP): unsupported configuration')
        self.K, self.kanchor, self.K_real = kernel_size, kanchor, tables.NUM_WEIGHT_SLOTS
        self.in_channels, self.out_channels = in_channels, out_channels
        self.radius, self.KP_extent = radius, KP_extent
        as_param = lambda a: nn.Parameter(torch.from_numpy(a), requires_grad=False)
        self.kernel_points = as_param(tables.kernel_points(radius))
        self.quotient_anchors = as_param(tables.quotient_anchors())
        self.anchors = as_param(tables.anchors())
        self.weights = nn.Parameter(torch.zeros((self.K_real, kanchor, in_channels, out_channels)))
        kidx = torch.from_numpy(tables.kernel_slot_table())                 # (K, R)
        ridx = torch.from_numpy(tables.anchor_slot_table())                 # (A, R)
        self.register_buffer('kidx_rot', kidx[:, None, :].expand(-1, kanchor, -1).contiguous())
        self.register_buffer('ridx_rot', ridx[None].expand(self.K, -1, -1).contiguous())
        nn.init.kaiming_uniform_(self.weights, a=math.sqrt(5))

    def forward(self, q_pts, s_pts, neighb_inds, x):
        return SF.kpconv_inter_so3(x, q_pts, s_pts, neighb_inds, self.kernel_points, self.weights,
                                   self.kidx_rot[:, 0, :], self.ridx_rot[0], self.KP_extent)

    def __repr__(self):
        return 'KPConvInterSO3(radius: {:.2f}, extent: {:.2f}, in_feat: {:d}, out_feat: {:d})'.format(
            self.radius, self.KP_extent, self.in_channels, self.out_channels)


class GroupNormEPN(nn.Module):
    """GroupNorm whose statistics span (channels of the group) x anchors x ALL stacked points."""

    def __init__(self, num_groups, num_channels):
        super().__init__()
        self.num_groups, self.num_channels = num_groups, num_channels
        self.norm = nn.GroupNorm(num_groups, num_channels)      # parameter container (names: norm.weight / norm.bias)

    def forward(self, x, leaky_slope=None, residual=None, x_bias=None):
        return SF.group_norm_rows(x, self.norm.weight, self.norm.bias, self.num_groups, self.norm.eps, leaky_slope,
                                  residual, x_bias)

    def pending(self, x, leaky_slope=None):
        """Inference: the statistics only -- x (a tensor, or a Pending with one stage) with this norm [+ LeakyReLU] pending on it."""
        return SF.norm_stats(x, self.norm.weight, self.norm.bias, self.num_groups, self.norm.eps, 1.0 if leaky_slope is None else leaky_slope)


class UnaryBlockEPN(nn.Module):
    def __init__(self, in_dim, out_dim, group_norm, bn_momentum, no_relu=False):
        super().__init__()
        self.no_relu, self.in_dim, self.out_dim = no_relu, in_dim, out_dim
        self.mlp = nn.Linear(in_dim, out_dim)
        self.norm = GroupNormEPN(group_norm, out_dim)

    def forward(self, x, batch=None, residual=None, final_slope=None):
        """`residual`/`final_slope` fuse the bottleneck tail lrelu(norm(mlp(x)) + shortcut) into the norm kernel."""
        x = SF.linear(x, self.mlp.weight)                    # the mlp bias is applied inside the GroupNorm kernels
        if residual is not None or final_slope is not None:
            return self.norm(x, leaky_slope=final_slope, residual=residual, x_bias=self.mlp.bias)
        return self.norm(x, leaky_slope=None if self.no_relu else 0.1, x_bias=self.mlp.bias)

    def pending_ok(self, x):
        return SF.dense_norm_ok(x, self.mlp.weight, self.norm.num_groups)

    def pending(self, x):
        """Inference: mlp with the norm's statistics from the GEMM epilogue (csrc/dense_norm.hip); x may itself be pending.  -> Pending."""
        n = self.norm
        p = SF.dense_norm(x, self.mlp.weight, self.mlp.bias, n.norm.weight, n.norm.bias, n.num_groups, n.norm.eps)
        p.slopes[-1] = 1.0 if self.no_relu else 0.1
        return p

    def stats(self, x):
        """Inference: the affine table of norm(mlp(x)) from a GEMM that stores nothing (the expanding layers of a block: their product is
        recomputed by the block's last kernel instead of making a round trip through HBM)."""
        n = self.norm
        return SF.dense_stats(x, self.mlp.weight, self.mlp.bias, n.norm.weight, n.norm.bias, n.num_groups, n.norm.eps)


class LastUnaryBlockEPN(nn.Module):
    def __init__(self, in_dim, out_dim, bias=True):
        super().__init__()
        self.mlp = nn.Linear(in_dim, out_dim, bias=bias)

    def forward(self, x):
        return SF.linear(x, self.mlp.weight, self.mlp.bias)


class KPConvInterSO3Block(nn.Module):
    def __init__(self, block_name, in_dim, out_dim, radius, sigma, group_norm, config):
        super().__init__()
        _check_epn_config(config)
        self.block_name, self.in_dim, self.out_dim = block_name, in_dim, out_dim
        self.conv = KPConvInterSO3(config.num_kernel_points, config.kanchor, in_dim, out_dim, sigma, radius,
                                   config.KP_influence, config.aggregation_mode, epn_kernel=config.epn_kernel,
                                   equiv_mode_kp=config.equiv_mode_kp, non_sep_conv=config.non_sep_conv,
                                   rot_by_permute=config.rot_by_permute, fixed_kernel_points=config.fixed_kernel_points,
                                   quotient_factor=config.quotient_factor,
                                   ignore_steer_constraint=config.ignore_steer_constraint,
                                   gather_by_idxing=config.gather_by_idxing)
        self.norm = GroupNormEPN(group_norm, out_dim)

    def forward(self, x, q_pts, s_pts, neighb_inds):
        return self.norm(self.conv(q_pts, s_pts, neighb_inds, x), leaky_slope=0.1)


class SimpleBlockEPN(nn.Module):
    def __init__(self, block_name, in_dim, out_dim, radius, sigma, group_norm, config):
        super().__init__()
        self.block_name, self.in_dim, self.out_dim = block_name, in_dim, out_dim
        self.interso3 = KPConvInterSO3Block(block_name, in_dim, out_dim, radius, sigma, group_norm, config)
        self.norm = GroupNormEPN(group_norm, out_dim)

    def forward(self, x, q_pts, s_pts, neighb_inds):
        if (SF.PENDING_NORM and not SF.AG.needs_grad(x, self.interso3.conv.weights, self.norm.norm.weight) and x.is_cuda
                and self.out_dim in (16, 32, 64, 128, 256, 512, 1024)):
            # inference: both norms as statistics passes over the convolution's output, applied together in one pass
            y = self.interso3.norm.pending(self.interso3.conv(q_pts, s_pts, neighb_inds, x), 0.1)
            return SF.norm_apply(self.norm.pending(y, 0.1))
        return self.norm(self.interso3(x, q_pts, s_pts, neighb_inds), leaky_slope=0.1)


class ResnetBottleneckBlockEPN(nn.Module):
    def __init__(self, block_name, in_dim, out_dim, radius, sigma, group_norm, config):
        super().__init__()
        self.block_name, self.in_dim, self.out_dim = block_name, in_dim, out_dim
        bn = getattr(config, 'batch_norm_momentum', 0.99)
        mid = out_dim // 4
        self.unary1 = UnaryBlockEPN(in_dim, mid, group_norm, bn) if in_dim != mid else nn.Identity()
        self.interso3 = KPConvInterSO3Block(block_name, mid, mid, radius, sigma, group_norm, config)
        self.norm = GroupNormEPN(group_norm, mid)
        self.unary2 = UnaryBlockEPN(mid, out_dim, group_norm, bn, no_relu=True)
        self.skip_conv = UnaryBlockEPN(in_dim, out_dim, group_norm, bn, no_relu=True) if in_dim != out_dim else nn.Identity()

    def _recompute_pays(self, rows):
        """The recomputed tail runs the GEMMs of unary2 (and skip_conv) twice and saves two passes over the block's output: a gain while the
        products are short (measured at the bench shapes, tools/micro/block_tail.py -> profiles/r04_block_tail.txt: mid + shortcut input
        width <= 192 -- except the 128 -> 512 layers with an identity shortcut, which gain only over >= 100 000 rows: 0.272 -> 0.227 ms at
        128 466 rows, 0.072 -> 0.074 ms at 33 036), a loss on the wide coarse layers, where the second GEMM costs more than the two passes."""
        k = self.unary2.in_dim + (0 if isinstance(self.skip_conv, nn.Identity) else self.skip_conv.in_dim)
        return k <= 192 and (k != 128 or rows >= 100000)

    def _forward_pending(self, x, q_pts, s_pts, neighb_inds):
        """Inference (blocks_epn.py:798-852 with the norms in pending form): every dense layer takes the statistics of the norm behind it
        from its accumulators and applies the norm in front of it while loading; activations are written once, raw, and only the
        convolution's input and the block's output are made concrete."""
        skip = x
        if not isinstance(self.unary1, nn.Identity):
            conv = self.interso3.conv                                           # the convolution gathers rows: concrete, in ITS gather layout
            x = SF.norm_apply(self.unary1.pending(x), blocked=True,
                              union=SF.kpconv_takes_union(q_pts, s_pts, conv.in_channels, conv.out_channels))
        y = self.interso3.norm.pending(self.interso3.conv(q_pts, s_pts, neighb_inds, x), 0.1)
        y = self.norm.pending(y, 0.1)                                         # norm of the activated norm: a second statistics pass
        if 'strided' in self.block_name:
            skip = SF.neighbor_max_pool(skip, neighb_inds)
        if SF.RECOMPUTE_TAIL and self.unary2.no_relu and self._recompute_pays(y.raw.numel() // y.raw.shape[-1]) and (
                isinstance(self.skip_conv, nn.Identity) or (self.skip_conv.no_relu and SF._ops.norm_weight_nonzero(self.skip_conv.norm.norm.weight))):
            # unary2 (mid -> 4 mid channels) and skip_conv as statistics-only GEMMs, then ONE kernel that runs both products again and writes
            # lrelu(norm(unary2) + shortcut): nothing of the output's width exists but the output (csrc/dense_norm.hip, round 4)
            aff2 = self.unary2.stats(y)
            if isinstance(self.skip_conv, nn.Identity):
                return SF.dense_residual(y, self.unary2.mlp.weight, aff2, residual=skip, final_slope=0.1)
            skip = skip.contiguous()
            return SF.dense_residual(y, self.unary2.mlp.weight, aff2, shortcut=(skip, self.skip_conv.mlp.weight, self.skip_conv.stats(skip)),
                                     final_slope=0.1)
        if not isinstance(self.skip_conv, nn.Identity):
            skip = self.skip_conv.pending(skip)
        return SF.norm_apply(self.unary2.pending(y), residual=skip, final_slope=0.1)

    def forward(self, x, q_pts, s_pts, neighb_inds):
        if (SF.PENDING_NORM and not SF.AG.needs_grad(x, self.interso3.conv.weights, self.norm.norm.weight) and self.unary2.pending_ok(x.new_empty((0, self.unary2.in_dim)))
                and all(isinstance(u, nn.Identity) or u.pending_ok(x) for u in (self.unary1, self.skip_conv))
                and self.unary2.in_dim in (16, 32, 64, 128, 256, 512, 1024)):
            return self._forward_pending(x, q_pts, s_pts, neighb_inds)
        skip = x
        x = self.unary1(x)
        x = self.interso3(x, q_pts, s_pts, neighb_inds)
        x = self.norm(x, leaky_slope=0.1)
        if 'strided' in self.block_name:
            skip = SF.neighbor_max_pool(skip, neighb_inds)
        skip = self.skip_conv(skip)
        return self.unary2(x, residual=skip, final_slope=0.1)       # lrelu(norm(mlp(x)) + shortcut)


class InvOutBlockEPN(nn.Module):
    """Equivariant -> invariant by max over the anchor axis (the attention-pooling variants are not used by SE3ET)."""

    def __init__(self, block_name, in_dim, config):
        super().__init__()
        if config.att_pooling or config.att_permute:
            raise NotImplementedError('InvOutBlockEPN (HIP): attention pooling is not part of the SE3ET hot path')
        self.block_name, self.in_dim = block_name, in_dim

    def forward(self, x, q_pts=None, s_pts=None, neighb_inds=None):
        return SF.anchor_max(x)


class LiftBlockEPN(nn.Module):
    def __init__(self, block_name, in_dim, config):
        super().__init__()
        self.block_name, self.in_dim, self.kanchor = block_name, in_dim, config.kanchor

    def forward(self, x):
        return x.unsqueeze(1).expand(-1, self.kanchor, -1)
