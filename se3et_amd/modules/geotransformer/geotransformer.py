import numpy as np
import torch
import torch.nn as nn

from ... import functional as SF
from ... import tables
from ..transformer import SinusoidalPositionalEmbedding, RPEConditionalTransformer


class GeometricStructureEmbedding(nn.Module):
    """E[n, m] = proj_d(sin-emb(|p_n - p_m| / sigma_d)) + max_k proj_a(sin-emb(angle_k(n, m) / sigma_a)) and, when
    n_level_equiv > 0, the anchor-rotated l <= 1 spherical harmonics of p_n - p_m (B, A, N, M, 4).
    `embedding_dtype` (attribute, default torch.float32 = the reference's arithmetic): torch.bfloat16 stores E rounded to
    bf16, which halves the dominant HBM term of every RPE self-attention call ('bf16 attention', BASELINE.json configs[2])."""

    def __init__(self, hidden_dim, sigma_d, sigma_a, angle_k, reduction_a='max', kanchor=1, n_level_equiv=0):
        super().__init__()
        if reduction_a != 'max':
            raise NotImplementedError("GeometricStructureEmbedding (HIP): reduction_a='max' only")
        if n_level_equiv not in (0, 2):
            raise NotImplementedError('GeometricStructureEmbedding (HIP): n_level_equiv must be 0 or 2')
        self.sigma_d, self.sigma_a, self.angle_k = sigma_d, sigma_a, angle_k
        self.factor_a = 180.0 / (sigma_a * np.pi)
        self.embedding = SinusoidalPositionalEmbedding(hidden_dim)
        self.proj_d = nn.Linear(hidden_dim, hidden_dim)
        self.proj_a = nn.Linear(hidden_dim, hidden_dim)
        self.n_level_equiv, self.kanchor, self.reduction_a = n_level_equiv, kanchor, reduction_a
        self.embedding_dtype = torch.float32
        if n_level_equiv > 0 and kanchor is not None and kanchor > 1:
            if kanchor != 6:
                raise NotImplementedError('kanchor=%d' % kanchor)
            self.anchors_wignerD = nn.ParameterList(
                [nn.Parameter(torch.from_numpy(t), requires_grad=False) for t in tables.wigner_tables()])

    def tables(self):
        """The tabulated proj_d / proj_a responses, validated against the current weights (once per forward is enough)."""
        from ... import ops as _ops
        return _ops.embedding_tables(self.embedding.div_term, self.proj_d.weight, self.proj_d.bias, self.proj_a.weight, self.proj_a.bias,
                                     self.sigma_a)

    def forward(self, points, tables=None):
        if points.shape[0] != 1:
            raise NotImplementedError('batch size must be 1')
        args = (points[0], self.embedding.div_term, self.proj_d.weight, self.proj_d.bias, self.proj_a.weight,
                self.proj_a.bias, self.sigma_d, self.sigma_a, self.angle_k)
        if self.n_level_equiv > 0:
            emb, eq = SF.geometric_embedding(*args, wigner_d1=self.anchors_wignerD[1], dtype=self.embedding_dtype, tables=tables)
            return emb.unsqueeze(0), eq.unsqueeze(0)
        return SF.geometric_embedding(*args, dtype=self.embedding_dtype, tables=tables).unsqueeze(0)


class GeometricTransformer(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim, num_heads, blocks, sigma_d, sigma_a, angle_k, dropout=None,
                 activation_fn='ReLU', supervise_rotation=False, anchor_matching=False, reduction_a='max', na=None,
                 attn_r_positive='sq', attn_r_positive_rot_supervise='sigmoid', align_mode='0', alternative_impl=False,
                 n_level_equiv=0):
        super().__init__()
        if supervise_rotation or anchor_matching:
            raise NotImplementedError('GeometricTransformer (HIP): rotation supervision / anchor matching')
        if na is None:
            raise NotImplementedError('GeometricTransformer (HIP): the invariant GeoTransformer baseline (na=None)')
        self.n_level_equiv = n_level_equiv
        self.d_equiv_embed = int((np.arange(n_level_equiv) * 2 + 1).sum())
        self.embedding = GeometricStructureEmbedding(hidden_dim, sigma_d, sigma_a, angle_k, reduction_a, na, n_level_equiv)
        self.in_proj = nn.Linear(input_dim, hidden_dim)
        self.na, self.supervise_rotation, self.anchor_matching = na, supervise_rotation, anchor_matching
        self.transformer = RPEConditionalTransformer(blocks, hidden_dim, num_heads, dropout, activation_fn, na=na,
                                                     attn_r_positive=attn_r_positive,
                                                     attn_r_positive_rot_supervise=attn_r_positive_rot_supervise,
                                                     align_mode=align_mode, alternative_impl=alternative_impl,
                                                     d_equiv_embed=self.d_equiv_embed)
        self.out_proj = nn.Linear(hidden_dim, output_dim)

    def forward(self, ref_points, src_points, ref_feats, src_feats, ref_masks=None, src_masks=None, gt_indices=None,
                gt_overlap=None, ref_normal=None, src_normal=None):
        """ref_feats (B, N, A, C) -> (B, N, C_out); returns the reference's 6-tuple (the last four are None)."""
        tabs = self.embedding.tables()
        if self.n_level_equiv == 0:
            ref_emb, src_emb = self.embedding(ref_points, tabs), self.embedding(src_points, tabs)
            ref_eq = src_eq = None
        else:
            ref_emb, ref_eq = self.embedding(ref_points, tabs)
            src_emb, src_eq = self.embedding(src_points, tabs)
        ref_feats = SF.linear(ref_feats.transpose(1, 2), self.in_proj.weight, self.in_proj.bias)
        src_feats = SF.linear(src_feats.transpose(1, 2), self.in_proj.weight, self.in_proj.bias)
        ref_feats, src_feats = self.transformer(ref_feats, src_feats, ref_emb, src_emb, masks0=ref_masks, masks1=src_masks,
                                                equiv_embed0=ref_eq, equiv_embed1=src_eq, ref_normal=ref_normal,
                                                src_normal=src_normal)
        ref_feats = SF.linear(ref_feats, self.out_proj.weight, self.out_proj.bias)
        src_feats = SF.linear(src_feats, self.out_proj.weight, self.out_proj.bias)
        return ref_feats, src_feats, None, None, None, None
