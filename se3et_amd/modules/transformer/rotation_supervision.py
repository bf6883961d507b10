import torch
import torch.nn as nn
import torch.nn.functional as F


class RotationAttentionLayer(nn.Module):
    """Anchor-to-anchor similarity of matched superpoints, the rotation-supervision head the KITTI model constructs
    (geotransformer/modules/transformer/rotation_supervision.py:6-45; experiments/se3eti.kitti/model.py:82-85).  `supervise_rotation` is
    off in all five SE3ET configurations, so the forward never reaches it: the parameters (`proj_q`, `proj_k`) exist so that reference
    checkpoints load with strict=True, and `forward` is the same arithmetic in plain torch (off the hot path, no kernel)."""

    def __init__(self, d_model, num_heads):
        super().__init__()
        if d_model % num_heads != 0:
            raise ValueError('`d_model` ({}) must be a multiple of `num_heads` ({}).'.format(d_model, num_heads))
        self.d_model, self.num_heads, self.d_model_per_head = d_model, num_heads, d_model // num_heads
        self.proj_q = nn.Linear(d_model, d_model)
        self.proj_k = nn.Linear(d_model, d_model)

    def forward(self, ref_feats_m, src_feats_m, ref_node_corr_indices, src_node_corr_indices):
        """(B, A, N, C), (B, A, M, C), matched indices (n,), (n,) -> (B, A, A) in [0, 1]: per head, the cosine between the matched rows of
        ref anchor a and src anchor e taken as ONE (n * c)-vector, averaged over heads, mapped from [-1, 1]."""
        def heads(x, idx):
            b, a = x.shape[:2]
            x = x[:, :, idx].reshape(b, a, idx.shape[0], self.num_heads, self.d_model_per_head).permute(0, 1, 3, 2, 4)
            return F.normalize(x.reshape(b, a, self.num_heads, -1), dim=-1)
        q = heads(self.proj_q(ref_feats_m), ref_node_corr_indices)
        k = heads(self.proj_k(src_feats_m), src_node_corr_indices)
        return (torch.einsum('bahx,behx->baeh', q, k).mean(3) + 1) / 2
