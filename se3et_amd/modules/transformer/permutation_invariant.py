import numpy as np
import torch
import torch.nn as nn

from ... import tables


class PermutationInvariantLayer(nn.Module):
    """The anchor-matching head the KITTI model constructs (geotransformer/modules/transformer/permutation_invariant.py:12-82;
    experiments/se3eti.kitti/model.py:87).  `anchor_matching` is off in all five SE3ET configurations: the parameters (`fc1`,
    `batch_norm`, `fc2`, the constant `anchors`) exist so that reference checkpoints load with strict=True.  `forward` restates the
    reference's live arithmetic (only `fc2` is applied there; fc1 / batch norm / dropout are commented out) in plain torch."""

    def __init__(self, na, d_model):
        super().__init__()
        if na != tables.KANCHOR:
            raise NotImplementedError('kanchor=%d is not implemented in the PermutationInvariantLayer()' % na)
        self.na = na
        self.fc1 = nn.Linear(na * d_model, na * d_model)
        self.batch_norm = nn.BatchNorm1d(na * d_model)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(p=0.2)
        self.fc2 = nn.Linear(na * d_model, d_model)
        self.anchors = nn.Parameter(torch.from_numpy(tables.rotations().astype(np.float32)), requires_grad=False)   # (24, 3, 3)
        self.trace_idx_ori = tables.trace_indices()[0]                                                              # (24, 6)

    def forward(self, ref_feats_m, src_feats_m, gt_T0):
        """(1, A, N, C), (1, A, M, C), ground-truth transform (4, 4) -> ref (A, N, C), src with its anchors permuted onto ref's under the
        group rotation nearest to the ground truth (A, M, C), and both through fc2 on the concatenated anchors: (1, N, C), (1, M, C)."""
        label = int(torch.einsum('ij,akj->aik', gt_T0[:3, :3], self.anchors).diagonal(dim1=-2, dim2=-1).sum(-1).argmax())
        order = torch.as_tensor(self.trace_idx_ori[label], dtype=torch.long, device=src_feats_m.device)
        src_matched = src_feats_m[:, order]
        cat = lambda x: x.permute(0, 2, 1, 3).reshape(x.shape[0], x.shape[2], -1)
        return ref_feats_m.squeeze(0), src_matched.squeeze(0), self.fc2(cat(ref_feats_m)), self.fc2(cat(src_matched))
