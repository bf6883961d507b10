"""Training step of the SE3ET hot path (BASELINE.json configs[4]).

What the reference spreads over experiments/se3ete.3dmatch/{model.py:110-131,172-178, loss.py:15-76,163-198},
geotransformer/modules/registration/matching.py:231-315, modules/geotransformer/superpoint_target.py:6-41,
modules/loss/circle_loss.py:44-86 and engine/base_trainer.py:66-78,181-196, restated for the build's own driver:

  * ground-truth superpoint correspondences (patch overlaps under the ground-truth transform) and the random target selection;
  * OverallLoss = weighted circle loss on the superpoint features + negative log-likelihood of the optimal-transport matrix on the
    ground-truth point correspondences of the selected patches;
  * `forward_train`: the model forward with autograd -- every HIP op runs its gfx950 kernel forward and differentiates a PyTorch
    restatement on the GPU in backward (se3et_amd/autograd.py);
  * `train_step`: forward, loss, backward, Adam;  `distributed_model`: one process per GPU, DistributedDataParallel over RCCL
    (backend 'nccl' on ROCm; 'gloo' in the CPU tests), gradient all-reduce in DDP's buckets, lr x world size as the reference.

Plain torch apart from the model forward: none of this is a hot kernel."""
import numpy as np
import torch
import torch.nn.functional as F

from . import functional as SF
from .modules.ops import pairwise_distance


# ---------------------------------------------------------------------------------------------------------------------------------------
# ground truth
# ---------------------------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def node_correspondences(ref_nodes, src_nodes, ref_knn_points, src_knn_points, transform, pos_radius, ref_masks, src_masks,
                         ref_knn_masks, src_knn_masks):
    """registration/matching.py:231-315 -> (corr_indices (C, 2) int64, corr_overlaps (C,)): patch pairs whose points overlap under the
    ground-truth transform, overlap = mean of the two covered fractions."""
    src_nodes = SF.apply_transform(src_nodes, transform)
    src_knn_points = SF.apply_transform(src_knn_points, transform)
    node_mask = ref_masks[:, None] & src_masks[None, :]
    ref_r = torch.linalg.norm(ref_knn_points - ref_nodes[:, None], dim=-1).masked_fill(~ref_knn_masks, 0.0).amax(1)
    src_r = torch.linalg.norm(src_knn_points - src_nodes[:, None], dim=-1).masked_fill(~src_knn_masks, 0.0).amax(1)
    dist = torch.sqrt(pairwise_distance(ref_nodes, src_nodes))
    hit = ((ref_r[:, None] + src_r[None, :] + pos_radius - dist) > 0) & node_mask            # enclosing spheres intersect
    sel_r, sel_s = torch.nonzero(hit, as_tuple=True)
    rm, sm = ref_knn_masks[sel_r], src_knn_masks[sel_s]
    d = pairwise_distance(ref_knn_points[sel_r], src_knn_points[sel_s])
    d = d.masked_fill(~(rm[:, :, None] & sm[:, None, :]), 1e12)
    close = d < pos_radius ** 2
    ref_cov = torch.count_nonzero(close.sum(-1), dim=-1).float() / rm.sum(-1).float()
    src_cov = torch.count_nonzero(close.sum(-2), dim=-1).float() / sm.sum(-1).float()
    overlaps = (ref_cov + src_cov) / 2
    keep = overlaps > 0
    return torch.stack((sel_r[keep], sel_s[keep]), 1), overlaps[keep]


@torch.no_grad()
def select_targets(gt_indices, gt_overlaps, num_targets, overlap_threshold, rng=None):
    """superpoint_target.py:6-41: the correspondences above the overlap threshold, at most `num_targets` of them (random subset drawn
    with numpy's global generator as the reference, or `rng` = a numpy Generator / RandomState)."""
    m = gt_overlaps > overlap_threshold
    idx, ov = gt_indices[m], gt_overlaps[m]
    if idx.shape[0] > num_targets:
        choice = (rng if rng is not None else np.random).choice(np.arange(idx.shape[0]), num_targets, replace=False)
        sel = torch.from_numpy(np.asarray(choice)).to(idx.device)
        idx, ov = idx[sel], ov[sel]
    return idx[:, 0], idx[:, 1], ov


# ---------------------------------------------------------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------------------------------------------------------
def weighted_circle_loss(pos_masks, neg_masks, feat_dists, pos_margin, neg_margin, pos_optimal, neg_optimal, log_scale, pos_scales=None):
    """modules/loss/circle_loss.py:44-86."""
    row_masks = ((pos_masks.sum(-1) > 0) & (neg_masks.sum(-1) > 0)).detach()
    col_masks = ((pos_masks.sum(-2) > 0) & (neg_masks.sum(-2) > 0)).detach()
    pos_w = torch.clamp(feat_dists - 1e5 * (~pos_masks).float() - pos_optimal, min=0.0)
    if pos_scales is not None:
        pos_w = pos_w * pos_scales
    pos_w = pos_w.detach()
    neg_w = torch.clamp(neg_optimal - (feat_dists + 1e5 * (~neg_masks).float()), min=0.0).detach()
    lp = log_scale * (feat_dists - pos_margin) * pos_w
    ln = log_scale * (neg_margin - feat_dists) * neg_w
    loss_row = F.softplus(torch.logsumexp(lp, -1) + torch.logsumexp(ln, -1)) / log_scale
    loss_col = F.softplus(torch.logsumexp(lp, -2) + torch.logsumexp(ln, -2)) / log_scale
    # (means over the masked entries as masked sums: a boolean-mask index is a nonzero() + a host synchronisation, forward and backward)
    zero = torch.zeros((), dtype=loss_row.dtype, device=loss_row.device)
    return (torch.where(row_masks, loss_row, zero).sum() / row_masks.sum() + torch.where(col_masks, loss_col, zero).sum() / col_masks.sum()) / 2


class OverallLoss(torch.nn.Module):
    """experiments/se3ete.3dmatch/loss.py:15-76,163-198 (rotation supervision off, as in the SE3ET configs)."""

    def __init__(self, cfg):
        super().__init__()
        c, self.positive_radius = cfg.coarse_loss, cfg.fine_loss.positive_radius
        self.circle = (c.positive_margin, c.negative_margin, c.positive_optimal, c.negative_optimal, c.log_scale)
        self.positive_overlap = c.positive_overlap
        self.weight_coarse_loss, self.weight_fine_loss = cfg.loss.weight_coarse_loss, cfg.loss.weight_fine_loss

    def coarse(self, out):
        ref, src = out['ref_feats_c'], out['src_feats_c']
        gi, go = out['gt_node_corr_indices'], out['gt_node_corr_overlaps']
        feat_dists = torch.sqrt(pairwise_distance(ref, src, normalized=True))
        overlaps = torch.zeros_like(feat_dists)
        overlaps[gi[:, 0], gi[:, 1]] = go
        pos = overlaps > self.positive_overlap
        neg = overlaps == 0
        return weighted_circle_loss(pos, neg, feat_dists, *self.circle, pos_scales=torch.sqrt(overlaps * pos.float()))

    def fine(self, out, transform):
        rp, sp = out['ref_node_corr_knn_points'], SF.apply_transform(out['src_node_corr_knn_points'], transform)
        rm, sm, scores = out['ref_node_corr_knn_masks'], out['src_node_corr_knn_masks'], out['matching_scores']
        corr = (pairwise_distance(rp, sp) < self.positive_radius ** 2) & (rm[:, :, None] & sm[:, None, :])
        labels = torch.zeros_like(scores, dtype=torch.bool)
        labels[:, :-1, :-1] = corr
        labels[:, :-1, -1] = (corr.sum(2) == 0) & rm
        labels[:, -1, :-1] = (corr.sum(1) == 0) & sm
        return -(torch.where(labels, scores, torch.zeros((), dtype=scores.dtype, device=scores.device)).sum() / labels.sum())

    def forward(self, out, data_dict):
        c, f = self.coarse(out), self.fine(out, data_dict['transform'])
        return {'loss': self.weight_coarse_loss * c + self.weight_fine_loss * f, 'c_loss': c, 'f_loss': f}


# ---------------------------------------------------------------------------------------------------------------------------------------
# step and data parallelism
# ---------------------------------------------------------------------------------------------------------------------------------------
def make_optimizer(model, cfg, world_size=1):
    """Adam with the reference's hyper-parameters; the learning rate is scaled by the world size (base_trainer.py:191-196)."""
    return torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=cfg.optim.lr * world_size,
                            weight_decay=cfg.optim.weight_decay)


def train_step(model, data_dict, loss_fn, optimizer, targets=None, rng=None):
    """One optimisation step on one pair: forward (HIP kernels), loss, backward (se3et_amd.autograd), Adam.  `model` may be the
    bare SE3ET or its DistributedDataParallel wrapper (the gradient all-reduce then overlaps backward).  Returns (losses, outputs)."""
    out = model(data_dict, train=True, targets=targets, rng=rng)
    losses = loss_fn(out, data_dict)
    optimizer.zero_grad(set_to_none=True)
    losses['loss'].backward()
    optimizer.step()
    return losses, out


def distributed_model(model, device=None):
    """DistributedDataParallel wrapper for an initialised process group (engine/base_trainer.py:181-189): one process per GPU,
    gradients averaged over ranks in DDP's buckets (RCCL all-reduce on 'nccl', gloo in the CPU tests).  Parameters that never
    receive a gradient in the SE3ET step (constant tables are requires_grad=False already) are tolerated."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    if device is not None and device.type == 'cuda':
        return DDP(model, device_ids=[device.index], find_unused_parameters=True)
    return DDP(model, find_unused_parameters=True)
