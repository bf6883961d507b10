"""Mirror of geotransformer/modules/sinkhorn/learnable_sinkhorn.py:5-70."""
import torch
import torch.nn as nn

from ... import functional as SF


class LearnableLogOptimalTransport(nn.Module):
    def __init__(self, num_iterations, inf=1e12):
        super().__init__()
        self.num_iterations, self.inf = num_iterations, inf
        self.register_parameter('alpha', nn.Parameter(torch.tensor(1.0)))

    def forward(self, scores, row_masks=None, col_masks=None):
        """scores (B, M, N), masks True = valid -> log assignment matrix (B, M+1, N+1) with dustbin row/column."""
        b, m, n = scores.shape
        if row_masks is None:
            row_masks = torch.ones((b, m), dtype=torch.bool, device=scores.device)
        if col_masks is None:
            col_masks = torch.ones((b, n), dtype=torch.bool, device=scores.device)
        return SF.log_optimal_transport(scores, row_masks, col_masks, self.alpha, self.num_iterations, self.inf)

    def __repr__(self):
        return self.__class__.__name__ + '(num_iterations={})'.format(self.num_iterations)
