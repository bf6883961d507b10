"""Transformer layers of the SE3ET coarse matcher on HIP kernels.

Mirror (names, constructor arguments, forward signatures, parameter names) of the classes the SE3ET configs use from
geotransformer.modules.transformer:
    SinusoidalPositionalEmbedding   positional_embedding.py:8-34
    RPEMultiHeadAttention / RPEAttentionLayer / RPETransformerLayer        rpe_transformer.py:18-194
    MultiHeadAttention / MultiHeadAttentionEQ / AttentionLayer / TransformerLayer   vanilla_transformer.py:23-85,87-870,872-946
    AttentionOutput / RotCompressOutput                                     output_layer.py:7-47
    RPEConditionalTransformer                                               conditional_transformer.py:98-390

Differences that are part of the design:
  * tensors keep the reference layout (B, [A,] N, C) with B == 1 per call (a registration pair is the unit of work);
  * the (B, [A,] H, N, M) attention-score tensors are NOT materialised: `scores` is returned as None unless the layer was
    built with `return_scores=True` (the reference returns them, the SE3ET model never reads them);
  * key masks (`memory_masks` / `key_masks`, True = masked, the reference's -inf path) are honoured on the invariant RPE layers and on
    plain cross attention by dropping the masked keys in front of the kernels;
  * options the SE3ET experiments never enable (dropout, key weights, attention factors / attention masks, '*_best' modes,
    rotation supervision, alternative_impl) raise NotImplementedError instead of silently taking another path.
"""
import math

import os
import numpy as np
import torch
import torch.nn as nn

from ... import functional as SF
from ... import tables


def _no(value, what):
    if value is not None:
        raise NotImplementedError('%s is not supported by the HIP SE3ET path' % what)


def _one(x, what):
    if x.shape[0] != 1:
        raise NotImplementedError('%s: batch size must be 1 (got %d)' % (what, x.shape[0]))
    return x.squeeze(0)          # (a view in both directions: x[0] costs the backward a zero fill and a copy per use, ~140 launches per training step)


class SinusoidalPositionalEmbedding(nn.Module):
    def __init__(self, d_model):
        super().__init__()
        if d_model % 2 != 0:
            raise ValueError(f'Sinusoidal positional encoding with odd d_model: {d_model}')
        self.d_model = d_model
        self.register_buffer('div_term', torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model)))

    def forward(self, emb_indices):
        return SF.sinusoidal_embedding(emb_indices, self.div_term)


class AttentionOutput(nn.Module):
    """LN(x + squeeze(relu(expand(x))))."""

    def __init__(self, d_model, dropout=None, activation_fn='ReLU'):
        super().__init__()
        _no(dropout, 'dropout')
        if activation_fn != 'ReLU':
            raise NotImplementedError('activation_fn %s' % activation_fn)
        self.expand = nn.Linear(d_model, d_model * 2)
        self.squeeze = nn.Linear(d_model * 2, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, input_states):
        # the squeeze bias is applied inside the add+LayerNorm kernel (bias-free GEMMs take the cheap library path)
        hidden = SF.linear(SF.linear(input_states, self.expand.weight, self.expand.bias, relu=True), self.squeeze.weight)
        return SF.add_layer_norm(hidden, input_states, self.norm.weight, self.norm.bias, self.norm.eps,
                                 hidden_bias=self.squeeze.bias)


class RotCompressOutput(nn.Module):
    """(B, A, N, C) -> (B, N, C): LN(max_a x + squeeze(relu(expand(concat_a x))))."""

    def __init__(self, d_model, dropout=None, activation_fn='ReLU', na=12, dual_align=False):
        super().__init__()
        _no(dropout, 'dropout')
        if dual_align:
            raise NotImplementedError('dual_align')
        self.na = na
        self.expand = nn.Linear(d_model * na, d_model * 2)
        self.squeeze = nn.Linear(d_model * 2, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, input_states):
        b, a, n, c = input_states.shape
        mx = SF.anchor_max(input_states, dim=1)
        flat = input_states.permute(0, 2, 1, 3).reshape(b, n, a * c)
        hidden = SF.linear(SF.linear(flat, self.expand.weight, self.expand.bias, relu=True), self.squeeze.weight)
        return SF.add_layer_norm(hidden, mx, self.norm.weight, self.norm.bias, self.norm.eps, hidden_bias=self.squeeze.bias)


class RPEMultiHeadAttention(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None, equivariant=False, d_equiv_embed=0, return_scores=False):
        super().__init__()
        if d_model % num_heads != 0:
            raise ValueError('`d_model` ({}) must be a multiple of `num_heads` ({}).'.format(d_model, num_heads))
        _no(dropout, 'dropout')
        self.d_model, self.num_heads, self.d_model_per_head = d_model, num_heads, d_model // num_heads
        self.equivariant, self.d_equiv_embed, self.return_scores = equivariant, d_equiv_embed, return_scores
        self.proj_q = nn.Linear(d_model, d_model)
        self.proj_k = nn.Linear(d_model, d_model)
        self.proj_v = nn.Linear(d_model, d_model)
        self.proj_p = nn.Linear(d_model, d_model)
        if equivariant and d_equiv_embed > 0:
            self.proj_eq = nn.Linear(d_equiv_embed, d_model)

    def stacked_projection(self):
        """(weight, bias, column offsets) of the fused [q | k | W_p^T q | W_eq^T q] projection, cached per weight version."""
        params = [self.proj_q.weight, self.proj_q.bias, self.proj_k.weight, self.proj_k.bias, self.proj_p.weight]
        use_eq = self.equivariant and self.d_equiv_embed > 0
        if use_eq:
            params.append(self.proj_eq.weight)
        key = tuple((p.data_ptr(), p._version) for p in params)
        cache = getattr(self, '_stack_cache', None)
        if cache is None or cache[0] != key:
            w, b, offs = SF.compose_self_attention_weights(self.proj_q.weight, self.proj_q.bias, self.proj_k.weight,
                                                           self.proj_k.bias, self.proj_p.weight,
                                                           self.proj_eq.weight if use_eq else None, self.num_heads)
            cache = (key, SF.shared_tensors(w, b), offs)          # composed on this thread's stream; other streams wait for it on the GPU
            self._stack_cache = cache
        w, b = cache[1].get()
        return w, b, cache[2]

    def forward_packed(self, x, starts, lengths, embs, eq_embs):
        """Self attention of several clouds packed row-wise in x ([A,] R, C) (see functional.pack_rows)."""
        use_eq = self.equivariant and self.d_equiv_embed > 0
        if use_eq and any(e is None for e in eq_embs):
            raise RuntimeError('Equivariant embedding required here.')
        w, b, offs = self.stacked_projection()
        return SF.rpe_self_attention_packed(x, starts, lengths, embs, eq_embs if use_eq else [None] * len(embs), w, b, offs,
                                            self.proj_v.weight, self.proj_v.bias, self.num_heads)

    def forward(self, input_q, input_k, input_v, embed_qk, key_weights=None, key_masks=None, attention_factors=None,
                embed_eq=None):
        _no(key_weights, 'key_weights'), _no(attention_factors, 'attention_factors')
        xq, xk, xv = _one(input_q, 'input_q'), _one(input_k, 'input_k'), _one(input_v, 'input_v')
        keep = None
        if key_masks is not None:
            # key_masks (B, M), True = masked: the reference fills those logits with -inf (rpe_transformer.py:114-119), i.e. the softmax
            # runs over the remaining keys -- done here by dropping the masked keys in front of the kernels.  For the equivariant layers
            # the reference's mask is mis-broadcast onto the QUERY axis (whole rows -inf -> NaN): not reproduced.
            if self.equivariant and self.d_equiv_embed > 0:
                raise NotImplementedError('key_masks on an equivariant RPE layer: the reference masks whole query rows there (NaN output)')
            keep = torch.nonzero(~_one(key_masks, 'key_masks'))[:, 0]
            num_keys = xk.shape[-2]
            xk, xv = xk.index_select(-2, keep), xv.index_select(-2, keep)
            embed_qk = embed_qk.index_select(2, keep)
        q = SF.linear(xq, self.proj_q.weight, self.proj_q.bias)
        k = SF.linear(xk, self.proj_k.weight, self.proj_k.bias)
        v = SF.project_values_transposed(xv, self.proj_v.weight, self.proj_v.bias)
        use_eq = self.equivariant and self.d_equiv_embed > 0
        if use_eq and embed_eq is None:
            raise RuntimeError('Equivariant embedding required here.')
        hidden, scores = SF.rpe_attention(q, k, v, _one(embed_qk, 'embed_qk'), self.proj_p.weight,
                                          _one(embed_eq, 'embed_eq') if use_eq else None,
                                          self.proj_eq.weight if use_eq else None, self.num_heads,
                                          return_scores=self.return_scores)
        if keep is not None and scores is not None:          # masked keys carry probability 0
            full = scores.new_zeros(scores.shape[:-1] + (num_keys,))
            scores = full.index_copy_(-1, keep, scores)
        if torch.is_grad_enabled() and hidden.requires_grad:
            # the position-projection biases are constant along the softmax axis (no effect on the output): they stay in the graph with
            # an exactly-zero gradient, as in the reference, so that the optimizer treats them the same way (weight decay)
            zero = self.proj_p.bias.sum() * 0.0
            if use_eq:
                zero = zero + self.proj_eq.bias.sum() * 0.0
            hidden = hidden + zero
        return hidden.unsqueeze(0), (scores.unsqueeze(0) if scores is not None else None)


class RPEAttentionLayer(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None, equivariant=False, d_equiv_embed=0):
        super().__init__()
        _no(dropout, 'dropout')
        self.attention = RPEMultiHeadAttention(d_model, num_heads, dropout, equivariant, d_equiv_embed)
        self.linear = nn.Linear(d_model, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, input_states, memory_states, position_states, memory_weights=None, memory_masks=None,
                attention_factors=None, equiv_states=None):
        hidden, scores = self.attention(input_states, memory_states, memory_states, position_states,
                                        key_weights=memory_weights, key_masks=memory_masks,
                                        attention_factors=attention_factors, embed_eq=equiv_states)
        hidden = SF.linear(hidden, self.linear.weight)
        return SF.add_layer_norm(hidden, input_states, self.norm.weight, self.norm.bias, self.norm.eps,
                                 hidden_bias=self.linear.bias), scores

    def forward_packed(self, x, starts, lengths, embs, eq_embs):
        hidden = self.attention.forward_packed(x, starts, lengths, embs, eq_embs)
        hidden = SF.linear(hidden, self.linear.weight)
        return SF.add_layer_norm(hidden, x, self.norm.weight, self.norm.bias, self.norm.eps, hidden_bias=self.linear.bias)


class RPETransformerLayer(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None, activation_fn='ReLU', equivariant=False, d_equiv_embed=0):
        super().__init__()
        self.attention = RPEAttentionLayer(d_model, num_heads, dropout, equivariant, d_equiv_embed)
        self.output = AttentionOutput(d_model, dropout, activation_fn)

    def forward(self, input_states, memory_states, position_states, memory_weights=None, memory_masks=None,
                attention_factors=None, equiv_states=None):
        hidden, scores = self.attention(input_states, memory_states, position_states, memory_weights, memory_masks,
                                        attention_factors, equiv_states)
        return self.output(hidden), scores

    def forward_pair(self, feats0, feats1, embeddings0, embeddings1, equiv0=None, equiv1=None):
        """Self attention of both clouds in one pass: rows packed into one tensor so that every dense layer (stacked
        q/k/folded-query projection, value projection, output linear, FFN, both LayerNorms) runs ONCE for the pair."""
        x0, x1 = _one(feats0, 'feats0'), _one(feats1, 'feats1')
        x, starts = SF.pack_rows([x0, x1])
        lengths = [x0.shape[-2], x1.shape[-2]]
        eqs = [_one(equiv0, 'equiv0') if equiv0 is not None else None, _one(equiv1, 'equiv1') if equiv1 is not None else None]
        y = self.attention.forward_packed(x, starts, lengths, [_one(embeddings0, 'emb0'), _one(embeddings1, 'emb1')], eqs)
        y = self.output(y)
        return (y[..., starts[0]:starts[0] + lengths[0], :].unsqueeze(0),
                y[..., starts[1]:starts[1] + lengths[1], :].unsqueeze(0))


class MultiHeadAttention(nn.Module):
    """Plain cross attention; 4-D `input_v` (B, A, M, C) applies the invariant scores to per-anchor values."""

    def __init__(self, d_model, num_heads, dropout=None):
        super().__init__()
        if d_model % num_heads != 0:
            raise ValueError('`d_model` ({}) must be a multiple of `num_heads` ({}).'.format(d_model, num_heads))
        _no(dropout, 'dropout')
        self.d_model, self.num_heads, self.d_model_per_head = d_model, num_heads, d_model // num_heads
        self.proj_q = nn.Linear(d_model, d_model)
        self.proj_k = nn.Linear(d_model, d_model)
        self.proj_v = nn.Linear(d_model, d_model)

    def forward(self, input_q, input_k, input_v, key_weights=None, key_masks=None, attention_factors=None,
                attention_masks=None, gt_indices=None, gt_overlap=None):
        _no(key_weights, 'key_weights'), _no(attention_factors, 'attention_factors')
        _no(attention_masks, 'attention_masks')
        xq, xk, xv = _one(input_q, 'input_q'), _one(input_k, 'input_k'), _one(input_v, 'input_v')
        if key_masks is not None:      # (B, M), True = masked -> -inf logits (vanilla_transformer.py:66-67) = softmax over the other keys
            keep = torch.nonzero(~_one(key_masks, 'key_masks'))[:, 0]
            xk, xv = xk.index_select(-2, keep), xv.index_select(-2, keep)
        if key_masks is None and (xq is xk or (xq.data_ptr() == xk.data_ptr() and xq.shape == xk.shape)):      # self attention: one stacked GEMM
            q, k = SF.project_qk(xq, self.proj_q.weight, self.proj_q.bias, self.proj_k.weight, self.proj_k.bias)
        else:
            q = SF.linear(xq, self.proj_q.weight, self.proj_q.bias)
            k = SF.linear(xk, self.proj_k.weight, self.proj_k.bias)
        v = SF.project_values_transposed(xv, self.proj_v.weight, self.proj_v.bias)
        hidden = SF.cross_attention(q, k, v, self.num_heads)
        return hidden.unsqueeze(0), None


class MultiHeadAttentionEQ(nn.Module):
    """Anchor-equivariant cross attention, attn_mode 'a_soft' or 'r_soft' (global weights 'sq', mean pooling).

    out[a] = sum_e W[a,e] softmax_m(q_a.k_e / sqrt(d)) v_e with W = g / sum_e g ('a_soft', g[a,e] = mean_{n,m} (mean_h S)^2)
    or W[a,e] = sum_{r: trace[r,a]=e} w[r], w[r] ~ mean_a g[a, trace[r,a]] ('r_soft': the reference's sum over the 24
    rotations collapsed onto the (A, A) anchor pairs)."""

    def __init__(self, d_model, num_heads, dropout=None, attn_mode=None, alternative_impl=False, kanchor=4,
                 attn_r_positive='sq', attn_r_positive_rot_supervise='sigmoid'):
        super().__init__()
        if d_model % num_heads != 0:
            raise ValueError('`d_model` ({}) must be a multiple of `num_heads` ({}).'.format(d_model, num_heads))
        _no(dropout, 'dropout')
        if attn_mode not in ('a_soft', 'r_soft') or kanchor != 6 or attn_r_positive != 'sq' or alternative_impl:
            raise NotImplementedError('MultiHeadAttentionEQ (HIP): attn_mode in {a_soft, r_soft}, kanchor=6, '
                                      "attn_r_positive='sq' only")
        self.d_model, self.num_heads, self.d_model_per_head = d_model, num_heads, d_model // num_heads
        self.attn_mode, self.kanchor, self.attn_r_multihead = attn_mode, kanchor, False
        self.proj_q = nn.Linear(d_model, d_model)
        self.proj_k = nn.Linear(d_model, d_model)
        self.proj_v = nn.Linear(d_model, d_model)
        self.anchors = nn.Parameter(torch.from_numpy(tables.rotations()), requires_grad=False)
        ori, rot = tables.trace_indices()
        self.trace_idx_ori = nn.Parameter(torch.from_numpy(ori), requires_grad=False)
        self.trace_idx_rot = nn.Parameter(torch.from_numpy(rot), requires_grad=False)
        self.nr, self.na = ori.shape

    def forward(self, input_q, input_k, input_v, key_weights=None, key_masks=None, attention_factors=None,
                attention_masks=None, gt_indices=None, gt_overlap=None):
        _no(key_weights, 'key_weights'), _no(key_masks, 'key_masks'), _no(attention_factors, 'attention_factors')
        _no(attention_masks, 'attention_masks')
        xq, xk, xv = _one(input_q, 'input_q'), _one(input_k, 'input_k'), _one(input_v, 'input_v')
        if xq is xk or (xq.data_ptr() == xk.data_ptr() and xq.shape == xk.shape):      # self attention: one stacked GEMM
            q, k = SF.project_qk(xq, self.proj_q.weight, self.proj_q.bias, self.proj_k.weight, self.proj_k.bias)
        else:
            q = SF.linear(xq, self.proj_q.weight, self.proj_q.bias)
            k = SF.linear(xk, self.proj_k.weight, self.proj_k.bias)
        v = SF.project_values_transposed(xv, self.proj_v.weight, self.proj_v.bias)
        hidden, w, mix = SF.cross_attention_eq(q, k, v, self.num_heads, self.attn_mode, self.trace_idx_ori)
        if self.attn_mode == 'a_soft':
            return hidden.unsqueeze(0), [None, w.reshape(1, self.na, self.na, 1, 1, 1)]
        # reference returns [scores, attn_r (b,r,1,1,1,1), attn_matrix, q_inv] (the last two only feed rotation supervision);
        # a fifth entry carries the rotation weights collapsed onto anchor pairs for eq2inv_soft
        return hidden.unsqueeze(0), [None, w.reshape(1, self.nr, 1, 1, 1, 1), None, None, mix]


class AttentionLayer(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None, equivariant=False, attn_mode=None, alternative_impl=False,
                 kanchor=4, attn_r_positive='sq', attn_r_positive_rot_supervise='sigmoid'):
        super().__init__()
        _no(dropout, 'dropout')
        self.equivariant = equivariant
        if equivariant:
            self.attention = MultiHeadAttentionEQ(d_model, num_heads, dropout, attn_mode, alternative_impl, kanchor,
                                                  attn_r_positive, attn_r_positive_rot_supervise)
        else:
            self.attention = MultiHeadAttention(d_model, num_heads, dropout)
        self.linear = nn.Linear(d_model, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, input_states, memory_states, value_states=None, memory_weights=None, memory_masks=None,
                attention_factors=None, attention_masks=None, gt_indices=None, gt_overlap=None):
        if value_states is None:
            value_states = memory_states
        hidden, scores = self.attention(input_states, memory_states, value_states, key_weights=memory_weights,
                                        key_masks=memory_masks, attention_factors=attention_factors,
                                        attention_masks=attention_masks, gt_indices=gt_indices, gt_overlap=gt_overlap)
        hidden = SF.linear(hidden, self.linear.weight)
        if hidden.dim() == input_states.dim() + 1:           # (B, A, N, C) hidden on a (B, N, C) query: broadcast residual
            input_states = input_states.unsqueeze(1)
        return SF.add_layer_norm(hidden, input_states, self.norm.weight, self.norm.bias, self.norm.eps,
                                 hidden_bias=self.linear.bias), scores


class TransformerLayer(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None, activation_fn='ReLU', equivariant=False, attn_mode=None,
                 alternative_impl=False, kanchor=4, attn_r_positive='sq', attn_r_positive_rot_supervise='sigmoid'):
        super().__init__()
        self.equivariant = equivariant
        self.attention = AttentionLayer(d_model, num_heads, dropout, equivariant, attn_mode, alternative_impl, kanchor,
                                        attn_r_positive, attn_r_positive_rot_supervise)
        self.output = AttentionOutput(d_model, dropout, activation_fn)

    def forward(self, input_states, memory_states, value_states=None, memory_weights=None, memory_masks=None,
                attention_factors=None, attention_masks=None, gt_indices=None, gt_overlap=None):
        hidden, scores = self.attention(input_states, memory_states, value_states, memory_weights, memory_masks,
                                        attention_factors, attention_masks, gt_indices, gt_overlap)
        return self.output(hidden), scores


def _block_is_eq(block):
    return 'eq' in block or 'soft' in block or 'best' in block


class RPEConditionalTransformer(nn.Module):
    """Block scheduler: 'self' / 'self_eq' (RPE self attention on each cloud), 'cross' (plain cross attention, ref<-src
    then src<-updated ref), 'cross_a_soft' / 'cross_r_soft' (anchor-equivariant cross attention)."""

    VALID = ('self', 'self_eq', 'cross', 'cross_a_soft', 'cross_r_soft')

    def __init__(self, blocks, d_model, num_heads, dropout=None, activation_fn='ReLU', return_attention_scores=False,
                 return_attention_weights=False, anchor_matching=False, parallel=False, na=4, attn_r_positive='sq',
                 attn_r_positive_rot_supervise='sigmoid', align_mode='0', alternative_impl=False, d_equiv_embed=0):
        super().__init__()
        if return_attention_scores or return_attention_weights or anchor_matching or parallel or align_mode != '0':
            raise NotImplementedError('RPEConditionalTransformer (HIP): rotation supervision / anchor matching / parallel / '
                                      "align_mode != '0' are outside the SE3ET inference path")
        self.blocks, self.align_mode, self.d_equiv_embed = list(blocks), align_mode, d_equiv_embed
        layers = []
        for block in self.blocks:
            if block not in self.VALID:
                raise ValueError('Unsupported block type "{}".'.format(block))
            if 'self' in block:
                layers.append(RPETransformerLayer(d_model, num_heads, dropout, activation_fn, equivariant=block == 'self_eq',
                                                  d_equiv_embed=d_equiv_embed))
            else:
                eq = block != 'cross'
                layers.append(TransformerLayer(d_model, num_heads, dropout, activation_fn, equivariant=eq,
                                               attn_mode=block[len('cross_'):] if eq else None,
                                               alternative_impl=alternative_impl, kanchor=na if eq else 4,
                                               attn_r_positive=attn_r_positive if eq else 'sq',
                                               attn_r_positive_rot_supervise=attn_r_positive_rot_supervise))
        self.layers = nn.ModuleList(layers)
        if 'cross_r_soft' in self.blocks:
            self.rotcompress = RotCompressOutput(d_model, dropout, activation_fn, na=na)
        # optional callable(block index, output of the layer's FIRST call = ref direction); what a forward hook on
        # `layers[i]` sees in the reference (self layers run both clouds in one packed pass here, so hooks cannot)
        self.layer_tap = None

    def eq2inv_soft(self, feats0, feats1, mix0):
        """feats1 re-expressed in the frame that the ref<-src rotation weights prefer (sum_r w0[r] feats1[trace[r, a]] =
        sum_e mix0[a, e] feats1[e]), then both compressed over anchors."""
        return self.rotcompress(feats0), self.rotcompress(SF.rotation_weighted_permute(feats1, mix0))

    def forward(self, feats0, feats1, embeddings0, embeddings1, masks0=None, masks1=None, gt_indices=None,
                gt_overlap=None, equiv_embed0=None, equiv_embed1=None, ref_normal=None, src_normal=None):
        _no(masks0, 'masks0'), _no(masks1, 'masks1'), _no(ref_normal, 'ref_normal'), _no(src_normal, 'src_normal')
        feats0_eq = feats1_eq = None
        nb = len(self.blocks)
        for i, block in enumerate(self.blocks):
            layer = self.layers[i]
            nxt = self.blocks[i + 1] if i + 1 < nb else None
            if 'self' in block:
                src0, src1 = (feats0_eq, feats1_eq) if feats0_eq is not None and feats1_eq is not None else (feats0, feats1)
                eq = block == 'self_eq'
                if torch.is_grad_enabled() and (src0.requires_grad or src1.requires_grad):
                    # training: one cloud per call through the ops that carry a backward (se3et_amd.autograd)
                    feats0, _ = layer(src0, src0, embeddings0, equiv_states=equiv_embed0 if eq else None)
                    feats1, _ = layer(src1, src1, embeddings1, equiv_states=equiv_embed1 if eq else None)
                else:
                    feats0, feats1 = layer.forward_pair(src0, src1, embeddings0, embeddings1, equiv_embed0 if eq else None,
                                                        equiv_embed1 if eq else None)
                if self.layer_tap is not None:
                    self.layer_tap(i, feats0)
                if eq and nxt == 'cross':
                    feats0_eq, feats1_eq = feats0, feats1
                    feats0, feats1 = SF.anchor_max(feats0_eq, dim=1), SF.anchor_max(feats1_eq, dim=1)
            elif block == 'cross':
                if nxt == 'self_eq' or (nxt is None and self.blocks[i - 1] == 'self_eq'):
                    feats0_eq, _ = layer(feats0, feats1, feats1_eq)
                    if self.layer_tap is not None:
                        self.layer_tap(i, feats0_eq)
                    feats0 = SF.anchor_max(feats0_eq, dim=1)
                    feats1_eq, _ = layer(feats1, feats0, feats0_eq)
                    feats1 = SF.anchor_max(feats1_eq, dim=1)
                else:
                    feats0, _ = layer(feats0, feats1)
                    if self.layer_tap is not None:
                        self.layer_tap(i, feats0)
                    feats1, _ = layer(feats1, feats0)
            else:
                feats0, s0 = layer(feats0, feats1)
                if self.layer_tap is not None:
                    self.layer_tap(i, feats0)
                feats1, s1 = layer(feats1, feats0)
                if block == 'cross_r_soft' and nxt is not None and not _block_is_eq(nxt):
                    feats0_eq = feats1_eq = None
                    feats0, feats1 = self.eq2inv_soft(feats0, feats1, s0[4])
        return feats0, feats1
