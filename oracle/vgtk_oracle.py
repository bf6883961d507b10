"""TEST INFRASTRUCTURE (oracle): CPU restatements of the EPN toolkit's CUDA kernels that have no PyTorch twin in the reference.

Only tests/ may import this.  numpy loops, small cases only.  The kernels that DO have an importable twin
(vgtk/spconv/functional.py:373-399 inter_zpconv_grouping_naive; gather = torch.gather) are pinned by fixtures generated from the
genuine reference (tests/golden/vgtk_ops.npz); the three below restate the .cu sources -- the reference holds no test for them:
'parity unpinned' beyond the sources for ball_query, furthest_point_sampling, the intra convolution and the two anchor queries."""
import math

import numpy as np


def ball_query(new_xyz, xyz, radius, nsample):
    """grouping_cuda_kernel.cu:52-99 with idx allocated as zeros (grouping_cuda.cpp:80-82).  new_xyz (b, 3, m), xyz (b, 3, n)."""
    b, _, m = new_xyz.shape
    n = xyz.shape[2]
    idx = np.zeros((b, m, nsample), np.int32)
    r2 = np.float32(radius) * np.float32(radius)
    for bi in range(b):
        for j in range(m):
            d = new_xyz[bi, :, j:j + 1].astype(np.float32) - xyz[bi].astype(np.float32)
            d2 = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
            hits = np.nonzero(d2 < r2)[0][:nsample]
            cnt = len(hits)
            idx[bi, j, :cnt] = hits
            if cnt < nsample - 1:
                for k in range(nsample - cnt):
                    idx[bi, j, k + cnt] = idx[bi, j, k]
    return idx


def furthest_point_sampling(dataset, m):
    """grouping_cuda_kernel.cu:337-452 including the block-reduction tie order (thread t of bs = 2^floor(log2 n) <= 1024 threads scans
    k = t, t + bs, ... keeping its first maximum; the tree keeps the lower thread on equal distances).  dataset (b, 3, n)."""
    b, _, n = dataset.shape
    bs = max(min(1 << int(math.log(float(n)) / math.log(2.0)), 1024), 1)
    out = np.zeros((b, m), np.int32)
    x = dataset.astype(np.float32)
    for bi in range(b):
        temp = np.full(n, 1e10, np.float32)
        selectable = (x[bi, 0] * x[bi, 0] + x[bi, 1] * x[bi, 1] + x[bi, 2] * x[bi, 2]) > np.float32(1e-3)
        old = 0
        for j in range(1, m):
            d = x[bi] - x[bi][:, old:old + 1]
            d2 = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
            temp = np.where(selectable, np.minimum(d2, temp), temp)
            best = (-1.0, 0, 0)
            cand = np.nonzero(selectable)[0]
            if len(cand):
                top = temp[cand].max()
                ties = cand[temp[cand] == top]
                k = min(ties, key=lambda kk: (kk % bs, kk))
                best = (float(top), int(k % bs), int(k))
            old = best[2] if best[0] >= 0 else 0
            out[bi, j] = old
    return out


def intra_zpconv(nbr, w, feats):
    """zpconv_cuda_kernel.cu:119-156: out[b, c, k, p, a] = sum_n w[a, k, n] feats[b, c, p, nbr[a, n]]."""
    g = feats[:, :, :, nbr]                                     # (b, c, p, a, n)
    return np.einsum('bcpan,akn->bckpa', g, w)


def anchor_query(grouped_xyz, anchors, kernel_points):
    """grouping_cuda_kernel.cu:166-233: grouped_xyz (b, 3, p, nn) -> (b, p, na, ks, nn), float32 arithmetic."""
    g = grouped_xyz.astype(np.float32)
    x, y, z = g[:, 0], g[:, 1], g[:, 2]                                    # (b, p, nn)
    norm = (np.sqrt(x * x + y * y + z * z) + np.float32(1e-6)).astype(np.float32)
    a = anchors.astype(np.float32)
    dot = x[:, :, None, :] * a[None, None, :, 0, None] + y[:, :, None, :] * a[None, None, :, 1, None] + z[:, :, None, :] * a[None, None, :, 2, None]
    with np.errstate(invalid='ignore'):
        theta = np.arccos((dot / norm[:, :, None, :]).astype(np.float32)).astype(np.float32)      # (b, p, na, nn)
    kw, kh = kernel_points[:, 0].astype(np.float32), kernel_points[:, 1].astype(np.float32)
    dr = kw[None, None, None, :, None] - norm[:, :, None, None, :]
    da = (kh[None, None, None, :, None] - theta[:, :, :, None, :]) * norm[:, :, None, None, :]
    return (dr * dr + da * da).astype(np.float32)


def initial_anchor_query(centers, xyz, kernel_points, radius, sigma):
    """grouping_cuda_kernel.cu:102-152: centers (b, 3, nc), xyz (m, 3), kernel_points (ks, na, 3) -> weights, counts (b, ks, nc, na);
    sums in point order, float32."""
    b, _, nc = centers.shape
    ks, na, _ = kernel_points.shape
    w = np.zeros((b, ks, nc, na), np.float32)
    n = np.zeros((b, ks, nc, na), np.float32)
    x = xyz.astype(np.float32)
    for bi in range(b):
        c = centers[bi].astype(np.float32).T                              # (nc, 3)
        for pm in range(x.shape[0]):
            d = c - x[pm]
            near = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]) <= np.float32(radius)
            for pn in np.nonzero(near)[0]:
                kp = kernel_points.astype(np.float32) + c[pn]                # (ks, na, 3)
                e = kp - x[pm]
                d2k = np.sqrt((e[..., 0] * e[..., 0] + e[..., 1] * e[..., 1]) + e[..., 2] * e[..., 2]).astype(np.float32)
                wt = (np.float32(1) - d2k * d2k / np.float32(sigma)).astype(np.float32)
                w[bi, :, pn, :] += np.where(wt > 0, wt, np.float32(0))
                n[bi, :, pn, :] += np.float32(1)
    return w, n
