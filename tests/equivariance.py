"""Shared pieces of the equivariance self-checks (tests/test_oracle_equivariance.py on the CPU oracle, tests/test_gpu_equivariance.py
on the HIP path).

What holds, and why the check needs symmetrised weights.  E2PN features live on the 6 cosets R_a.C4 of the 24-element rotation group
(anchors = coset representatives, se3et_amd/tables.py).  Rotating both clouds by a group rotation R_g permutes the cosets, so an
equivariant network must answer with its anchor axis permuted.  The reference's KPConvInterSO3 ties its weights over the C4 orbits of
the KERNEL POINTS (blocks_epn.py:237-253, K_real = 6 slots) but indexes the INPUT ANCHOR by the coset of R_r^-1 R_a with one fixed
representative R_r per output anchor (blocks_epn.py:283-296): a rotation g maps output anchor r to r' with R_r' = g R_r s for some s in
C4, and that s rotates the four 'equatorial' input-anchor slots among themselves.  The layer is therefore exactly equivariant only for
weights that are invariant under the C4 action on the anchor slot -- a linear subspace of the reference's parameters (measured on
the oracle: 0.8-1.0 relative deviation with generic weights, 8e-7 after projecting onto that subspace).  The same holds for proj_eq
of the self_eq layers (the anchor-frame spherical harmonics D_a Y_1 are defined up to the stabiliser rotation about the anchor axis:
only Y_0 and the axial component are invariant) and for RotCompressOutput (it concatenates the anchors in a fixed order).
`symmetrise_state` is that projection.  With it, for every one of the 24 group rotations:
    equivariant tensors (backbone feats_c, outputs of self_eq / cross_a_soft / cross_r_soft layers)  ->  anchor axis permuted by
                                                                                         trace_idx_rot[g]
    invariant tensors (fine features, outputs of the invariant layers, final superpoint features)   ->  unchanged
This pins what no fixture can (SURVEY 8c, e3nn unavailable): the l = 1 spherical-harmonics component order / sign convention against
the Wigner-D tables (an inconsistent pair breaks the self_eq layers' permutation property), the anchor tables against the kernel-slot
tables, and the 24 -> 6x6 collapse of cross_r_soft against trace_idx_ori."""
import numpy as np
import torch


def slot_action():
    """perm[s][j] = coset index of (s R_j) for the 4 stabiliser rotations s (se3et_amd.tables.quotient_anchors)."""
    from se3et_amd import tables
    anchors, stab = tables.anchors(), tables.quotient_anchors()

    def coset(R):
        for j in range(anchors.shape[0]):
            M = anchors[j].T @ R
            if any(np.abs(M - q).max() < 1e-4 for q in stab):
                return j
        raise RuntimeError('rotation outside the group')

    return [[coset(s @ anchors[j]) for j in range(anchors.shape[0])] for s in stab]


def symmetrise_state(state, num_anchors=6):
    """Projection of a reference-layout state dict onto the exactly-equivariant subspace (see the module docstring)."""
    perms = slot_action()
    out = {}
    for k, v in state.items():
        if k.endswith('interso3.conv.weights'):                       # (K_real, A slots, Cin, Cout)
            v = torch.stack([v[:, p] for p in perms]).mean(0)
        elif k.endswith('proj_eq.weight'):                            # (C, 4): [Y0, x, y, z] in the anchor frame; axis = z
            v = v.clone()
            v[:, 1:3] = 0
        elif k.endswith('rotcompress.expand.weight'):                 # (2C, A * C): anchor blocks in concatenation order
            o, ac = v.shape
            v = v.view(o, num_anchors, ac // num_anchors).mean(1, keepdim=True).expand(o, num_anchors, ac // num_anchors).reshape(o, ac)
        out[k] = v.contiguous()
    return out


def rotated_data(data, R):
    """The same pyramid (identical index tables) with every stage's points rotated: p -> R p.  (Grid subsampling itself is not
    rotation equivariant -- voxels are axis aligned with a data-dependent origin --, so the property is stated on a fixed pyramid.)"""
    d = dict(data)
    d['points'] = [(p @ R.t().to(p)).contiguous() for p in data['points']]
    return d


def group():
    """(rotations (24, 3, 3) float32, anchor permutation per rotation (24, 6))."""
    from se3et_amd import tables
    _, rot = tables.trace_indices()
    return torch.from_numpy(tables.rotations()).float(), torch.from_numpy(rot).long()
