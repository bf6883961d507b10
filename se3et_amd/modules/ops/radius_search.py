from ... import ops as _ops


def radius_search(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit):
    """Stack-mode radius neighbour search on the GPU.

    Same contract as geotransformer/modules/ops/radius_search.py:7-27: (Nq, min(limit, max count)) int64, ascending
    distance, padded with the total support size.  Rows that hold EXACTLY tied distances come in the reference's order (its k-d tree walk +
    std::sort, csrc/radius_ties.hip; se3et_amd.ops.RADIUS_REFERENCE_TIES = False: index order).  One host synchronisation (the column count
    and the number of rows with ties), a second one only when such rows exist."""
    if neighbor_limit <= 0 or neighbor_limit > 64:
        raise RuntimeError('radius_search: neighbor_limit must be in [1, 64] on the HIP path')
    full, max_count = _ops.radius_search_reference_order(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit)
    width = min(int(neighbor_limit), int(max_count))
    return full if width == full.shape[1] else full[:, :width].contiguous()
