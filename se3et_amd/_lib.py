"""ctypes binding of libse3et_hip.so (C ABI declared in include/se3et_hip.h).

There is deliberately no fallback: if the library is missing or a kernel reports an error the caller gets a
RuntimeError -- the product never computes a hot-path op on the CPU or through eager PyTorch."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SE3_LIB: another build of the same library, for A/B runs of two kernel versions on ONE box -- tools/r5/build_ab.sh)
LIB_PATH = os.environ.get('SE3_LIB') or os.path.join(_HERE, 'csrc', 'libse3et_hip.so')

_vp, _i64, _i32, _f32, _sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/se3et_hip.h declares (tests/test_cabi.py checks)
SIGNATURES = {
    'se3_version': (ctypes.c_char_p, []),
    'se3_last_error': (ctypes.c_char_p, []),
    'se3_debug_set_bias_variant': (None, [_i32, _i32]),
    'se3_debug_set_attention_variant': (None, [_i32]),
    'se3_debug_set_kpconv_variant': (None, [_i32]),
    'se3_debug_set_kpconv_union_variant': (None, [_i32]),
    'se3_debug_set_sinkhorn_variant': (None, [_i32]),
    'se3_debug_dense_saturated_rows': (ctypes.c_uint64, [_i32]),
    'se3_debug_attention_saturated': (ctypes.c_uint64, [_i32]),
    'se3_debug_set_attention_profile': (None, [_vp]),
    'se3_debug_kernel_timing': (None, [_i32]),
    'se3_debug_kernel_timing_collect': (_i32, [_vp, _vp, _i32]),
    'se3_debug_kernel_timing_collect_ex': (_i32, [_vp, _vp, _vp, _i32]),
    'se3_radius_neighbors': (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _f32, _i32, _vp, _vp, _vp]),
    'se3_radius_grid_workspace_bytes': (_sz, [_i64, _i32]),
    'se3_radius_grid_build': (_i32, [_vp, _i64, _vp, _i32, _f32, _vp, _sz, _vp]),
    'se3_radius_neighbors_grid': (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _vp, _f32, _i32, _vp, _vp, _i32, _vp]),
    'se3_radius_neighbors_ties': (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'se3_radius_neighbors_grid_ties': (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _vp, _f32, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    'se3_kdtree_max_bytes': (_sz, [_i64, _i32]),
    'se3_kdtree_build_host': (_i32, [_vp, _i64, _vp, _i32, _vp, _sz, _vp]),
    'se3_radius_tie_scratch_bytes': (_sz, [_i64, _i32]),
    'se3_radius_neighbors_tie_order': (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp, _f32, _i32, _vp, _i64, _i32, _vp, _sz, _vp, _vp]),
    'se3_debug_std_sort_host': (_i64, [_vp, _i64, _i32]),
    'se3_debug_radius_tie_order_host': (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp, _f32, _i32, _vp, _i64, _i32, _vp, _vp]),
    'se3_grid_subsample_workspace_bytes': (_sz, [_i64, _i32]),
    'se3_grid_subsample': (_i32, [_vp, _vp, _i64, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    'se3_grid_subsample_dev': (_i32, [_vp, _vp, _i64, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    'se3_group_norm_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'se3_group_norm_fwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _i32, _f32, _vp, _vp, _sz, _vp]),
    'se3_group_norm_bwd_workspace_bytes': (_sz, [_i32]),
    'se3_group_norm_segments_bwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i32, _f32, _i32, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    'se3_group_norm_segments_fwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i32, _f32, _i32, _f32, _vp, _vp, _sz, _vp]),
    'se3_add_layer_norm_fwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _f32, _vp, _vp]),
    'se3_add_layer_norm_bwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _f32, _vp, _vp, _vp]),
    'se3_group_norm_stats_workspace_bytes': (_sz, [_i32]),
    'se3_group_norm_stats': (_i32, [_vp, _vp, _f32, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i32, _f32, _vp, _vp, _sz, _vp]),
    'se3_group_norm_apply': (_i32, [_vp, _vp, _f32, _vp, _f32, _vp, _vp, _f32, _i64, _i32, _vp, _i32, _i32, _vp, _vp]),
    'se3_dense_norm_workspace_bytes': (_sz, [_i32]),
    'se3_dense_norm_fwd': (_i32, [_vp, _i64, _i32, _vp, _f32, _vp, _f32, _vp, _i32, _vp, _vp, _vp, _i32, _f32, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    'se3_dense_residual_fwd': (_i32, [_vp, _i64, _i32, _vp, _f32, _vp, _f32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _f32, _vp, _i32, _vp, _vp]),
    'se3_linear_stream': (_i32, [_vp, _i64, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _i64, _vp]),
    'se3_linear_stream_transposed': (_i32, [_vp, _i64, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _i64, _vp]),
    'se3_patch_scores': (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _i32, _f32, _vp, _vp]),
    'se3_anchor_mix_stack': (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp]),
    'se3_linear_stream_segments': (_i32, [_vp, _i64, _i32, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _i64, _vp]),
    'se3_transformer_workspace_bytes': (_sz, [_vp]),
    'se3_transformer_plan_layout': (None, [_vp]),
    'se3_transformer_forward': (_i32, [_vp, _vp, _vp, _vp, _sz, _vp]),
    'se3_dense_norm_set_target_chunks': (None, [_i32]),
    'se3_linear_weight_pieces_bytes': (_sz, [_i32, _i32]),
    'se3_linear_split_weights_f16': (_i32, [_vp, _i32, _i32, _vp, _vp]),
    'se3_linear_f16': (_i32, [_vp, _i64, _i32, _i64, _vp, _vp, _i32, _i32, _vp, _i64, _vp]),
    'se3_gather_rows_padded': (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    'se3_neighbor_max_pool': (_i32, [_vp, _vp, _i64, _i64, _i32, _i64, _vp, _vp]),
    'se3_neighbor_max_pool_bwd': (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i64, _vp, _vp]),
    'se3_kpconv_so3_gather': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _i64, _i32, _i32, _vp, _vp]),
    'se3_kpconv_so3_gather_bwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _i64, _i32, _i32, _vp, _vp]),
    'se3_kpconv_so3_gather_bwd_fixed': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i64, _i64, _i32, _i32, _vp, _vp, _vp]),
    'se3_kpconv_fixed_to_float': (_i32, [_vp, _i64, _vp, _i64, _vp, _vp]),
    'se3_fixed_to_float': (_i32, [_vp, _i64, _vp, _i64, _i32, _vp, _vp]),
    'se3_neighbor_max_pool_bwd_fixed': (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i64, _vp, _vp, _vp]),
    'se3_scatter_add_rows_fixed': (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    'se3_add_layer_norm_bwd_blocks': (_i64, [_i64]),
    'se3_add_layer_norm_bwd_partials': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _f32, _vp, _vp, _vp]),
    'se3_kpconv_sums_bytes': (_sz, [_i64, _i32]),
    'se3_kpconv_so3_gather_sums': (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    'se3_kpconv_weight_pieces_bytes': (_sz, [_i32, _i32]),
    'se3_kpconv_split_weights_f16': (_i32, [_vp, _i32, _i32, _vp, _vp]),
    'se3_kpconv_so3_contract_f16': (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    'se3_kpconv_neighbor_table_bytes': (_sz, [_i64, _i32]),
    'se3_kpconv_neighbor_table': (_i32, [_vp, _vp, _vp, _vp, _f32, _i64, _i64, _i32, _vp, _sz, _vp]),
    'se3_kpconv_fused_split_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'se3_kpconv_so3_fused': (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _i32, _vp]),
    'se3_point_order_groups': (_i64, [_vp, _i32]),
    'se3_point_order': (_i32, [_vp, _i64, _vp, _i32, _f32, _vp, _vp]),
    'se3_point_order_stages': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    'se3_point_order_keys': (_i32, [_vp, _i64, _vp, _i32, _f32, _vp, _vp]),
    'se3_point_order_place': (_i32, [_vp, _vp, _i64, _vp, _i32, _vp, _vp]),
    'se3_kpconv_union_plan_bytes': (_sz, [_i64, _i32]),
    'se3_kpconv_union_plan': (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _sz, _vp]),
    'se3_kpconv_union_split_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'se3_kpconv_so3_union': (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _i32, _vp, _vp]),
    'se3_kpconv_so3_fused_scaled': (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _i32, _vp, _vp]),
    'se3_group_norm_apply_amax': (_i32, [_vp, _vp, _f32, _vp, _f32, _vp, _vp, _f32, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    'se3_rpe_bias_fwd': (_i32, [_vp, _vp, _i32, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'se3_attention_fwd': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _f32, _vp, _vp]),
    'se3_rpe_bias_stack_fwd': (_i32, [_vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    'se3_rpe_bias_stack_bf16_fwd': (_i32, [_vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    'se3_attention_kv_pieces_bytes': (_sz, [_i32, _i64, _i32, _i32]),
    'se3_attention_stack_fwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f32, _vp, _vp, _sz, _vp]),
    'se3_rpe_self_attention_stack_fwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    'se3_rpe_self_attention_stack_bf16_fwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _sz, _vp]),
    'se3_cross_eq_stats': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    'se3_cross_eq_mix': (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    'se3_cross_eq_x6_workspace_bytes': (_sz, [_i32, _i64, _i64, _i32, _i32]),
    'se3_cross_eq_stack_x6_fwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _sz, _vp]),
    'se3_pairwise_distance': (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i64, _i64, _i32, _vp, _vp]),
    'se3_gram_stack': (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _i32, _vp, _vp]),
    'se3_gram_frobenius': (_i32, [_vp, _vp, _i32, _i32, _i64, _f32, _vp, _vp]),
    'se3_cross_eq_stack_fwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i64, _i32, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'se3_cross_eq_apply': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    'se3_geo_embedding_workspace_bytes': (_sz, [_i32]),
    'se3_geo_embedding_bwd_operands': (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'se3_geo_embedding_fwd': (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _f32, _vp, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    'se3_geo_embedding_bf16_fwd': (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _f32, _vp, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    'se3_embedding_table_state_bytes': (_sz, []),
    'se3_embedding_table_refresh': (_i32, [_vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp]),
    'se3_knn3': (_i32, [_vp, _i32, _vp, _vp]),
    'se3_knn3_stack': (_i32, [_vp, _vp, _i32, _vp, _vp]),
    'se3_anchor_max': (_i32, [_vp, _i32, _i64, _i32, _i64, _i64, _vp, _vp]),
    'se3_point_to_node_partition': (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'se3_superpoint_scores': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    'se3_superpoint_scores_stack': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp]),
    'se3_point_to_node_partition_stack': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'se3_weighted_procrustes': (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _f32, _f32, _vp, _vp]),
    'se3_mutual_topk_mask': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    'se3_count_inliers': (_i32, [_vp, _vp, _i64, _vp, _i32, _f32, _vp, _vp]),
    'se3_count_inliers_ranges': (_i32, [_vp, _vp, _i64, _vp, _i32, _vp, _vp, _f32, _vp, _vp]),
    'se3_weighted_procrustes_segments': (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _f32, _f32, _vp, _vp]),
    'se3_vgtk_gather_points_fwd': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    'se3_vgtk_gather_points_bwd': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    'se3_vgtk_anchor_query': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'se3_vgtk_initial_anchor_query': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp]),
    'se3_vgtk_ball_query': (_i32, [_vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp]),
    'se3_vgtk_furthest_point_sampling': (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    'se3_vgtk_inter_zpconv_fwd': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'se3_vgtk_inter_zpconv_bwd_workspace_bytes': (_sz, [_i32, _i32, _i32, _i32, _i32, _i32]),
    'se3_vgtk_inter_zpconv_bwd': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    'se3_vgtk_intra_zpconv_fwd': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'se3_vgtk_intra_zpconv_bwd': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'se3_grid_subsample_host': (_i32, [_vp, _vp, _i64, _vp, _i32, _f32, _vp, _vp, _vp]),
    'se3_neighbor_table_trim': (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    'se3_radius_neighbors_host': (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _f32, _i64, _vp, _vp]),
    'se3_log_sinkhorn_fwd': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    'se3_log_sinkhorn_bwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('%s is missing: build it with `python -m se3et_amd.build` '
                               '(there is no CPU / eager fallback for the SE3ET hot path)' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        raise RuntimeError('%s failed (code %d): %s' % (what, status, lib().se3_last_error().decode()))
