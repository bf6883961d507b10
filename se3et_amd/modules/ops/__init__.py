"""Mirror of geotransformer/modules/ops/__init__.py:1-21: the operators of the hot path on HIP kernels, the rigid-transform helpers
their callers import from the same namespace in torch.  Names of the reference's package that are not on the path (Rodrigues helpers,
vector_angle, knn / ball-query partitions) resolve to the reference's own files when se3et_amd.dropin installed the aliases and the
reference tree is importable (module __getattr__ below), and raise AttributeError otherwise."""
from .grid_subsample import grid_subsample
from .radius_search import radius_search
from .index_select import index_select
from .pairwise_distance import pairwise_distance
from .pointcloud_partition import get_point_to_node_indices, point_to_node_partition
from .transformation import (apply_transform, apply_rotation, inverse_transform, get_transform_from_rotation_translation,
                             get_rotation_translation_from_transform)


def __getattr__(name):
    from ... import dropin
    return dropin.reference_attribute(__name__, name, ('transformation', 'vector_angle'))
