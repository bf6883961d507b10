"""Mirror of geotransformer/modules/ops/__init__.py:1-21 (hot-path subset)."""
from .grid_subsample import grid_subsample
from .radius_search import radius_search
from .index_select import index_select
from .pairwise_distance import pairwise_distance
from .pointcloud_partition import point_to_node_partition
