"""Mirror of geotransformer.modules.geotransformer for the SE3ET hot path:
    GeometricStructureEmbedding / GeometricTransformer    geotransformer.py:19-121,124-317
    SuperPointMatching                                     superpoint_matching.py:7-55
    SuperPointTargetGenerator                              superpoint_target.py:6-41
    LocalGlobalRegistration                                local_global_registration.py:11-235
"""
from .geotransformer import GeometricStructureEmbedding, GeometricTransformer
from .superpoint_matching import SuperPointMatching
from .superpoint_target import SuperPointTargetGenerator
from .local_global_registration import LocalGlobalRegistration
