"""Mirror of the one function of geotransformer.modules.registration that the SE3ET models call (registration/__init__.py:1-11):
    get_node_correspondences        matching.py:231-315
The fitted transform of the forward is LocalGlobalRegistration (se3et_amd.modules.geotransformer, csrc/registration.hip).  The
package's evaluation metrics and the stand-alone Procrustes module are off the hot path and not restated: with the aliases of
se3et_amd.dropin installed next to a reference tree, `geotransformer.modules.registration.metrics` / `.procrustes` load from that tree
(package __path__), and the remaining names of the reference's matching.py through the module __getattr__ below."""
from .matching import get_node_correspondences


def __getattr__(name):
    from ... import dropin
    if name in ('modified_chamfer_distance', 'relative_rotation_error', 'relative_translation_error', 'isotropic_transform_error',
                'anisotropic_transform_error', 'weighted_procrustes', 'WeightedProcrustes') and dropin._state['installed']:
        import importlib
        sub = 'procrustes' if 'rocrustes' in name else 'metrics'
        try:
            return getattr(importlib.import_module('geotransformer.modules.registration.' + sub), name)
        except ImportError:
            pass
    return dropin.reference_attribute(__name__, name, ('matching',))
