"""Run by tests/test_dropin_recipe.py in a fresh interpreter (the aliases rewrite sys.modules): executes the recipe of INTEGRATION.md
section 1 -- the text of its first python block, read from the file -- in front of the reference's UNCHANGED experiment files and
prints one JSON line.  TEST INFRASTRUCTURE: needs /root/reference (build container only)."""
import json
import logging
import os
import re
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get('SE3ET_REFERENCE_ROOT', '/root/reference')
VARIANT_OF = {'se3ete.3dmatch': 'se3ete', 'se3ete2.3dmatch': 'se3ete2', 'se3eti.3dmatch': 'se3eti', 'se3eti.kitti': 'se3eti_kitti',
              'se3eti2.3dmatch': 'se3eti2'}


def recipe_text():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    section = text[text.index('## 1. Python level'):]
    return re.search(r'```python\n(.*?)```', section, re.S).group(1)


def container_stubs():
    """Third-party modules the reference imports at module level that this container lacks (a reference user has them installed; they are
    not part of the recipe): IPython, ipdb, easydict.  No stub touches geotransformer.* or vgtk."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    class EasyDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v
    for name, attrs in (('IPython', dict(embed=lambda *a, **k: None)), ('ipdb', dict(set_trace=lambda *a, **k: None)),
                        ('easydict', dict(EasyDict=EasyDict)), ('coloredlogs', dict(ColoredFormatter=logging.Formatter))):
        try:
            __import__(name)
        except ImportError:
            mod(name, **attrs)


def main(experiment):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, REFERENCE)
    container_stubs()
    exec(compile(recipe_text(), 'INTEGRATION.md section 1', 'exec'), {})          # <- the recipe, verbatim
    import geotransformer.utils.common as common                                   # config.py creates output directories on import
    common.ensure_dir = lambda *a, **k: None
    sys.path.insert(0, os.path.join(REFERENCE, 'experiments', experiment))
    import config as ref_config
    import backbone as ref_backbone                                               # noqa: F401
    import model as ref_model
    import loss as ref_loss
    cfg = ref_config.make_cfg()
    model = ref_model.create_model(cfg)
    ref_loss.OverallLoss(cfg)
    foreign = sorted({type(m).__module__ + '.' + type(m).__name__ for m in model.modules()
                      if not type(m).__module__.startswith(('se3et_amd.', 'torch.'))
                      and type(m).__module__ not in ('model', 'backbone')})
    import torch
    import se3et_amd.model as own
    ours = own.create_model(own.make_cfg(VARIANT_OF[experiment])).state_dict()
    theirs = model.state_dict()
    describe = lambda sd: {k: [list(v.shape), str(v.dtype)] for k, v in sd.items()}
    # the target generator and the ground-truth helper the reference model holds / imports must be the mirrors
    tg = model.coarse_target
    import numpy as np
    idx = torch.stack([torch.arange(300), torch.arange(300)], 1)
    ov = torch.linspace(0, 1, 300)
    np.random.seed(3)
    got = tg(idx, ov)
    np.random.seed(3)
    want = np.random.choice(np.arange(int((ov > tg.overlap_threshold).sum())), tg.num_targets, replace=False)
    print(json.dumps({
        'experiment': experiment,
        'foreign_modules': foreign,
        'model_class_module': type(model).__module__,
        'transformer_module': type(model.transformer).__module__,
        'target_generator_module': type(tg).__module__,
        'node_correspondences_module': ref_model.get_node_correspondences.__module__,
        'point_to_node_partition_module': ref_model.point_to_node_partition.__module__,
        'target_rng_matches': bool(np.array_equal(got[0].numpy(), idx[ov > tg.overlap_threshold][want][:, 0].numpy())),
        'n_keys': len(theirs),
        'only_reference': sorted(set(theirs) - set(ours)),
        'only_ours': sorted(set(ours) - set(theirs)),
        'mismatched': sorted(k for k in set(theirs) & set(ours) if describe(theirs)[k] != describe(ours)[k]),
        'metrics_from': sys.modules['geotransformer.modules.registration.metrics'].__file__,
    }))


if __name__ == '__main__':
    main(sys.argv[1])
