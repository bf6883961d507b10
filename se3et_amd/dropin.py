"""The drop-in recipe of INTEGRATION.md section 1 as ONE function, shared by the documentation and by tests/test_dropin_recipe.py so that
the two cannot drift.

The reference has no plugin registry: experiments/<variant>/{model,backbone,loss}.py import the operator layer BY NAME from
`geotransformer.modules.*` (experiments/se3ete.3dmatch/model.py:6-16, backbone.py:4-6).  `install_aliases()` therefore registers the
packages of `se3et_amd.modules` under those names in `sys.modules` before the experiment's `import model`:

    import se3et_amd.dropin
    se3et_amd.dropin.install_aliases()
    from model import create_model            # the reference's own experiments/<variant>/model.py, unchanged

What is replaced: geotransformer.modules.{ops, e2pn, e2pn.blocks_epn, kpconv, transformer, geotransformer, sinkhorn, registration},
`geotransformer.ext` (host-memory twin, se3et_amd.ext) and -- because the EPN toolkit's CUDA extensions cannot be built on ROCm -- `vgtk`,
`vgtk.functional`, `vgtk.so3conv` (se3et_amd.vgtk; the experiments import them and never call them).  What is NOT replaced keeps coming
from the reference tree when it is importable: sub-modules the hot path does not contain (`geotransformer.modules.loss`,
`...registration.metrics`, `...ops.vector_angle`, `geotransformer.utils.*`) are found through the reference's directories appended to the
mirrored packages' `__path__`, and single names of a mirrored package that the mirror does not define (e.g. `rodrigues_rotation_matrix`)
through `reference_attribute`.  Nothing here is on the forward's path."""
import importlib
import importlib.util
import os
import pkgutil
import sys
import types

PREFIX = 'geotransformer.modules.'
OPERATOR_PACKAGES = ('ops', 'e2pn', 'e2pn.blocks_epn', 'kpconv', 'transformer', 'geotransformer', 'sinkhorn', 'registration')
VGTK_NAMES = ('vgtk', 'vgtk.functional', 'vgtk.so3conv')
_state = {'installed': False, 'reference_modules': None}


def _reference_modules_dir():
    """<reference>/geotransformer/modules of an importable reference tree, or None (the aliases work without one)."""
    if _state['reference_modules'] is None:
        found = ''
        try:
            spec = importlib.util.find_spec('geotransformer')
        except (ImportError, ValueError):
            spec = None
        for base in (list(spec.submodule_search_locations) if spec is not None and spec.submodule_search_locations else []):
            if os.path.isdir(os.path.join(base, 'modules')):
                found = os.path.join(base, 'modules')
                break
        _state['reference_modules'] = found
    return _state['reference_modules'] or None


def install_aliases(vgtk=True, ext=True):
    """Registers the mirrors under the reference's module names (idempotent).  Call before the experiment's `import model`.  Returns the
    list of module names that were registered."""
    done = []
    ref = _reference_modules_dir()
    for name in OPERATOR_PACKAGES:
        mod = importlib.import_module('se3et_amd.modules.' + name)
        sys.modules[PREFIX + name] = mod
        done.append(PREFIX + name)
        # the mirror's own sub-modules under both names as ONE module object (a second import under the reference's name would execute
        # the file again, with its relative imports pointing into the reference's package)
        for info in (pkgutil.iter_modules(list(mod.__path__)) if hasattr(mod, '__path__') else ()):
            if not info.ispkg and PREFIX + name + '.' + info.name not in sys.modules:
                sys.modules[PREFIX + name + '.' + info.name] = importlib.import_module(mod.__name__ + '.' + info.name)
        # sub-modules the mirror does not hold (off the hot path) are searched in the reference's directory AFTER the mirror's own
        sub = os.path.join(ref, *name.split('.')) if ref else None
        if sub and hasattr(mod, '__path__') and os.path.isdir(sub) and sub not in list(mod.__path__):
            mod.__path__.append(sub)
    # `import geotransformer.modules.ops` binds attributes on the parent packages as it goes: make them agree with sys.modules.  Without a
    # reference tree (the GPU box) the two parents are empty namespace modules created here.
    parents = []
    for pname in ('geotransformer', 'geotransformer.modules'):
        if ref:
            parents.append(importlib.import_module(pname))
        else:
            parents.append(sys.modules.setdefault(pname, types.ModuleType(pname)))
            if not hasattr(parents[-1], '__path__'):
                parents[-1].__path__ = []
    parents[0].modules = parents[1]
    for name in OPERATOR_PACKAGES:
        if '.' not in name:
            setattr(parents[1], name, sys.modules[PREFIX + name])
    if ext:
        from . import ext as host_ext
        sys.modules['geotransformer.ext'] = host_ext
        done.append('geotransformer.ext')
        parents[0].ext = host_ext
    if vgtk:
        from . import vgtk as toolkit
        pkg = types.ModuleType('vgtk')
        pkg.__doc__ = 'se3et_amd.dropin: the EPN toolkit names the experiments import, on se3et_amd.vgtk'
        pkg.__path__ = []
        pkg.__getattr__ = lambda name: getattr(toolkit, name)
        pkg.functional = pkg.so3conv = toolkit
        sys.modules['vgtk'] = pkg
        sys.modules['vgtk.functional'] = sys.modules['vgtk.so3conv'] = toolkit
        done += list(VGTK_NAMES)
    _state['installed'] = True
    return done


_loaded = {}


def reference_attribute(package, name, shadowed=()):
    """Module-level __getattr__ of a mirrored package: `name` is not defined by the mirror.  With the aliases installed and a reference
    tree importable, looks it up in the reference's files of the same package that the mirror shadows by name (`shadowed`) and in the
    reference package's remaining sub-modules; AttributeError otherwise (what a plain module would raise)."""
    ref = _reference_modules_dir()
    if name.startswith('__') or not _state['installed'] or ref is None:
        raise AttributeError('module %r has no attribute %r' % (package, name))
    rel = package.split('se3et_amd.modules.', 1)[-1]
    directory = os.path.join(ref, *rel.split('.'))
    for sub in shadowed:
        key = (rel, sub)
        if key not in _loaded:
            path = os.path.join(directory, sub + '.py')
            mod = None
            if os.path.exists(path):
                spec = importlib.util.spec_from_file_location(PREFIX + rel + '._reference_' + sub, path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
            _loaded[key] = mod
        if _loaded[key] is not None and hasattr(_loaded[key], name):
            return getattr(_loaded[key], name)
    raise AttributeError('module %r has no attribute %r (not part of the SE3ET hot path, and not found in the reference tree at %s)'
                         % (package, name, directory))
