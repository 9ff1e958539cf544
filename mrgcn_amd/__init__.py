"""mrgcn_amd — MI355X-native full-batch R-GCN message passing behind the API surface of
wxwilcke/mrgcn's `mrgcn.layers.graph.GraphConvolution`, `mrgcn.models.rgcn.RGCN`,
`mrgcn.models.mrgcn.MRGCN` and `mrgcn.data.batch.FullBatch`.

Compute lives in `csrc/` (hand-written HIP for gfx950 behind the C ABI of
`include/mrgcn_hip.h`); this package is the Python host side.  There is no CPU or
PyTorch fallback for the kernels: without the built library the ops raise.
"""
__version__ = "0.2.0"

# the reference modules this package answers for (run.py:12-19, tasks/node_classification.py:9-16,
# tasks/link_prediction.py:12-19 import these names)
_ALIASES = {
    "mrgcn.layers.graph": "mrgcn_amd.layers.graph",
    "mrgcn.models.rgcn": "mrgcn_amd.models.rgcn",
    "mrgcn.models.mrgcn": "mrgcn_amd.models.mrgcn",
    "mrgcn.models.perceptron": "mrgcn_amd.models.perceptron",
    "mrgcn.models.temporal_cnn": "mrgcn_amd.models.temporal_cnn",
    "mrgcn.data.batch": "mrgcn_amd.data.batch",
    "mrgcn.data.io.tarball": "mrgcn_amd.data.io.tarball",
}
_PARENTS = ("mrgcn", "mrgcn.layers", "mrgcn.models", "mrgcn.data", "mrgcn.data.io")


def install_as_mrgcn():
    """Makes `import mrgcn.layers.graph`, `mrgcn.models.{rgcn,mrgcn,perceptron,temporal_cnn}`,
    `mrgcn.data.batch` and `mrgcn.data.io.tarball` resolve to this package's modules, so that scripts
    written against the reference (`run.py:12-19`) get the MI355X implementations unchanged.

    Only those seven leaf modules are replaced.  Every other `mrgcn.*` module (`mrgcn.tasks.*`,
    `mrgcn.data.utils`, `mrgcn.encodings.*`, `mrgcn.data.io.tsv`, ...) keeps resolving to the
    reference installation through the normal import machinery, in any import order: a finder at the
    head of `sys.meta_path` answers for the leaves, and a second one at the tail provides empty
    namespace packages for `mrgcn`, `mrgcn.layers`, ... only when no reference installation is
    importable at all.  Call it before the task modules are imported (a module that already ran
    `from mrgcn.models.mrgcn import MRGCN` keeps the class it bound).  Idempotent.  See INTEGRATION.md."""
    import importlib
    import importlib.abc
    import importlib.machinery
    import sys

    class _Loader(importlib.abc.Loader):
        def __init__(self, real):
            self.real = real

        def create_module(self, spec):
            return importlib.import_module(self.real)

        def exec_module(self, module):
            pass

    class _LeafFinder(importlib.abc.MetaPathFinder):
        _mrgcn_amd = "leaves"

        def find_spec(self, fullname, path=None, target=None):
            real = _ALIASES.get(fullname)
            if real is None:
                return None
            return importlib.machinery.ModuleSpec(fullname, _Loader(real))

    class _ParentFinder(importlib.abc.MetaPathFinder):
        """Last resort: the reference is not installed, the parents of the leaves are empty packages."""
        _mrgcn_amd = "parents"

        def find_spec(self, fullname, path=None, target=None):
            if fullname not in _PARENTS:
                return None
            spec = importlib.machinery.ModuleSpec(fullname, None, is_package=True)
            spec.submodule_search_locations = []
            return spec

    if not any(getattr(f, "_mrgcn_amd", None) == "leaves" for f in sys.meta_path):
        sys.meta_path.insert(0, _LeafFinder())
    if not any(getattr(f, "_mrgcn_amd", None) == "parents" for f in sys.meta_path):
        sys.meta_path.append(_ParentFinder())
    # leaves that were imported from the reference before this call are replaced in place
    for alias, real in _ALIASES.items():
        parent, leaf = alias.rsplit(".", 1)
        if alias in sys.modules or parent in sys.modules:
            mod = importlib.import_module(real)
            sys.modules[alias] = mod
            if parent in sys.modules:
                setattr(sys.modules[parent], leaf, mod)
