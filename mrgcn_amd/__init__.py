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


_TASK_MODULES = ("mrgcn.tasks.node_classification", "mrgcn.tasks.link_prediction")


def patch_task_optimizer(module) -> None:
    """Gives one of the reference's task modules (`mrgcn.tasks.node_classification`, `...link_prediction`) this
    package's `optim.RowSparseAdam` (as `optim.Adam`) and `nn.utils.clip_grad_norm_` (mrgcn_amd.optim) — the pair that
    understands the row-sparse node-table gradient; bound together, never one without the other: the module's names `optim` and `nn`
    — `import torch.optim as optim`, `import torch.nn as nn` at its top — are rebound to pass-through namespaces that
    differ from torch's in exactly those two attributes.  The training loop of the module
    (node_classification.py:35-37, :190-193; link_prediction.py:325) then runs unchanged on the row-sparse fast
    path.  Nothing outside that module's namespace changes."""
    import types

    import torch

    from . import optim as fast

    class _Through(types.SimpleNamespace):
        def __init__(self, base, **over):
            super().__init__(**over)
            self.__dict__["_base"] = base

        def __getattr__(self, name):
            return getattr(self.__dict__["_base"], name)

    if hasattr(module, "optim"):
        module.optim = _Through(torch.optim, Adam=fast.RowSparseAdam)
    if hasattr(module, "nn"):
        module.nn = _Through(torch.nn, utils=_Through(torch.nn.utils, clip_grad_norm_=fast.clip_grad_norm_))


def install_as_mrgcn(patch_optimizer: bool = False):
    """Makes `import mrgcn.layers.graph`, `mrgcn.models.{rgcn,mrgcn,perceptron,temporal_cnn}`,
    `mrgcn.data.batch` and `mrgcn.data.io.tarball` resolve to this package's modules, so that scripts
    written against the reference (`run.py:12-19`) get the MI355X implementations unchanged.

    Only those seven leaf modules are replaced.  Every other `mrgcn.*` module (`mrgcn.tasks.*`,
    `mrgcn.data.utils`, `mrgcn.encodings.*`, `mrgcn.data.io.tsv`, ...) keeps resolving to the
    reference installation through the normal import machinery, in any import order: a finder at the
    head of `sys.meta_path` answers for the leaves, and a second one at the tail provides empty
    namespace packages for `mrgcn`, `mrgcn.layers`, ... only when no reference installation is
    importable at all.  Call it before the task modules are imported (a module that already ran
    `from mrgcn.models.mrgcn import MRGCN` keeps the class it bound).  Idempotent.  See INTEGRATION.md.

    `patch_optimizer=True` additionally rebinds `optim.Adam` and `nn.utils.clip_grad_norm_` inside the reference's two
    task modules (`patch_task_optimizer`), now if they are imported already and otherwise when they are: the
    reference's training loop then reaches the row-sparse node-table update without a changed line."""
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

    class _TaskLoader(importlib.abc.Loader):
        def __init__(self, inner):
            self.inner = inner

        def create_module(self, spec):
            return self.inner.create_module(spec)

        def exec_module(self, module):
            self.inner.exec_module(module)
            patch_task_optimizer(module)

    class _TaskFinder(importlib.abc.MetaPathFinder):
        """Wraps the loader of the reference's task modules: after the module ran, its `optim` / `nn` are patched."""
        _mrgcn_amd = "tasks"

        def find_spec(self, fullname, path=None, target=None):
            if fullname not in _TASK_MODULES:
                return None
            for f in sys.meta_path:
                if f is self or not hasattr(f, "find_spec"):
                    continue
                spec = f.find_spec(fullname, path, target)
                if spec is not None and spec.loader is not None:
                    spec.loader = _TaskLoader(spec.loader)
                    return spec
            return None

    if not any(getattr(f, "_mrgcn_amd", None) == "leaves" for f in sys.meta_path):
        sys.meta_path.insert(0, _LeafFinder())
    if patch_optimizer:
        if not any(getattr(f, "_mrgcn_amd", None) == "tasks" for f in sys.meta_path):
            sys.meta_path.insert(0, _TaskFinder())
        for name in _TASK_MODULES:
            if name in sys.modules:
                patch_task_optimizer(sys.modules[name])
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
