"""mrgcn_amd — MI355X-native full-batch R-GCN message passing behind the API surface of
wxwilcke/mrgcn's `mrgcn.layers.graph.GraphConvolution`, `mrgcn.models.rgcn.RGCN`,
`mrgcn.models.mrgcn.MRGCN` and `mrgcn.data.batch.FullBatch`.

Compute lives in `csrc/` (hand-written HIP for gfx950 behind the C ABI of
`include/mrgcn_hip.h`); this package is the Python host side.  There is no CPU or
PyTorch fallback for the kernels: without the built library the ops raise.
"""
__version__ = "0.2.0"

from .stats import reset_stats, stats  # noqa: E402,F401  (which path did a step take?)

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
_TASK_ROW_SPARSE = False   # what install_as_mrgcn last asked for (the task modules may be imported later)


def patch_task_optimizer(module, row_sparse: bool = True) -> None:
    """Rebinds `optim.Adam` inside one of the reference's task modules (`mrgcn.tasks.node_classification`,
    `...link_prediction`; `import torch.optim as optim`, `import torch.nn as nn` at their top) through pass-through
    namespaces that differ from torch's in exactly the attributes named here.  Nothing outside that module's
    namespace changes.

    `row_sparse=False`: `optim.Adam` -> `mrgcn_amd.optim.ReferenceLayoutAdam` — torch's own Adam whose `state_dict()` /
    `load_state_dict()` speak the reference's `(B*N, out)` layout for the node-major `weight_I`, so that the module's
    unpatched checkpoint lines (node_classification.py:73-80, run.py:230-236) exchange optimizer states with the
    reference.  `nn` stays torch's.

    `row_sparse=True`: `optim.Adam` -> `optim.RowSparseAdam` and `nn.utils.clip_grad_norm_` -> this package's — the pair
    that understands the row-sparse node-table gradient; bound together, never one without the other.  The training
    loop of the module (node_classification.py:35-37, :190-193; link_prediction.py:325) then runs unchanged on the
    row-sparse fast path (same checkpoint layout)."""
    import types

    import torch

    from . import optim as fast

    class _Through(types.SimpleNamespace):
        def __init__(self, base, **over):
            super().__init__(**over)
            self.__dict__["_base"] = base

        def __getattr__(self, name):
            return getattr(self.__dict__["_base"], name)

    if not row_sparse:
        if hasattr(module, "optim"):
            module.optim = _Through(torch.optim, Adam=fast.ReferenceLayoutAdam)
        if type(getattr(module, "nn", None)).__name__ == "_Through":
            module.nn = torch.nn   # (a previous row-sparse patch: the pair goes together)
        return
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

    The reference's two task modules always get `optim.Adam` rebound (`patch_task_optimizer`), now if they are imported
    already and otherwise when they are: by default to torch's own Adam with the checkpoint layout translated
    (`optim.ReferenceLayoutAdam`: a reference-shaped `optimizer_state_dict` loads and steps under the module's
    unpatched lines, and what it saves loads in the reference); with `patch_optimizer=True` to `RowSparseAdam`
    together with this package's `nn.utils.clip_grad_norm_`: the reference's training loop then reaches the row-sparse
    node-table update without a changed line.  The last call's choice wins."""
    global _TASK_ROW_SPARSE
    _TASK_ROW_SPARSE = bool(patch_optimizer)
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
            patch_task_optimizer(module, _TASK_ROW_SPARSE)

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
    if not any(getattr(f, "_mrgcn_amd", None) == "tasks" for f in sys.meta_path):
        sys.meta_path.insert(0, _TaskFinder())
    for name in _TASK_MODULES:
        if name in sys.modules:
            patch_task_optimizer(sys.modules[name], _TASK_ROW_SPARSE)
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
