"""mrgcn_amd — MI355X-native full-batch R-GCN message passing behind the API surface of
wxwilcke/mrgcn's `mrgcn.layers.graph.GraphConvolution`, `mrgcn.models.rgcn.RGCN`,
`mrgcn.models.mrgcn.MRGCN` and `mrgcn.data.batch.FullBatch`.

Compute lives in `csrc/` (hand-written HIP for gfx950 behind the C ABI of
`include/mrgcn_hip.h`); this package is the Python host side.  There is no CPU or
PyTorch fallback for the kernels: without the built library the ops raise.
"""
__version__ = "0.1.0"


def install_as_mrgcn():
    """Registers this package's modules under the reference's import names
    (`mrgcn.layers.graph`, `mrgcn.models.{rgcn,mrgcn,perceptron,temporal_cnn}`, `mrgcn.data.batch`,
    `mrgcn.data.io.tarball`)
    so that scripts written against the reference (`run.py:12-19`) import the MI355X
    implementations unchanged.  See INTEGRATION.md."""
    import importlib
    import sys
    import types

    names = {
        "mrgcn.layers.graph": "mrgcn_amd.layers.graph",
        "mrgcn.models.rgcn": "mrgcn_amd.models.rgcn",
        "mrgcn.models.mrgcn": "mrgcn_amd.models.mrgcn",
        "mrgcn.models.perceptron": "mrgcn_amd.models.perceptron",
        "mrgcn.models.temporal_cnn": "mrgcn_amd.models.temporal_cnn",
        "mrgcn.data.batch": "mrgcn_amd.data.batch",
        "mrgcn.data.io.tarball": "mrgcn_amd.data.io.tarball",
    }
    for pkg in ("mrgcn", "mrgcn.layers", "mrgcn.models", "mrgcn.data", "mrgcn.data.io"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    for alias, real in names.items():
        mod = importlib.import_module(real)
        sys.modules[alias] = mod
        parent, leaf = alias.rsplit(".", 1)
        setattr(sys.modules[parent], leaf, mod)
    for pkg in ("mrgcn.layers", "mrgcn.models", "mrgcn.data"):
        setattr(sys.modules["mrgcn"], pkg.split(".")[1], sys.modules[pkg])
    setattr(sys.modules["mrgcn.data"], "io", sys.modules["mrgcn.data.io"])
