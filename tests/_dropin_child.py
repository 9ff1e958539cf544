"""Child process of tests/test_dropin.py (AUTHORING CONTAINER ONLY: needs /root/reference on disk).
Usage: python tests/_dropin_child.py <first|after_pkg|after_leaf>"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_goldens import REF, _stub_rdflib  # noqa: E402

_stub_rdflib()
sys.path.insert(0, REF)
order = sys.argv[1]
import mrgcn_amd  # noqa: E402

if order == "first":
    mrgcn_amd.install_as_mrgcn()
elif order == "after_pkg":  # the reference's package and a non-replaced module are already in
    import mrgcn
    import mrgcn.data.utils
    mrgcn_amd.install_as_mrgcn()
elif order == "after_leaf":  # replaced modules were already imported from the reference
    import mrgcn.data.batch
    import mrgcn.models.mrgcn
    mrgcn_amd.install_as_mrgcn()
elif order == "patched_first":   # the optimizer drop-ins: task modules imported after the call ...
    mrgcn_amd.install_as_mrgcn(patch_optimizer=True)
elif order == "patched_after":   # ... and before it (after a first, plain install)
    mrgcn_amd.install_as_mrgcn()
    import mrgcn.tasks.node_classification
    import mrgcn.tasks.link_prediction
    mrgcn_amd.install_as_mrgcn(patch_optimizer=True)
else:
    raise SystemExit("order?")
mrgcn_amd.install_as_mrgcn(patch_optimizer=order.startswith("patched"))  # idempotent

import mrgcn.data.io.tsv  # noqa: E402,F401  (reference modules that are not replaced stay importable)
import mrgcn.data.utils  # noqa: E402,F401
import mrgcn.tasks.link_prediction as lp  # noqa: E402
import mrgcn.tasks.node_classification as nc  # noqa: E402
import mrgcn.tasks.utils  # noqa: E402,F401

import mrgcn_amd.data.batch as my_batch  # noqa: E402
import mrgcn_amd.models.mrgcn as my_mrgcn  # noqa: E402

assert nc.__file__.startswith(REF) and lp.__file__.startswith(REF)
import torch  # noqa: E402
if order.startswith("patched"):
    # exactly two names differ from torch's inside the reference's task modules (node_classification.py:35, :192;
    # link_prediction.py:325); everything else passes through, and torch itself is untouched
    import mrgcn_amd.optim as fast
    for mod in (nc, lp):
        assert mod.optim.Adam is fast.Adam and mod.nn.utils.clip_grad_norm_ is fast.clip_grad_norm_
        assert mod.optim.SGD is torch.optim.SGD and mod.nn.CrossEntropyLoss is torch.nn.CrossEntropyLoss
        assert mod.nn.utils.clip_grad_value_ is torch.nn.utils.clip_grad_value_
    assert torch.optim.Adam is not fast.Adam and torch.nn.utils.clip_grad_norm_ is not fast.clip_grad_norm_
else:
    assert nc.optim is torch.optim and nc.nn is torch.nn
assert nc.MRGCN is my_mrgcn.MRGCN and lp.MRGCN is my_mrgcn.MRGCN
assert nc.FullBatch is my_batch.FullBatch and nc.MiniBatch is my_batch.MiniBatch
import mrgcn.layers.graph  # noqa: E402
import mrgcn.models.rgcn  # noqa: E402
from mrgcn.data.io.tarball import Tarball  # noqa: E402
import mrgcn_amd.data.io.tarball as my_tar  # noqa: E402

assert Tarball is my_tar.Tarball
assert mrgcn.layers.graph.GraphConvolution.__module__ == "mrgcn_amd.layers.graph"

# the reference's own build_model (node_classification.py:385-430) constructs mrgcn_amd's MRGCN
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402

N, R, C = 30, 5, 3
Y = {"train": sp.csr_matrix((N, C), dtype=np.int8)}
A = sp.csr_matrix((N, R * N), dtype=np.float32)
config = {"model": {"layers": [{"hidden_nodes": 8, "type": "mrgcn"}, {"hidden_nodes": 8, "type": "mrgcn"}],
                    "num_bases": 2, "p_dropout": 0.0, "bias": True}, "task": {}}
model = nc.build_model(0, Y, A, [], config, True)
assert type(model) is my_mrgcn.MRGCN, type(model)
names = sorted(n for n, _ in model.named_parameters())
assert names == ["rgcn.layers.layer_0.b", "rgcn.layers.layer_0.weight_I", "rgcn.layers.layer_0.weight_I_comp",
                 "rgcn.layers.layer_1.b", "rgcn.layers.layer_1.weight_F", "rgcn.layers.layer_1.weight_F_comp"], names
# ... and its optimizer grouping (tasks/utils.py:8-45) accepts the model
groups = mrgcn.tasks.utils.optimizer_params(model, {}, True)
assert sum(len(g["params"]) for g in groups) == len(names)
print("dropin ok", order)
