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
        assert mod.optim.Adam is fast.RowSparseAdam and mod.nn.utils.clip_grad_norm_ is fast.clip_grad_norm_
        assert mod.optim.SGD is torch.optim.SGD and mod.nn.CrossEntropyLoss is torch.nn.CrossEntropyLoss
        assert mod.nn.utils.clip_grad_value_ is torch.nn.utils.clip_grad_value_
    assert torch.optim.Adam is not fast.RowSparseAdam and torch.nn.utils.clip_grad_norm_ is not fast.clip_grad_norm_
else:
    # default install: only `optim.Adam` differs — torch's own Adam whose checkpoints are in the reference's layout
    import mrgcn_amd.optim as fast
    for mod in (nc, lp):
        assert mod.optim.Adam is fast.ReferenceLayoutAdam and issubclass(mod.optim.Adam, torch.optim.Adam)
        assert mod.optim.SGD is torch.optim.SGD and mod.nn is torch.nn
    assert torch.optim.Adam is not fast.ReferenceLayoutAdam
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

# ... and the reference's checkpoint lines (node_classification.py:35-37, :73-80; run.py:230-236), unpatched: an
# optimizer state shaped like the REFERENCE model's tensors (= this model's state_dict() shapes) loads, steps, and
# comes back out in that shape — equal to plain torch.optim.Adam over reference-shaped copies of the parameters
optimizer = nc.optim.Adam(groups, lr=0.01, weight_decay=0.0)
gen = torch.Generator().manual_seed(0)
plist = [p for g in groups for p in g["params"]]
pname = {id(p): n for n, p in model.named_parameters()}
sd_model = model.state_dict()
ref_params = [torch.nn.Parameter(sd_model[pname[id(p)]].detach().clone()) for p in plist]
ref_opt = torch.optim.Adam(ref_params, lr=0.01, weight_decay=0.0)
for rp in ref_params:
    rp.grad = torch.randn(rp.shape, generator=gen)
ref_opt.step()                       # (a state to checkpoint)
import copy  # noqa: E402
ckpt = {"optimizer_state_dict": copy.deepcopy(ref_opt.state_dict())}   # (as read back from disk: no shared tensors)
wi = [pname[id(p)] for p in plist].index("rgcn.layers.layer_0.weight_I")
assert tuple(ckpt["optimizer_state_dict"]["state"][wi]["exp_avg"].shape) == (2 * N, 8) and tuple(plist[wi].shape) == (N, 2, 8)
model.load_state_dict({pname[id(p)]: rp.detach() for p, rp in zip(plist, ref_params)})
optimizer.load_state_dict(ckpt["optimizer_state_dict"])                       # node_classification.py:79
if not order.startswith("patched"):   # (RowSparseAdam steps with HIP kernels: its step is tests/test_gpu_reference_loop.py's)
    for p, rp in zip(plist, ref_params):
        rp.grad = torch.randn(rp.shape, generator=gen)
        g_ = rp.grad
        p.grad = (g_.view(2, N, 8).permute(1, 0, 2).contiguous() if getattr(p, "_mrgcn_node_major", False) else g_.clone()).to(p.device)
    optimizer.step()
    ref_opt.step()
    sd_after = model.state_dict()
    for p, rp in zip(plist, ref_params):
        torch.testing.assert_close(sd_after[pname[id(p)]].cpu(), rp.detach(), rtol=1e-6, atol=1e-7)
out_sd, want_sd = optimizer.state_dict(), ref_opt.state_dict()               # run.py:233
for k, st in want_sd["state"].items():
    assert tuple(out_sd["state"][k]["exp_avg"].shape) == tuple(st["exp_avg"].shape)
    torch.testing.assert_close(out_sd["state"][k]["exp_avg"].cpu(), st["exp_avg"], rtol=1e-6, atol=1e-8)
    torch.testing.assert_close(out_sd["state"][k]["exp_avg_sq"].cpu(), st["exp_avg_sq"], rtol=1e-6, atol=1e-10)

# ... and an am.toml-shaped config with its hub tuples (configs/am.toml: distilbert for strings, mobilenet_v2 for
# images; graph_features.py:184-236 turns them into `modules_config`) builds unchanged: torch.hub.load is the one
# network call, replaced here by small stand-ins
import torch.nn as tnn  # noqa: E402


class _TinyLM(tnn.Module):
    def __init__(self):
        super().__init__()
        self.emb = tnn.Embedding(50, 12)
        self.lin = tnn.Linear(12, 12)

    def forward(self, ids):
        return (self.lin(self.emb(ids)),)


class _TinyImageNet(tnn.Module):
    def __init__(self):
        super().__init__()
        self.features = tnn.Sequential(tnn.Conv2d(3, 6, 3, padding=1), tnn.ReLU(), tnn.Conv2d(6, 8, 3, padding=1))
        self.classifier = tnn.Linear(8, 5)


hub_calls = []


def _fake_hub(*a, **k):
    hub_calls.append(a)
    return _TinyLM() if a[1] == "model" else _TinyImageNet()


torch.hub.load = _fake_hub
modules_config = [("blob.image", (["pytorch/vision:v0.10.0", "mobilenet_v2", "MobileNet_V2_Weights.IMAGENET1K_V1"],
                                  {"mode": "RGB", "mean": [0.485, 0.456, 0.406], "std": [0.229, 0.224, 0.225]}, 128, 0.0), False),
                  ("xsd.date", (3, 3, 0.0), False), ("xsd.numeric", (1, 4, 0.0), False), ("xsd.numeric", (1, 4, 0.0), False),
                  ("xsd.string", (["huggingface/pytorch-transformers", "model", "distilbert-base-multilingual-cased"], 16, 0.0), False)]
am_config = {"model": {"layers": [{"hidden_nodes": 10, "type": "mrgcn"}, {"hidden_nodes": 10, "type": "mrgcn"}],
                       "num_bases": 40, "p_dropout": 0.0, "bias": False}, "task": {}}
Y11 = {"train": sp.csr_matrix((N, 11), dtype=np.int8)}
am_model = nc.build_model(155, Y11, A, modules_config, am_config, False)
assert type(am_model) is my_mrgcn.MRGCN and len(hub_calls) == 2, hub_calls
assert am_model.modality_out_dim == 155 and am_model.rgcn.layers["layer_0"].weight_F.shape == (40, 155, 10)
# tasks/utils.py:8-45: one optimizer group per datatype (named by the module_dict keys), one for the gates; frozen
# backbone parameters are left out (imagecnn.py:18-20, transformer.py:17-19)
optim_config = {dt: {"lr": 0.001} for dt in ("blob.image", "xsd.date", "xsd.numeric", "xsd.string", "gate_weights")}
am_groups = mrgcn.tasks.utils.optimizer_params(am_model, optim_config, False)
assert sum(len(g["params"]) for g in am_groups) == sum(p.requires_grad for p in am_model.parameters())
assert len(am_groups) == 6, len(am_groups)
print("dropin ok", order)
