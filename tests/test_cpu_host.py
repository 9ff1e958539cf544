"""CPU-side checks (no GPU): the C ABI library loads and exports every symbol the header
declares, the host mirrors construct with the reference's parameter names / shapes / init
stream, FullBatch reproduces the reference's boundary cast, the synthetic generator follows
the adjacency layout contract.  No compute call is made."""
import os
import re

import numpy as np
import pytest
import torch

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mrgcn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mrgcn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from mrgcn_amd import _lib
    names = header_functions()
    assert len(names) >= 20
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    lib = _lib.load()  # raises if the .so is missing or a symbol is absent
    for n in names:
        assert hasattr(lib, n), n
    assert lib.mrgcn_abi_version() == 5
    assert lib.mrgcn_arch() == b"gfx950"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mrgcn_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MrgcnError):
        _lib.load()


def test_no_cpu_path():
    """The product path refuses CPU tensors instead of falling back."""
    from mrgcn_amd._lib import MrgcnError
    from mrgcn_amd.plan import GraphPlan
    g, A = util.load_graph("graph_small")
    t = util.coo_tensor(A, "ref_int8")
    with pytest.raises(MrgcnError):
        GraphPlan(t, int(g["num_nodes"]), 2 * int(g["num_pred"]) + 1)


def test_product_never_imports_oracle():
    """Nothing under mrgcn_amd/ imports `oracle`, and no string that is not a docstring names it (a path handed to
    ctypes / subprocess / importlib would be such a string).  Looks at the syntax tree: prose may mention the word."""
    import ast
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mrgcn_amd")):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            docstrings = set()
            for node in ast.walk(tree):
                if isinstance(node, (ast.Module, ast.ClassDef, ast.FunctionDef, ast.AsyncFunctionDef)):
                    b = node.body
                    if b and isinstance(b[0], ast.Expr) and isinstance(b[0].value, ast.Constant) \
                            and isinstance(b[0].value.value, str):
                        docstrings.add(id(b[0].value))
            for node in ast.walk(tree):
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                elif isinstance(node, ast.Constant) and isinstance(node.value, str) and id(node) not in docstrings:
                    assert "oracle" not in node.value.lower(), f"{f}:{node.lineno} holds a string naming the oracle"
                    continue
                else:
                    continue
                for n in names:
                    assert n.split(".")[0] != "oracle", f"{f}:{node.lineno} imports {n}"


@pytest.mark.parametrize("name", util.rgcn_cases())
def test_rgcn_parameters_and_init_stream_match_reference(name):
    """Same names, shapes and — given the same seed — the same initial values as the
    reference's RGCN (state-dict keys are API: checkpoints, optimizer_params)."""
    c = util.load_case(name)
    model, _ = util.build_rgcn_from_case(c, "cpu")
    sd = model.state_dict()
    ref_keys = sorted(k[len("init."):] for k in c.files if k.startswith("init."))
    assert sorted(sd.keys()) == ref_keys
    for k in ref_keys:
        assert tuple(sd[k].shape) == tuple(c["init." + k].shape), k
        if not k.endswith(".b"):  # the golden generator overwrote the (zero) biases
            np.testing.assert_array_equal(sd[k].numpy(), c["init." + k], err_msg=k)
        else:
            assert float(sd[k].abs().max()) == 0.0


@pytest.mark.parametrize("name", ["mrgcn_small_featureless_b0", "mrgcn_small_featureless_b3",
                                  "mrgcn_small_encoders_b3"])
def test_mrgcn_state_dict_and_optimizer_groups(name):
    from mrgcn_amd.models.mrgcn import MRGCN
    c = util.load_case(name)
    N, R = int(c["meta.num_nodes"]), int(c["meta.R"])
    enc = bool(c["meta.with_encoders"])
    modules_config = []
    if enc:
        modules_config = [("xsd.numeric", (4, 4, 0.0), False), ("xsd.boolean", (1, 2, 0.0), False)]
        modules_config.sort(key=lambda t: t[0])
    xw = 6 if enc else 0
    modules = [(xw, int(c["meta.hidden"]), "mrgcn", torch.nn.ReLU()),
               (int(c["meta.hidden"]), int(c["meta.num_classes"]), "mrgcn", None)]
    torch.manual_seed(int(c["meta.seed"]))
    model = MRGCN(modules, modules_config, R, N, num_bases=int(c["meta.num_bases"]), p_dropout=0.0,
                  featureless=not enc, bias=False)
    names = [n for n, _ in model.named_parameters()]
    assert names == [str(x) for x in c["param_names"]]
    sd = model.state_dict()
    for k in c.files:
        if k.startswith("init."):
            np.testing.assert_array_equal(sd[k[5:]].cpu().numpy(), c[k], err_msg=k)
    assert "relational" in model.devices and model.rgcn.num_layers == 2
    assert set(model.gate_map) == ({"xsd_boolean_0", "xsd_numeric_1"} if enc else set())


@pytest.mark.parametrize("gname", ["graph_small", "graph_smoke"])
def test_fullbatch_boundary_cast(gname):
    from mrgcn_amd.data.batch import FullBatch
    g, A = util.load_graph(gname)
    N = int(g["num_nodes"])
    b = FullBatch(A, [np.empty((N, 0), dtype=float)], np.arange(N))
    b.pad_(); b.to_dense_(); b.as_tensors_()
    assert b.A.dtype == torch.int8 and b.A.is_sparse
    np.testing.assert_array_equal(b.A._indices().numpy(), g["coo_indices"])
    np.testing.assert_array_equal(b.A._values().numpy(), g["coo_values_i8"])
    b2 = FullBatch(A, None, np.arange(N), value_mode="norm_f32")
    b2.as_tensors_()
    np.testing.assert_array_equal(b2.A._values().numpy(), A.data)


def test_synth_matches_layout_contract():
    import scipy.sparse as sp
    from mrgcn_amd import synth
    from oracle import rgcn_oracle as O
    g = synth.make_graph("aifb", seed=3, scale=0.25)
    A = O.build_stacked_adjacency(g.triples, g.num_nodes, g.num_pred)
    B = sp.csr_matrix((g.vals, (g.rows, g.cols)), shape=A.shape)
    assert A.nnz == B.nnz == g.nnz == 2 * len(g.triples) + g.num_nodes
    assert abs(A - B).max() == 0.0
    gi = synth.make_graph("aifb", seed=3, scale=0.25, value_mode="ref_int8")
    idx, val = O.csr_to_coo(A, "ref_int8")
    C = sp.csr_matrix((gi.vals.astype(np.float32), (gi.rows, gi.cols)), shape=A.shape)
    D = sp.csr_matrix((val.astype(np.float32), (idx[0], idx[1])), shape=A.shape)
    assert abs(C - D).max() == 0.0
    assert len(np.unique(g.triples, axis=0)) == len(g.triples)


def test_numpy_plan_reference_is_consistent():
    g, A = util.load_graph("graph_smoke")
    N, R = int(g["num_nodes"]), 2 * int(g["num_pred"]) + 1
    idx = g["coo_indices"]
    p = util.numpy_plan(idx[0], idx[1], A.data, N, N, R)
    assert p["nnz"] == A.nnz
    # every CSR entry maps to the compact column holding its literal column
    np.testing.assert_array_equal(p["ulcol"][p["ccol"]], p["lcol"])
    np.testing.assert_array_equal(np.sort(p["rperm"]), np.arange(p["ncols"]))
    assert p["cptr"][-1] == p["nnz"] and p["nptr"][-1] == p["ncols"] and p["relptr"][-1] == p["ncols"]


def test_label_reach_order_puts_reachable_nodes_first():
    """data.reorder: a permutation; the nodes that rows within reach of the labels read from come
    first (rising old id inside both groups); relabelling keeps the relation of every entry."""
    import numpy as np
    from mrgcn_amd.data import reorder
    N, R = 50, 3
    # chain 0 <- 1 <- 2 <- ... (row i reads node i+1 under relation i % 2), self loops under relation 2
    rows = np.concatenate([np.arange(N - 1), np.arange(N)])
    cols = np.concatenate([(np.arange(N - 1) % 2) * N + np.arange(1, N), 2 * N + np.arange(N)])
    idx = np.array([10, 30])
    order, inv = reorder.label_reach_order(rows, cols, N, R, idx, hops=2)
    assert sorted(order.tolist()) == list(range(N))
    assert (order[inv] == np.arange(N)).all()
    assert order[:6].tolist() == [10, 11, 12, 30, 31, 32]          # 2 hops along the chain
    assert (np.diff(order[6:]) > 0).all()
    r2, c2 = reorder.relabel_coo(rows, cols, N, inv)
    assert (c2 // N == cols // N).all()
    assert (order[r2] == rows).all() and (order[c2 % N] == cols % N).all()


@pytest.mark.parametrize("row_sparse", [False])
def test_reference_optimizer_checkpoint_loads_and_steps_under_the_unpatched_lines(row_sparse):
    """node_classification.py:35-37, :73-80 / run.py:230-236 on a module namespace as `install_as_mrgcn()` leaves the
    reference's task modules: `optimizer = optim.Adam(...)`; `optimizer.load_state_dict(checkpoint[...])` with the
    REFERENCE's own optimizer state (tests/golden/optim_checkpoint.npz: torch.optim.Adam over the reference model after
    two epochs, `weight_I` moments `(B*N, out)`); one more epoch — gradients from the float64 oracle, there is no GPU
    here — lands on the reference's parameters and optimizer state after ITS third epoch; `optimizer.state_dict()` hands
    the moments back in the reference's shapes."""
    import types

    import mrgcn_amd
    from mrgcn_amd.models.rgcn import RGCN
    from oracle import rgcn_oracle as O
    c = np.load(os.path.join(util.GOLDEN, "optim_checkpoint.npz"))
    g, A = util.load_graph("graph_small")
    N, R, B = int(c["meta.num_nodes"]), int(c["meta.R"]), int(c["meta.num_bases"])
    dims = [tuple(int(x) for x in d) for d in c["dims"]]
    task = types.ModuleType("fake_task_module")
    task.optim, task.nn = torch.optim, torch.nn
    mrgcn_amd.patch_task_optimizer(task, row_sparse=row_sparse)
    model = RGCN([(6, 8, "mrgcn", torch.nn.ReLU()), (8, 4, "mrgcn", None)], R, N, B, 0.0, False, True, False)
    names = [n for n, _ in model.named_parameters()]
    assert names == [str(n) for n in c["param_names"]]
    optimizer = task.optim.Adam([{"params": list(model.parameters())}], lr=0.01, weight_decay=0.0)
    assert isinstance(optimizer, torch.optim.Adam)
    checkpoint = {"model_state_dict": {n: torch.from_numpy(c["state2." + n]) for n in names},
                  "optimizer_state_dict": {"state": {i: {"step": torch.tensor(float(c[f"optim2.{i}.step"])),
                                                         "exp_avg": torch.from_numpy(c[f"optim2.{i}.exp_avg"]),
                                                         "exp_avg_sq": torch.from_numpy(c[f"optim2.{i}.exp_avg_sq"])}
                                                     for i in range(len(names))},
                                           "param_groups": torch.optim.Adam([torch.nn.Parameter(torch.zeros(1)) for _ in names],
                                                                            lr=0.01).state_dict()["param_groups"]}}
    model.load_state_dict(checkpoint["model_state_dict"])            # node_classification.py:78
    optimizer.load_state_dict(checkpoint["optimizer_state_dict"])    # node_classification.py:79
    wI = model.layers["layer_0"].weight_I
    assert tuple(optimizer.state[wI]["exp_avg"].shape) == tuple(wI.shape) == (N, B, 8)
    # the third epoch's gradients (float64 oracle on the reference's state after two), clipped like the loop does
    state = {n: c["state2." + n] for n in names}
    idx, val = O.csr_to_coo(A, "norm_f32")
    rec = O.train_steps(dims, R, N, B, True, False, state, c["X"], O.coo_to_csr(idx, val, A.shape), c["labels_idx"],
                        c["labels_y"], 1)[0]
    np.testing.assert_allclose(rec["loss"], float(c["loss_step3"]), rtol=1e-5)
    for n, p in model.named_parameters():
        gr = torch.from_numpy(rec["grads"][n]).float()
        p.grad = gr.view(B, N, -1).permute(1, 0, 2).contiguous() if p is wI else gr
    task.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    optimizer.step()
    sd = model.state_dict()
    for n in names:
        diff = np.abs(sd[n].numpy() - c["state3." + n])
        assert diff.max() <= 2e-5, (n, float(diff.max()))        # (moments in play: no sign-flip slack needed)
    out = optimizer.state_dict()                                     # run.py:233
    for i, n in enumerate(names):
        st = out["state"][i]
        assert tuple(st["exp_avg"].shape) == c[f"optim3.{i}.exp_avg"].shape, n
        np.testing.assert_allclose(st["exp_avg"].numpy(), c[f"optim3.{i}.exp_avg"], rtol=1e-3, atol=1e-7, err_msg=n)
        np.testing.assert_allclose(st["exp_avg_sq"].numpy(), c[f"optim3.{i}.exp_avg_sq"], rtol=2e-3, atol=1e-10, err_msg=n)
        assert float(st["step"]) == 3.0


def test_library_configuration_table_through_the_abi():
    """include/mrgcn_hip.h, configuration: one table, read / written through the ABI — no launcher reads the
    environment after the first use, a set value is what the next call sees."""
    from mrgcn_amd import _lib
    cfg = _lib.config()
    assert len(cfg) >= 30 and cfg["adam_list"] in (0, 1) and cfg["node_band"] == 131072
    prev = _lib.set_config(adam_list=0, spmm_wpe=3)
    try:
        now = _lib.config()
        assert now["adam_list"] == 0 and now["spmm_wpe"] == 3 and prev["spmm_wpe"] == cfg["spmm_wpe"]
        # the environment spelling names the same entry; the environment itself is no longer consulted
        import ctypes as C
        lib = _lib.load()
        v = C.c_int64()
        os.environ["MRGCN_SPMM_WPE"] = "5"
        try:
            assert lib.mrgcn_config_get(b"MRGCN_SPMM_WPE", C.byref(v)) == 0 and v.value == 3
        finally:
            del os.environ["MRGCN_SPMM_WPE"]
        assert lib.mrgcn_config_set(b"no_such_switch", 1) != 0
    finally:
        _lib.set_config(**prev)
    assert _lib.config() == cfg
    # no getenv left in the compute sources (the table's initialiser is the only reader)
    for f in os.listdir(os.path.join(ROOT, "mrgcn_amd", "csrc")):
        if f.endswith((".hip", ".hpp")) and f != "config.hip":
            assert "getenv" not in open(os.path.join(ROOT, "mrgcn_amd", "csrc", f)).read(), f


def test_cpu_pool_follows_the_container_quota(monkeypatch, tmp_path):
    """mrgcn_amd.host: cgroup v2 `cpu.max` -> CPUs per period (None for "max" / no file); the torch intra-op pool is
    sized to it, never enlarged."""
    import builtins
    import torch
    from mrgcn_amd import host
    real_open = builtins.open

    def fake(content):
        f = tmp_path / "cpu.max"
        f.write_text(content)
        return lambda p, *a, **k: real_open(f if p == "/sys/fs/cgroup/cpu.max" else p, *a, **k)
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert host.cpu_quota() == 16.0
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert host.cpu_quota() is None
    before = torch.get_num_threads()
    try:
        monkeypatch.setattr(builtins, "open", fake("200000 100000\n"))
        assert host.fit_cpu_pool_to_quota() == 2.0 and torch.get_num_threads() == min(before, 2)
        monkeypatch.setattr(builtins, "open", fake("%d 100000\n" % (100000 * 4096)))
        host.fit_cpu_pool_to_quota()
        assert torch.get_num_threads() == min(before, 2)      # (never enlarged)
    finally:
        monkeypatch.undo()
        torch.set_num_threads(before)
