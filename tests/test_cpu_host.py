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
    assert lib.mrgcn_abi_version() == 3
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
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mrgcn_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"


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
