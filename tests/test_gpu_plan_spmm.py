"""GPU parity of the graph plan (indices bit-exact) and of the stacked-CSR sparse x dense
product through the C ABI, against the numpy plan reference / scipy (the oracle)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests import util

pytestmark = pytest.mark.gpu

ARRAYS = ["rowptr", "lcol", "ccol", "val", "cptr", "crow", "cval", "urel", "unode", "nptr",
          "rowidx", "ulcol", "rperm", "relptr", "mpos", "mcol", "mval", "rowmap", "ptr3"]


def _plan_from_coo(rows, cols, vals, num_rows, N, R, prune=False, row_bytes=None, lean=False):
    from mrgcn_amd.plan import GraphPlan
    idx = torch.from_numpy(np.stack([rows, cols]).astype(np.int64))
    v = torch.from_numpy(np.asarray(vals))
    A = torch.sparse_coo_tensor(idx, v, (num_rows, R * N)).cuda()
    return GraphPlan(A, N, R, prune_zeros=prune, operand_row_bytes=row_bytes, lean=lean)


def _check_plan(plan, ref):
    from mrgcn_amd import _lib as L
    assert plan.nnz == ref["nnz"] and plan.ncols == ref["ncols"]
    for i, name in enumerate(ARRAYS):
        got = plan.export(getattr(L, "ARR_" + name.upper()))
        np.testing.assert_array_equal(got, ref[name], err_msg=name)


def _random_graph(rng, num_rows, N, R, nnz, hub_rows=0, hub_len=0, hub_cols=0):
    RN = R * N
    rows = rng.integers(0, num_rows, nnz)
    cols = rng.integers(0, RN, nnz)
    if hub_rows:
        hr = rng.choice(num_rows, hub_rows, replace=False)
        rows = np.concatenate([rows] + [np.full(hub_len, h) for h in hr])
        cols = np.concatenate([cols] + [rng.choice(RN, hub_len, replace=False) for _ in hr])
    if hub_cols:
        hc = rng.choice(RN, hub_cols, replace=False)
        rows = np.concatenate([rows] + [rng.choice(num_rows, min(hub_len, num_rows), replace=False) for _ in hc])
        cols = np.concatenate([cols] + [np.full(min(hub_len, num_rows), c) for c in hc])
    key = np.unique(rows.astype(np.int64) * RN + cols)
    rows, cols = key // RN, key % RN
    perm = rng.permutation(len(key))  # uncoalesced, unordered COO as input
    rows, cols = rows[perm], cols[perm]
    vals = rng.standard_normal(len(rows)).astype(np.float32)
    return rows, cols, vals


@pytest.mark.parametrize("gname,mode", [("graph_small", "ref_int8"), ("graph_small", "norm_f32"),
                                        ("graph_smoke", "ref_int8"), ("graph_smoke", "norm_f32")])
def test_plan_indices_bit_exact_on_golden_graphs(gname, mode):
    from mrgcn_amd.plan import GraphPlan
    g, A = util.load_graph(gname)
    N, R = int(g["num_nodes"]), 2 * int(g["num_pred"]) + 1
    t = util.coo_tensor(A, mode, "cuda")
    plan = GraphPlan(t, N, R)
    idx = g["coo_indices"]
    vals = g["coo_values_i8"] if mode == "ref_int8" else A.data
    _check_plan(plan, util.numpy_plan(idx[0], idx[1], vals, N, N, R))
    # pruning the zeros the int8 cast produced
    plan2 = GraphPlan(t, N, R, prune_zeros=True)
    _check_plan(plan2, util.numpy_plan(idx[0], idx[1], vals, N, N, R, prune=True))
    if mode == "ref_int8":
        assert plan2.nnz == int((g["coo_values_i8"] != 0).sum())


def test_plan_on_skewed_random_graph_and_errors():
    from mrgcn_amd._lib import MrgcnError
    rng = np.random.default_rng(5)
    N, R, num_rows = 3000, 7, 3000
    rows, cols, vals = _random_graph(rng, num_rows, N, R, 40000, hub_rows=3, hub_len=2500, hub_cols=3)
    plan = _plan_from_coo(rows, cols, vals, num_rows, N, R)
    _check_plan(plan, util.numpy_plan(rows, cols, vals, num_rows, N, R))
    assert plan.long_rows >= 3 and plan.long_cols >= 3 and plan.max_row_nnz >= 2500
    # ragged: empty graph, rectangular slice, out-of-range index
    empty = _plan_from_coo(np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32), 10, 10, 3)
    assert empty.nnz == 0 and empty.ncols == 0
    rect = _plan_from_coo(rows[rows < 100], cols[rows < 100], vals[rows < 100], 100, N, R)
    _check_plan(rect, util.numpy_plan(rows[rows < 100], cols[rows < 100], vals[rows < 100], 100, N, R))
    with pytest.raises(MrgcnError):
        _plan_from_coo(np.array([0, 11]), np.array([0, 1]), np.ones(2, np.float32), 10, 10, 3)
    with pytest.raises(MrgcnError):
        _plan_from_coo(np.array([0, 1]), np.array([0, 30]), np.ones(2, np.float32), 10, 10, 3)


def test_plan_bit_exact_on_multi_band_graph():
    """AM/5-shaped graph: 333k nodes = 3 node bands of the transform order, 8 000 long rows."""
    from mrgcn_amd import synth
    g = synth.make_graph("am", seed=4, scale=0.2)
    plan = _plan_from_coo(g.rows, g.cols, g.vals, g.num_nodes, g.num_nodes, g.num_relations)
    assert g.num_nodes > 2 * util.NODE_BAND
    _check_plan(plan, util.numpy_plan(g.rows, g.cols, g.vals, g.num_nodes, g.num_nodes, g.num_relations))


FEATS = [1, 2, 3, 4, 5, 8, 10, 11, 12, 16, 17, 31, 64, 200, 300]


@pytest.fixture(scope="module")
def skewed():
    rng = np.random.default_rng(11)
    N, R, num_rows = 2000, 5, 2000
    rows, cols, vals = _random_graph(rng, num_rows, N, R, 30000, hub_rows=2, hub_len=1700, hub_cols=2)
    plan = _plan_from_coo(rows, cols, vals, num_rows, N, R)
    A = sp.csr_matrix((vals.astype(np.float64), (rows, cols)), shape=(num_rows, R * N))
    return plan, A, util.numpy_plan(rows, cols, vals, num_rows, N, R), rng


@pytest.mark.parametrize("F", FEATS)
def test_spmm_literal_compact_transposed(skewed, F):
    """Tolerance: fp32 products accumulated in a different order than scipy's float64:
    |err| <= 1e-4 * (1 + |ref|) with rows of up to ~1700 terms."""
    from mrgcn_amd import _lib as L
    plan, A, ref, rng = skewed
    RN = A.shape[1]
    D = rng.standard_normal((RN, F)).astype(np.float32)
    Dg = torch.from_numpy(D).cuda()
    Y_ref = A @ D.astype(np.float64)
    Y = plan.spmm(L.VIEW_LITERAL, Dg).cpu().numpy()
    np.testing.assert_allclose(Y, Y_ref, rtol=1e-4, atol=1e-4)
    # compact operand with padded leading dimension
    for ld in sorted({F, (F + 3) // 4 * 4, F + 5}):
        M = np.zeros((plan.ncols, ld), dtype=np.float32)
        M[ref["mpos"], :F] = D[ref["ulcol"]]  # compact column c is stored at row mpos[c]
        M[:, F:] = 1e30  # padding must never leak into the result
        Yc = plan.spmm(L.VIEW_COMPACT, torch.from_numpy(M).cuda(), F=F).cpu().numpy()
        np.testing.assert_allclose(Yc, Y_ref, rtol=1e-4, atol=1e-4)
    # bias + ReLU epilogue
    b = rng.standard_normal(F).astype(np.float32)
    Yb = plan.spmm(L.VIEW_LITERAL, Dg, bias=torch.from_numpy(b).cuda(), relu=True).cpu().numpy()
    np.testing.assert_allclose(Yb, np.maximum(Y_ref + b, 0), rtol=1e-4, atol=1e-4)
    # transposed product = autograd of the above
    dY = rng.standard_normal((A.shape[0], F)).astype(np.float32)
    dM_ref = (A.T @ dY.astype(np.float64))[ref["ulcol"]]
    dM = plan.spmm(L.VIEW_TRANSPOSED, torch.from_numpy(dY).cuda()).cpu().numpy()
    np.testing.assert_allclose(dM, dM_ref, rtol=1e-4, atol=1e-4)
    # scattered into the literal (R*N) x F gradient
    dD = torch.zeros((RN, F), device="cuda")
    ptr, _ = plan.array_ptr(L.ARR_ULCOL)
    plan.spmm(L.VIEW_TRANSPOSED, torch.from_numpy(dY).cuda(), out=dD, out_index=ptr)
    np.testing.assert_allclose(dD.cpu().numpy(), A.T @ dY.astype(np.float64), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("F", [1, 10, 16, 33])
def test_lean_plan_products(F):
    """MRGCN_PLAN_LEAN (the quick build for mini-batch slices): identity row / operand orders, the three views
    multiply like the full plan's (hub rows and hub columns included), the structure arrays equal the numpy plan's."""
    from mrgcn_amd import _lib as L
    rng = np.random.default_rng(5)
    N, R, num_rows = 1500, 4, 700
    rows, cols, vals = _random_graph(rng, num_rows, N, R, 20000, hub_rows=2, hub_len=1300, hub_cols=2)
    plan = _plan_from_coo(rows, cols, vals, num_rows, N, R, lean=True)
    ref = util.numpy_plan(rows, cols, vals, num_rows, N, R)
    assert plan.lean and plan.nop == plan.ncols == ref["ncols"] and plan.n_rep == 0
    for name in ("rowptr", "lcol", "ccol", "cptr", "crow", "ulcol", "urel", "unode"):
        np.testing.assert_array_equal(plan.export(getattr(L, "ARR_" + name.upper())), ref[name], err_msg=name)
    np.testing.assert_array_equal(plan.export(L.ARR_MPOS), np.arange(plan.ncols))
    A = sp.csr_matrix((vals.astype(np.float64), (rows, cols)), shape=(num_rows, R * N))
    D = rng.standard_normal((R * N, F)).astype(np.float32)
    Y_ref = A @ D.astype(np.float64)
    np.testing.assert_allclose(plan.spmm(L.VIEW_LITERAL, torch.from_numpy(D).cuda()).cpu().numpy(), Y_ref, rtol=1e-4,
                               atol=1e-4)
    for ld in sorted({F, (F + 3) // 4 * 4}):
        M = np.full((plan.ncols, ld), 1e30, dtype=np.float32)
        M[:, :F] = D[ref["ulcol"]]   # operand row c = compact column c
        b = rng.standard_normal(F).astype(np.float32)
        Yc = plan.spmm(L.VIEW_COMPACT, torch.from_numpy(M).cuda(), F=F, bias=torch.from_numpy(b).cuda(), relu=True)
        np.testing.assert_allclose(Yc.cpu().numpy(), np.maximum(Y_ref + b, 0), rtol=1e-4, atol=1e-4)
        Yp = plan.spmm(L.VIEW_COMPACT, torch.from_numpy(M).cuda(), F=F, padded_rows=True)
        np.testing.assert_allclose(Yp.cpu().numpy(), Y_ref, rtol=1e-4, atol=1e-4)
    dY = rng.standard_normal((num_rows, F)).astype(np.float32)
    dM = plan.spmm(L.VIEW_TRANSPOSED, torch.from_numpy(dY).cuda()).cpu().numpy()
    np.testing.assert_allclose(dM, (A.T @ dY.astype(np.float64))[ref["ulcol"]], rtol=1e-4, atol=1e-4)
    del plan   # (aliased arrays are released once)
    torch.cuda.synchronize()


@pytest.mark.parametrize("F", [3, 10, 11, 13])
def test_compact_product_into_rows_with_a_writable_pad(skewed, F):
    """MRGCN_SPMM_PAD_WRITABLE: the product may zero columns F .. 4*ceil(F/4)-1 of Y's rows (whole 16-byte
    stores) and nothing else; without the flag nothing outside [0, F) is touched; the values are bitwise those of
    the dense output either way (rows of all three classes, rows of several chunks)."""
    from mrgcn_amd import _lib as L
    plan, A, ref, rng = skewed
    rows, ld4 = A.shape[0], (F + 3) // 4 * 4
    M = torch.from_numpy(rng.standard_normal((plan.nop, ld4)).astype(np.float32)).cuda()
    M[:, F:] = 1e30
    b = torch.from_numpy(rng.standard_normal(F).astype(np.float32)).cuda()
    for bias, relu in ((None, False), (b, True)):
        dense = torch.full((rows, F), float("nan"), device="cuda")
        plan.spmm(L.VIEW_COMPACT, M, F=F, out=dense, bias=bias, relu=relu)
        assert torch.isfinite(dense).all()
        for ld, flag in ((ld4, True), (ld4, False), (ld4 + 4, True)):
            buf = torch.full((rows, ld), 7.0, device="cuda")
            plan.spmm(L.VIEW_COMPACT, M, F=F, out=buf[:, :F], bias=bias, relu=relu, pad_writable=flag)
            assert torch.equal(buf[:, :F], dense)
            if flag:
                assert (buf[:, F:ld4] == 0).all() and (buf[:, ld4:] == 7.0).all()
            else:
                assert (buf[:, F:] == 7.0).all()
        own = plan.spmm(L.VIEW_COMPACT, M, F=F, bias=bias, relu=relu, padded_rows=True)   # the plan's own buffer
        assert own.shape == (rows, F) and own.stride(0) == ld4 and torch.equal(own, dense)
        assert plan.spmm(L.VIEW_COMPACT, M, F=F, bias=bias, relu=relu).is_contiguous()


def test_relu_backward_on_row_strided_operands():
    from mrgcn_amd.functional import relu_bwd
    g = torch.Generator("cuda").manual_seed(4)
    for rows, F, ld in ((1, 1, 4), (1000, 10, 12), (777, 33, 36)):
        Y = torch.randn((rows, ld), device="cuda", generator=g)[:, :F]
        dY = torch.randn((rows, F), device="cuda", generator=g)
        assert torch.equal(relu_bwd(dY, Y), dY * (Y > 0))
        assert torch.equal(relu_bwd(dY, Y.contiguous()), dY * (Y > 0))


def test_spmm_is_deterministic_and_linear(skewed):
    from mrgcn_amd import _lib as L
    plan, A, ref, rng = skewed
    D1 = torch.randn((A.shape[1], 10), device="cuda")
    D2 = torch.randn((A.shape[1], 10), device="cuda")
    y1 = plan.spmm(L.VIEW_LITERAL, D1)
    assert torch.equal(y1, plan.spmm(L.VIEW_LITERAL, D1))  # bitwise reproducible (no atomics)
    y12 = plan.spmm(L.VIEW_LITERAL, D1 + D2)
    torch.testing.assert_close(y12, y1 + plan.spmm(L.VIEW_LITERAL, D2), rtol=1e-4, atol=1e-4)


def test_spmm_autograd_functions(skewed):
    from mrgcn_amd import functional as Fn
    plan, A, ref, rng = skewed
    D = torch.randn((A.shape[1], 6), device="cuda", requires_grad=True)
    bias = torch.randn(6, device="cuda", requires_grad=True)
    Y = Fn.spmm_literal(plan, D, bias=bias, relu=True)
    w = torch.randn_like(Y)
    (Y * w).sum().backward()
    At = torch.from_numpy(A.toarray().astype(np.float32)).cuda()
    D2 = D.detach().clone().requires_grad_(True)
    b2 = bias.detach().clone().requires_grad_(True)
    Y2 = torch.relu(At @ D2 + b2)
    (Y2 * w).sum().backward()
    torch.testing.assert_close(Y, Y2, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(D.grad, D2.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(bias.grad, b2.grad, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("F", [4, 7, 10, 11, 12, 16])
def test_entry_sliced_transposed_product_opt_in(skewed, F):
    """`spmm_t_seg` (opt-in): the general TRANSPOSED product with a slot per four ENTRIES and a segmented sum — rows
    that span slots, super-rounds and waves (hub columns of > 1 000 entries), dense and padded outputs — against the
    float64 product; two runs are bitwise equal (no atomics)."""
    from mrgcn_amd import _lib as L
    plan, A, ref, rng = skewed
    dY = torch.from_numpy(rng.standard_normal((A.shape[0], F)).astype(np.float32)).cuda()
    ref64 = (A.T @ dY.double().cpu().numpy())[ref["ulcol"]]
    old = L.set_config(spmm_t_seg=1)
    try:
        for ld in sorted({F, (F + 3) // 4 * 4}):
            out = torch.full((plan.ncols, ld), 9.0, device="cuda")
            got = plan.spmm(L.VIEW_TRANSPOSED, dY, F=F, out=out[:, :F] if ld != F else out)
            np.testing.assert_allclose(got.cpu().numpy()[:, :F], ref64, rtol=1e-4, atol=1e-4)
            again = plan.spmm(L.VIEW_TRANSPOSED, dY, F=F)
            assert torch.equal(again, got.contiguous())
            if ld != F:
                assert float(out[:, F:].min()) == 9.0 and float(out[:, F:].max()) == 9.0   # the pad is the caller's
    finally:
        L.set_config(**old)


@pytest.mark.parametrize("F", [1, 3, 10, 11, 16, 20])
@pytest.mark.parametrize("zero_frac", [0.0, 0.9, 0.997, 1.0])
def test_transposed_product_with_zero_operand_rows(skewed, F, zero_frac):
    """mrgcn_spmm_transposed_live_f32 (the backward product when few rows of dY carry gradient):
    bitwise the result of the general transposed product, and a flag per compact column that is
    0 only where the result row is all zeros.  Hub columns (> 32 entries) take the split-row path."""
    from mrgcn_amd import _lib as L
    plan, A, ref, rng = skewed
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    dY = rng.standard_normal((A.shape[0], F)).astype(np.float32)
    dead = rng.random(A.shape[0]) < zero_frac
    dY[dead] = 0.0
    dY[dead & (rng.random(A.shape[0]) < 0.5)] = -0.0
    dYg = torch.from_numpy(dY).cuda()
    want = plan.spmm(L.VIEW_TRANSPOSED, dYg)  # [ncols, F]
    for ld in sorted({F, (F + 3) // 4 * 4}):
        got = torch.full((plan.ncols, ld), 9.0, device="cuda")
        row_live = torch.empty(int(lib.mrgcn_spmm_transposed_live_scratch(plan.handle)), dtype=torch.uint8,
                               device="cuda")
        col_live = torch.full((plan.ncols,), 7, dtype=torch.uint8, device="cuda")
        n_live = torch.full((1,), 77, dtype=torch.int32, device="cuda")
        L.check(lib.mrgcn_spmm_transposed_live_f32(plan.handle, dYg.data_ptr(), F, F, got.data_ptr(), ld,
                                                   row_live.data_ptr(), col_live.data_ptr(), n_live.data_ptr(), 1, s))
        assert int(n_live) == (int((~dead).sum()) if F <= 16 else -1)
        short_t = torch.from_numpy(np.diff(ref["cptr"]) <= 32).cuda()
        assert torch.equal(got[:, :F][short_t], want[short_t]), f"ld={ld}"
        torch.testing.assert_close(got[:, :F], want, rtol=1e-5, atol=1e-5)  # split rows: order may differ
        if F <= 16:  # (wider layers run the general product and never touch the scratch)
            np.testing.assert_array_equal(row_live[:plan.num_rows].cpu().numpy(),
                                          (~dead).astype(np.uint8) if zero_frac else 1)
        cl = col_live.cpu().numpy()
        assert set(np.unique(cl)) <= {0, 1}
        nz = (want != 0).any(1).cpu().numpy()
        assert not (nz & (cl == 0)).any()          # a column with gradient is never flagged dead
        if F <= 16:                                # flag = "some contributing row is live"
            csc = abs(sp.csc_matrix(A)[:, ref["ulcol"]])
            reach = (csc.T @ (~dead).astype(np.float64)) > 0 if zero_frac else np.ones(plan.ncols, bool)
            np.testing.assert_array_equal(cl, reach.astype(np.uint8))
    if F <= 16:
        # write_dead_rows = 0: rows flagged dead are left alone (consumers go by the flags)
        ld = (F + 3) // 4 * 4
        got = torch.full((plan.ncols, ld), 9.0, device="cuda")
        L.check(lib.mrgcn_spmm_transposed_live_f32(plan.handle, dYg.data_ptr(), F, F, got.data_ptr(), ld,
                                                   row_live.data_ptr(), col_live.data_ptr(), 0, 0, s))
        lv = col_live.bool()
        torch.testing.assert_close(got[:, :F][lv], want[lv], rtol=1e-5, atol=1e-5)
        short_dead = (~lv) & torch.from_numpy(np.diff(ref["cptr"]) <= 32).cuda()
        assert (got[short_dead] == 9.0).all()
    # NaN rows are live
    dYg[0, F - 1] = float("nan")
    row_live = torch.empty(plan.num_rows, dtype=torch.uint8, device="cuda")
    L.check(lib.mrgcn_rows_nonzero_f32(dYg.data_ptr(), F, F, plan.num_rows, row_live.data_ptr(), s))
    assert int(row_live[0]) == 1


@pytest.mark.parametrize("K,F,need_dX", [(7, 10, True), (155, 10, False), (10, 11, True), (40, 33, True),
                                         (200, 16, True), (155, 10, True), (256, 10, True), (65, 3, True),
                                         (129, 11, True)])
@pytest.mark.parametrize("live_frac", [0.0, 0.08, 1.0])
def test_transform_backward_over_live_columns(skewed, K, F, need_dX, live_frac):
    """mrgcn_rel_transform_bwd_live_f32 == mrgcn_rel_transform_bwd_f32 when the rows of dM that
    the flags call dead are zeros (dW: MFMA accumulation order differs -> 1e-5 relative)."""
    from mrgcn_amd import _lib as L
    plan, A, ref, rng = skewed
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    N, R = plan.num_nodes, plan.num_relations
    ld = (F + 3) // 4 * 4
    dM = rng.standard_normal((plan.ncols, ld)).astype(np.float32)
    live = rng.random(plan.ncols) < live_frac
    dM[~live] = 0.0
    dM_poison = dM.copy()
    dM_poison[~live] = np.nan  # with flags, dead rows are never read (or zeroed first by a fallback)
    X = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).cuda()
    W = torch.from_numpy(rng.standard_normal((R, K, F)).astype(np.float32)).cuda()
    dMg = torch.from_numpy(dM).cuda()
    liveg = torch.from_numpy(live.astype(np.uint8)).cuda()
    nws = int(lib.mrgcn_rel_transform_bwd_workspace(plan.handle, K, F, int(need_dX), 1))
    outs = []
    dMpg = torch.from_numpy(dM_poison).cuda()
    for flags in (0, liveg.data_ptr()):
        ws = torch.full((max(nws, 1),), float("nan"), device="cuda")  # dead rows of Z must never be read
        dX = torch.full((N, K), 5.0, device="cuda")
        dW = torch.full((R, K, F), 5.0, device="cuda")
        L.check(lib.mrgcn_rel_transform_bwd_live_f32(plan.handle, (dMpg if flags else dMg).data_ptr(), ld, flags,
                                                     X.data_ptr(), K, K,
                                                     W.data_ptr(), F, dX.data_ptr() if need_dX else 0, K,
                                                     dW.data_ptr(), ws.data_ptr(), nws, s))
        outs.append((dX.cpu().numpy(), dW.cpu().numpy()))
    (dX0, dW0), (dX1, dW1) = outs
    # float64 restatement (autograd of graph.py:93-94): dX[j] = sum_c dM[c] W[r_c]^T, dW[r] = sum_c X[j_c]^T dM[c] —
    # wide inputs take the dX pass in slices of 64 output columns on the matrix cores
    ul = ref["ulcol"]
    rel, node = ul // N, ul % N
    dM64, W64, X64 = dM[:, :F].astype(np.float64), W.cpu().numpy().astype(np.float64), X.cpu().numpy().astype(np.float64)
    dW_want = np.zeros((R, K, F))
    np.add.at(dW_want, rel, X64[node][:, :, None] * dM64[:, None, :])
    np.testing.assert_allclose(dW0, dW_want, rtol=1e-4, atol=1e-4 * (np.abs(dW_want).max() + 1e-30))
    if need_dX:
        dX_want = np.zeros((N, K))
        np.add.at(dX_want, node, np.einsum("cf,ckf->ck", dM64, W64[rel]))
        np.testing.assert_allclose(dX0, dX_want, rtol=1e-4, atol=1e-4 * (np.abs(dX_want).max() + 1e-30))
    np.testing.assert_allclose(dW1, dW0, rtol=1e-5, atol=1e-5 * (np.abs(dW0).max() + 1e-30))
    if need_dX:
        np.testing.assert_allclose(dX1, dX0, rtol=1e-5, atol=1e-5 * (np.abs(dX0).max() + 1e-30))
    if live_frac == 0.0:
        assert not dW1.any() and (not need_dX or not dX1.any())


@pytest.mark.parametrize("F,ld", [(10, 12), (11, 12), (16, 16), (3, 4)])
def test_compact_product_on_an_operand_with_replicas(F, ld):
    """MRGCN_PLAN_REPLICATE: every entry of a column read by fewer than 16 rows owns an operand row in its reader's
    stream; the producers write a column once (MPOS), mrgcn_operand_replicate fills the copies, and the product
    equals the plain one.  Entries of non-hot columns reference consecutive operand rows in processing order."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphPlan
    rng = np.random.default_rng(F)
    N, R = 3000, 5
    rows, cols, vals = _random_graph(rng, N, N, R, 12 * N, hub_rows=3, hub_len=2500, hub_cols=4)
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    base, plan = GraphPlan(A, N, R, replicate=False), GraphPlan(A, N, R, replicate=True)
    assert base.n_rep == 0 and base.nop == base.ncols
    cnt = np.diff(plan.export(L.ARR_CPTR))
    n_hot = int((cnt >= util.HOT_MIN_REFS).sum())
    assert plan.nop == n_hot + int(cnt[cnt < util.HOT_MIN_REFS].sum())
    assert plan.n_rep == int((cnt[cnt < util.HOT_MIN_REFS] - 1).sum())
    mcol, mpos = plan.export(L.ARR_MCOL), plan.export(L.ARR_MPOS)
    stream = mcol[mcol >= n_hot]
    assert np.array_equal(stream, n_hot + np.arange(len(stream)))          # read front to back
    assert np.array_equal(np.sort(mpos[cnt >= util.HOT_MIN_REFS]), np.arange(n_hot))
    Mc = rng.standard_normal((plan.ncols, F)).astype(np.float32)
    M = torch.full((plan.nop, ld), float("nan"), device="cuda")
    M[torch.from_numpy(mpos.astype(np.int64)).cuda(), :F] = torch.from_numpy(Mc).cuda()
    plan.replicate(M)
    assert not torch.isnan(M[:, :F]).any()                                # every operand row was produced
    Y = plan.spmm(L.VIEW_COMPACT, M, F=F).cpu().numpy()
    Mb = torch.zeros((base.ncols, ld), device="cuda")
    Mb[torch.from_numpy(base.export(L.ARR_MPOS).astype(np.int64)).cuda(), :F] = torch.from_numpy(Mc).cuda()
    Yb = base.spmm(L.VIEW_COMPACT, Mb, F=F).cpu().numpy()
    A_csr = sp.csr_matrix((vals.astype(np.float64), (rows, cols)), shape=(N, R * N))
    ref = A_csr[:, plan.export(L.ARR_ULCOL).astype(np.int64)] @ Mc.astype(np.float64)
    np.testing.assert_allclose(Y, ref, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(Y, Yb, rtol=1e-5, atol=1e-5)


def test_model_on_a_plan_with_replicas_matches_the_golden():
    """A whole golden epoch on a plan built with operand replicas (layers pick the plan up from the adjacency)."""
    from mrgcn_amd.plan import GraphPlan
    from mrgcn_amd.train import ClipAdam, train_step
    name = "rgcn_smoke_ft_b5_norm_f32"
    c = util.load_case(name)
    model, dims = util.build_rgcn_from_case(c, "cuda")
    util.load_state_from_case(model, c)
    model = model.cuda()
    g, A_csr = util.load_graph(util.graph_of_case(name))
    A = util.coo_tensor(A_csr, str(c["value_mode"]), "cuda")
    A._mrgcn_plan = GraphPlan(A, int(c["meta.num_nodes"]), int(c["meta.R"]), replicate=True)
    assert A._mrgcn_plan.n_rep > 0
    X = torch.from_numpy(c["X"]).cuda()
    with torch.no_grad():
        np.testing.assert_allclose(model(X, A).cpu().numpy(), c["logits"], rtol=1e-4, atol=1e-4)
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0)
    idx, tgt = torch.from_numpy(c["labels_idx"]).cuda(), torch.from_numpy(c["labels_y"]).cuda()
    for step in range(1, int(c["meta.n_adam"]) + 1):
        loss = train_step(model, lambda: model(X, A), idx, tgt, opt)
        np.testing.assert_allclose(float(loss), float(c[f"loss_step{step}"]), rtol=2e-4, atol=2e-5)


def test_operand_order_keeps_reread_rows_off_the_straddling_slots():
    """plan.hip::k_avoid_straddle on a graph with many multi-reader columns: the operand order stays a permutation,
    every aligned group of 32 positions holds the same columns as the plain hot / first-touch order (a local
    exchange), and no column with several readers sits on a slot whose 48-byte row straddles a 128-byte line while
    its group still has a single-reader column on a straddle-free slot; MRGCN_AVOID_STRADDLE=0 gives the plain order.
    The product is the same on both."""
    import os
    import subprocess
    import sys
    from mrgcn_amd import _lib as L
    from mrgcn_amd import synth
    from mrgcn_amd.plan import GraphPlan
    g = synth.make_graph("am", seed=2, scale=0.02)
    N, R = g.num_nodes, g.num_relations
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals), (N, R * N)).cuda()
    plan = GraphPlan(A, N, R)
    mpos = plan.export(L.ARR_MPOS).astype(np.int64)
    cptr = plan.export(L.ARR_CPTR).astype(np.int64)
    nc = plan.ncols
    assert np.array_equal(np.sort(mpos), np.arange(nc))
    readers = np.diff(cptr)
    at = np.empty(nc, dtype=np.int64)
    at[mpos] = np.arange(nc)                     # column at each position
    multi = (readers[at] > 1)[: nc // 32 * 32].reshape(-1, 32)
    pos = np.arange(32, dtype=np.int64)
    bad = (pos * 48) // 128 != (pos * 48 + 47) // 128
    single_elsewhere = (~multi[:, ~bad]).any(1)
    n_hot = int((readers >= util.HOT_MIN_REFS).sum())
    inner = np.ones(len(multi), dtype=bool)
    inner[n_hot // 32] = n_hot % 32 == 0         # (the group that holds the end of the hot region stays as it is)
    assert not (multi[:, bad].any(1) & single_elsewhere & inner).any()
    assert multi.any()                            # the case is not vacuous
    code = ("import numpy as np, torch\n"
            "from mrgcn_amd import synth, _lib as L\nfrom mrgcn_amd.plan import GraphPlan\n"
            "g = synth.make_graph('am', seed=2, scale=0.02)\nN, R = g.num_nodes, g.num_relations\n"
            "A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals), (N, R * N)).cuda()\n"
            "p = GraphPlan(A, N, R)\nnp.save('/tmp/_mpos_plain.npy', p.export(L.ARR_MPOS))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", code], check=True, cwd=root, env=dict(os.environ, MRGCN_AVOID_STRADDLE="0"))
    plain = np.load("/tmp/_mpos_plain.npy").astype(np.int64)
    assert not np.array_equal(plain, mpos)
    assert np.array_equal(plain // 32, mpos // 32)   # every column stayed inside its group of 32
    M = torch.randn((nc, 12), device="cuda")
    Y = plan.spmm(L.VIEW_COMPACT, M, F=10).cpu().numpy()
    # the same product with the operand rows laid out in the plain order
    Mp = torch.empty_like(M)
    Mp[torch.from_numpy(plain).cuda()] = M[torch.from_numpy(mpos).cuda()]
    import scipy.sparse as sp
    ccol = plan.export(L.ARR_CCOL).astype(np.int64)
    rowidx = plan.export(L.ARR_ROWIDX).astype(np.int64)
    val = plan.export(L.ARR_VAL).astype(np.float64)
    want = sp.csr_matrix((val, (rowidx, mpos[ccol])), shape=(N, nc)) @ M[:, :10].double().cpu().numpy()
    np.testing.assert_allclose(Y, want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("F", [10, 16, 3])
def test_rows_of_several_chunks_finished_inside_the_product_equal_the_two_pass_form(F):
    """k_spmm3 finishes a row cut into several chunks in the wave that delivers its last partial sum (arrival counter
    per row, agent-scope stores / loads of the partial sums across the XCDs).  The summation order is fixed, so the
    result must be bitwise that of the two-pass form (MRGCN_SPMM_TWO_PASS: a second launch adds the partials) —
    for every launch of a back-to-back series whose operands change from launch to launch (a stale partial sum of
    the previous launch would show), while another stream keeps the memory system busy."""
    from mrgcn_amd import _lib as L
    rng = np.random.default_rng(5)
    N, R, num_rows = 60000, 3, 60000
    RN = R * N
    lens = np.concatenate([rng.integers(129, 1500, 1500), rng.integers(1500, 9000, 40), [70000],
                           rng.integers(33, 128, 500)])
    hub = rng.choice(num_rows, len(lens), replace=False)
    rows = np.concatenate([rng.integers(0, num_rows, 200000)] + [np.full(n, h) for n, h in zip(lens, hub)])
    cols = np.concatenate([rng.integers(0, RN, 200000)] + [rng.choice(RN, n, replace=False) for n in lens])
    key = np.unique(rows.astype(np.int64) * RN + cols)
    rows, cols = key // RN, key % RN
    vals = rng.standard_normal(len(rows)).astype(np.float32)
    plan = _plan_from_coo(rows, cols, vals, num_rows, N, R)
    ld = (F + 3) // 4 * 4
    g = torch.Generator("cuda").manual_seed(1)
    Ms = [torch.randn((plan.nop, ld), device="cuda", generator=g) for _ in range(3)]
    bias = torch.randn(F, device="cuda", generator=g)
    want = [plan.spmm(L.VIEW_COMPACT, M, F=F, bias=bias, relu=True, two_pass=True) for M in Ms]
    A = sp.csr_matrix((vals.astype(np.float64), (rows, cols)), shape=(num_rows, RN))
    ref = util.numpy_plan(rows, cols, vals, num_rows, N, R)
    D = np.zeros((RN, F))
    D[ref["ulcol"]] = Ms[0].cpu().numpy()[ref["mpos"], :F]
    np.testing.assert_allclose(want[0].cpu().numpy(), np.maximum(A @ D + bias.cpu().numpy(), 0), rtol=1e-4, atol=2e-3)
    side = torch.cuda.Stream()
    big = torch.empty(1 << 27, device="cuda")
    outs = []
    with torch.cuda.stream(side):
        for _ in range(6):
            big.add_(1.0)
    for it in range(60):
        outs.append(plan.spmm(L.VIEW_COMPACT, Ms[it % 3], F=F, bias=bias, relu=True))
    torch.cuda.synchronize()
    for it, y in enumerate(outs):
        assert torch.equal(y, want[it % 3]), f"launch {it}"
    # into padded rows as well (the pad of a finished row is zeroed by the finishing wave)
    if F % 4:
        buf = torch.full((num_rows, ld), 7.0, device="cuda")
        plan.spmm(L.VIEW_COMPACT, Ms[1], F=F, out=buf[:, :F], bias=bias, relu=True, pad_writable=True)
        assert torch.equal(buf[:, :F], want[1]) and (buf[:, F:] == 0).all()


@pytest.mark.parametrize("row_bytes", [(40,), (40, 44), (44, 24), (64, 128, 32, 800)])
def test_plan_hinted_for_packed_operand_rows(row_bytes):
    """mrgcn_plan_create_hinted: the operand order keeps re-read columns off the positions whose row would straddle
    a 128-byte line for EVERY announced row size (bit-exact against the numpy plan); sizes that never straddle
    (divisors of 128; rows of 128 bytes and more always span lines) leave the plain order; the product on packed rows (ld = F: the last 16-byte vector
    of a row overlaps its neighbour's) equals scipy's for every F."""
    from mrgcn_amd import _lib as L
    rng = np.random.default_rng(3)
    N, R, num_rows = 4000, 4, 4000
    rows, cols, vals = _random_graph(rng, num_rows, N, R, 40000, hub_rows=2, hub_len=900, hub_cols=3)
    plan = _plan_from_coo(rows, cols, vals, num_rows, N, R, row_bytes=row_bytes)
    eff = tuple(b for b in row_bytes if 128 % b and b < 128)
    ref = util.numpy_plan(rows, cols, vals, num_rows, N, R, row_bytes=eff)
    _check_plan(plan, ref)
    if eff:
        cnt = np.diff(ref["cptr"])
        at = np.empty(plan.ncols, dtype=np.int64)
        at[ref["mpos"]] = np.arange(plan.ncols)
        pos = np.arange(plan.ncols, dtype=np.int64)
        bad = np.zeros(plan.ncols, dtype=bool)
        for b in eff:
            bad |= (pos * b) // 128 != (pos * b + b - 1) // 128
        lukewarm = (cnt[at] > 1) & (cnt[at] < util.HOT_MIN_REFS)
        plain = util.numpy_plan(rows, cols, vals, num_rows, N, R, row_bytes=())
        at0 = np.empty(plan.ncols, dtype=np.int64)
        at0[plain["mpos"]] = np.arange(plan.ncols)
        assert (bad & lukewarm).sum() < ((cnt[at0] > 1) & (cnt[at0] < util.HOT_MIN_REFS) & bad).sum()
    A = sp.csr_matrix((vals.astype(np.float64), (rows, cols)), shape=(num_rows, R * N))
    for F in (4, 5, 7, 10, 11, 13, 15, 16):
        D = rng.standard_normal((R * N, F)).astype(np.float32)
        M = np.full((plan.ncols + 1, F), 1e30, dtype=np.float32)   # (nothing may be read past the last row)
        M[ref["mpos"]] = D[ref["ulcol"]]
        Mg = torch.from_numpy(M).cuda()[: plan.ncols]
        b = torch.from_numpy(rng.standard_normal(F).astype(np.float32)).cuda()
        want = np.maximum(A @ D.astype(np.float64) + b.cpu().numpy(), 0)
        Y = plan.spmm(L.VIEW_COMPACT, Mg, F=F, bias=b, relu=True)
        np.testing.assert_allclose(Y.cpu().numpy(), want, rtol=1e-4, atol=1e-4)
        Yp = plan.spmm(L.VIEW_COMPACT, Mg, F=F, bias=b, relu=True, padded_rows=True)
        assert torch.equal(Yp, Y)
        assert torch.equal(plan.spmm(L.VIEW_COMPACT, Mg, F=F, bias=b, relu=True, two_pass=True), Y)


def test_products_of_one_plan_on_two_streams_at_once_equal_the_serial_results():
    """include/mrgcn_hip.h: a plan keeps its product scratch (partial sums of split rows, arrival counters of the
    in-kernel finalize) per stream.  Two COMPACT products with different operands run side by side on two streams,
    many times over, on a graph with rows of several blocks; each must equal its own serial result bit for bit (a
    shared scratch would mix the partial sums of the two, or let one launch's last arriver consume the other's
    counter)."""
    from mrgcn_amd import _lib as L
    rng = np.random.default_rng(11)
    N, R, F = 6000, 7, 10
    rows, cols, vals = _random_graph(rng, N, N, R, 60000, hub_rows=6, hub_len=5000, hub_cols=2)
    plan = _plan_from_coo(rows, cols, vals, N, N, R, row_bytes=[40])
    assert plan.long_rows > 0
    Ma = torch.randn((plan.nop, F), device="cuda")
    Mb = torch.randn((plan.nop, F), device="cuda")
    ref_a = plan.spmm(L.VIEW_COMPACT, Ma, F=F).clone()
    ref_b = plan.spmm(L.VIEW_COMPACT, Mb, F=F).clone()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs_a, outs_b = [], []
    for _ in range(40):
        with torch.cuda.stream(sa):
            outs_a.append(plan.spmm(L.VIEW_COMPACT, Ma, F=F))
        with torch.cuda.stream(sb):
            outs_b.append(plan.spmm(L.VIEW_COMPACT, Mb, F=F))
    torch.cuda.synchronize()
    for ya, yb in zip(outs_a, outs_b):
        assert torch.equal(ya, ref_a) and torch.equal(yb, ref_b)
    # the transposed (split-column) product too
    dY = torch.randn((N, F), device="cuda")
    ref_t = plan.spmm(L.VIEW_TRANSPOSED, dY, F=F).clone()
    outs = []
    for _ in range(20):
        with torch.cuda.stream(sa):
            outs.append(plan.spmm(L.VIEW_TRANSPOSED, dY, F=F))
        with torch.cuda.stream(sb):
            outs.append(plan.spmm(L.VIEW_COMPACT, Mb, F=F))
    torch.cuda.synchronize()
    for i, y in enumerate(outs):
        assert torch.equal(y, ref_t if i % 2 == 0 else ref_b)
    plan.close()


def test_a_new_stream_cannot_take_its_first_product_inside_a_capture():
    from mrgcn_amd import _lib as L
    rng = np.random.default_rng(12)
    N, R, F = 500, 3, 10
    rows, cols, vals = _random_graph(rng, N, N, R, 3000)
    plan = _plan_from_coo(rows, cols, vals, N, N, R)
    M = torch.randn((plan.nop, F), device="cuda")
    plan.spmm(L.VIEW_COMPACT, M, F=F)               # the build stream's own set
    g = torch.cuda.CUDAGraph()
    with pytest.raises(L.MrgcnError, match="before capturing"):
        with torch.cuda.graph(g):                   # (torch captures on a side stream the plan has not seen)
            plan.spmm(L.VIEW_COMPACT, M, F=F)
    torch.cuda.synchronize()
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("F", [10, 40, 200])
def test_the_general_product_finishes_split_rows_in_kernel_like_the_two_pass_form(F):
    """k_spmm (LITERAL / TRANSPOSED views, and the COMPACT view beyond 16 features): a row of several chunks is summed
    by the wave that delivers its last partial sum, in chunk order (arrival counter per long row, agent-scope hand-off)
    — bitwise the two-pass form (MRGCN_SPMM_TWO_PASS), launch after launch with changing operands under memory load."""
    from mrgcn_amd import _lib as L
    rng = np.random.default_rng(9)
    N, R, num_rows = 20000, 3, 20000
    RN = R * N
    lens = np.concatenate([rng.integers(600, 3000, 60), [30000], rng.integers(33, 500, 300)])
    hub = rng.choice(num_rows, len(lens), replace=False)
    rows = np.concatenate([rng.integers(0, num_rows, 60000)] + [np.full(n, h) for n, h in zip(lens, hub)])
    cols = np.concatenate([rng.integers(0, RN, 60000)] + [rng.choice(RN, n, replace=False) for n in lens])
    # hub COLUMNS too (long rows of the transposed view)
    hc = rng.choice(RN, 20, replace=False)
    rows = np.concatenate([rows] + [rng.choice(num_rows, 2500, replace=False) for _ in hc])
    cols = np.concatenate([cols] + [np.full(2500, c) for c in hc])
    key = np.unique(rows.astype(np.int64) * RN + cols)
    rows, cols = key // RN, key % RN
    vals = rng.standard_normal(len(rows)).astype(np.float32)
    plan = _plan_from_coo(rows, cols, vals, num_rows, N, R, row_bytes=[4 * F] if F > 16 else None)
    assert plan.long_rows > 0 and plan.long_cols > 0
    g = torch.Generator("cuda").manual_seed(2)
    bias = torch.randn(F, device="cuda", generator=g)
    side = torch.cuda.Stream()
    big = torch.empty(1 << 26, device="cuda")
    for view, nrows_in in ((L.VIEW_LITERAL, RN), (L.VIEW_TRANSPOSED, num_rows), (L.VIEW_COMPACT, plan.nop)):
        if view == L.VIEW_COMPACT and F <= 16:
            continue  # (k_spmm3's own path: the test above)
        Ds = [torch.randn((nrows_in, F), device="cuda", generator=g) for _ in range(2)]
        kw = dict(bias=bias, relu=True) if view != L.VIEW_TRANSPOSED else {}
        want = [plan.spmm(view, D, F=F, two_pass=True, **kw) for D in Ds]
        with torch.cuda.stream(side):
            for _ in range(4):
                big.add_(1.0)
        outs = [plan.spmm(view, Ds[it % 2], F=F, **kw) for it in range(24)]
        torch.cuda.synchronize()
        for it, y in enumerate(outs):
            assert torch.equal(y, want[it % 2]), f"view {view} launch {it}"


@pytest.mark.parametrize("F", [1, 3, 4, 7, 8, 10, 11, 12, 16])
def test_literal_product_on_the_compact_views_row_classes(skewed, F):
    """LITERAL products of narrow layers run k_spmm3 on the compact view's class-major rows with literal columns
    (`spmm_literal_v3`; the index array is built by the first such call): equal to scipy and, to rounding, to the
    general kernel; bias / ReLU epilogue, padded and packed operand rows, a captured call included."""
    from mrgcn_amd import _lib as L
    plan, A, ref, rng = skewed
    RN = A.shape[1]
    D = rng.standard_normal((RN, F)).astype(np.float32)
    b = rng.standard_normal(F).astype(np.float32)
    want = np.maximum(A @ D.astype(np.float64) + b, 0)
    for ld in sorted({F, (F + 3) // 4 * 4}):
        Dg = torch.zeros((RN, ld), device="cuda")
        Dg[:, :F] = torch.from_numpy(D).cuda()
        got = {}
        for on in (0, 1):
            old = L.set_config(spmm_literal_v3=on)
            try:
                got[on] = plan.spmm(L.VIEW_LITERAL, Dg, F=F, bias=torch.from_numpy(b).cuda(), relu=True)
                assert torch.equal(got[on], plan.spmm(L.VIEW_LITERAL, Dg, F=F, bias=torch.from_numpy(b).cuda(), relu=True))
            finally:
                L.set_config(**old)
        for on in (0, 1):
            np.testing.assert_allclose(got[on].cpu().numpy(), want, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(got[1].cpu().numpy(), got[0].cpu().numpy(), rtol=1e-5, atol=1e-5)
    # inside a capture (the index array exists by now)
    Dg = torch.from_numpy(D).cuda()
    out = torch.empty((A.shape[0], F), device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.spmm(L.VIEW_LITERAL, Dg, out=out)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            plan.spmm(L.VIEW_LITERAL, Dg, out=out)
        out.zero_()
        g.replay()
    side.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), A @ D.astype(np.float64), rtol=1e-4, atol=1e-4)
