"""Gradient support (include/mrgcn_hip.h: mrgcn_support_*): its index arrays against a numpy restatement (bit-exact),
its products through the C ABI against dense float64 arithmetic, and the epoch that runs on it against the epoch on
the per-epoch marking path and against the float64 oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import util
from tests.test_gpu_plan_spmm import _plan_from_coo, _random_graph

pytestmark = pytest.mark.gpu


def _numpy_support(ref, row_flags, N, R):
    """live columns / kept entries / nodes of `row_flags` on the numpy plan `ref` (tests/util.numpy_plan)"""
    rowidx, ccol = ref["rowidx"].astype(np.int64), ref["ccol"].astype(np.int64)
    live_entry = row_flags[rowidx] != 0
    col_flags = np.zeros(ref["ncols"], dtype=np.uint8)
    col_flags[ccol[live_entry]] = 1
    lcol = np.nonzero(col_flags)[0].astype(np.int32)
    lpos = np.cumsum(col_flags) - col_flags
    cptr, crow, cval = ref["cptr"], ref["crow"], ref["cval"]
    lptr, lrow, lval = [0], [], []
    for c in lcol:
        for e in range(cptr[c], cptr[c + 1]):
            if row_flags[crow[e]]:
                lrow.append(crow[e])
                lval.append(cval[e])
        lptr.append(len(lrow))
    nptr = ref["nptr"].astype(np.int64)
    lpos_ext = np.concatenate([lpos, [len(lcol)]])
    nlptr = lpos_ext[nptr].astype(np.int32)
    node_flags = (np.diff(nlptr) > 0).astype(np.uint8)
    return dict(col_flags=col_flags, lcol=lcol, lrel=ref["urel"][lcol], nlptr=nlptr, node_flags=node_flags,
                lptr=np.asarray(lptr, dtype=np.int32), lrow=np.asarray(lrow, dtype=np.int32),
                lval=np.asarray(lval, dtype=np.float32), lnode=np.nonzero(node_flags)[0].astype(np.int32))


@pytest.mark.parametrize("seed,N,R,nnz,hubs,labelled", [(0, 300, 7, 2500, 0, 12), (1, 1500, 11, 20000, 3, 40),
                                                        (2, 200, 3, 600, 1, 200), (3, 4000, 9, 30000, 2, 1)])
def test_support_arrays_equal_the_numpy_restatement(seed, N, R, nnz, hubs, labelled):
    from mrgcn_amd import _lib as L
    rng = np.random.default_rng(seed)
    rows, cols, vals = _random_graph(rng, N, N, R, nnz, hub_rows=hubs, hub_len=min(400, N), hub_cols=hubs)
    plan = _plan_from_coo(rows, cols, vals, N, N, R)
    ref = util.numpy_plan(rows, cols, vals, N, N, R)
    flags = np.zeros(N, dtype=np.uint8)
    flags[rng.choice(N, labelled, replace=False)] = 1
    ft = torch.from_numpy(flags).cuda()
    sup = plan.support_for(ft)
    assert sup is plan.support_for(ft)  # kept under the identity of the flags tensor
    want = _numpy_support(ref, flags, N, R)
    assert (sup.L, sup.E, sup.NL) == (len(want["lcol"]), len(want["lrow"]), int(want["node_flags"].sum()))
    for name in ("col_flags", "node_flags", "lcol", "lrel", "nlptr", "lptr", "lrow", "lval", "lnode"):
        np.testing.assert_array_equal(sup.export(getattr(L, "SUP_" + name.upper())), want[name], err_msg=name)
    # the relation-major list: a permutation of the live numbers, relations rising inside a node band
    lperm = sup.export(L.SUP_LPERM)
    np.testing.assert_array_equal(np.sort(lperm), np.arange(sup.L))
    band = min(util.NODE_BAND, N)
    unode = ref["unode"][want["lcol"][lperm]].astype(np.int64)
    key = ((unode // band) * R + want["lrel"][lperm]) * band + unode % band
    assert np.all(np.diff(key) > 0)
    ft2 = ft.clone()
    assert plan.support_for(ft2) is not sup  # another tensor: another support
    plan.close()


@pytest.mark.parametrize("F", [4, 10, 11, 16])
def test_support_transposed_product_reads_the_flagged_rows_only(F):
    from mrgcn_amd import _lib as L
    rng = np.random.default_rng(5)
    N, R = 2500, 13
    rows, cols, vals = _random_graph(rng, N, N, R, 30000, hub_rows=2, hub_len=1500, hub_cols=3)
    plan = _plan_from_coo(rows, cols, vals, N, N, R)
    ref = util.numpy_plan(rows, cols, vals, N, N, R)
    flags = np.zeros(N, dtype=np.uint8)
    flags[rng.choice(N, 900, replace=False)] = 1  # (hub columns then keep hundreds of entries: the split-row path)
    sup = plan.support_for(torch.from_numpy(flags).cuda())
    dY = rng.standard_normal((N, F)).astype(np.float32)
    dYp = dY.copy()
    dYp[flags == 0] = np.nan  # rows outside the set must never be read
    ld = (F + 3) // 4 * 4
    dM = torch.full((sup.L, ld), 7.0, device="cuda")
    d = torch.from_numpy(dYp).cuda()
    L.check(L.load().mrgcn_support_spmm_t_f32(sup.handle, d.data_ptr(), d.stride(0), F, dM.data_ptr(), ld,
                                              torch.cuda.current_stream().cuda_stream))
    want = _numpy_support(ref, flags, N, R)
    exp = np.zeros((sup.L, F))
    for k in range(sup.L):
        for e in range(want["lptr"][k], want["lptr"][k + 1]):
            exp[k] += float(want["lval"][e]) * dY[want["lrow"][e]].astype(np.float64)
    got = dM.cpu().numpy()[:, :F]
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-5)
    plan.close()


def _epoch_runs(case_name, support, steps=3, row_sparse=None):
    import mrgcn_amd.functional as Fn
    from mrgcn_amd.train import ClipAdam, train_step
    from tests.test_gpu_layers import _adjacency
    c = util.load_case(case_name)
    model, dims = util.build_rgcn_from_case(c, "cuda")
    util.load_state_from_case(model, c)
    model = model.cuda()
    At = _adjacency(c, case_name)
    X = None if bool(c["meta.featureless"]) else torch.from_numpy(c["X"]).cuda()
    idx = torch.from_numpy(c["labels_idx"]).cuda()
    tgt = torch.from_numpy(c["labels_y"]).cuda()
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0)
    prev, Fn._SUPPORT = Fn._SUPPORT, support
    try:
        losses = [float(train_step(model, lambda: model(X, At), idx, tgt, opt, row_sparse=row_sparse))
                  for _ in range(steps)]
    finally:
        Fn._SUPPORT = prev
    return losses, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


def _cases():
    return [n for n in util.rgcn_cases() if "labels_idx" in util.load_case(n).files]


@pytest.mark.parametrize("case_name", _cases())
@pytest.mark.parametrize("row_sparse", [None, False])
def test_epochs_on_the_support_equal_epochs_on_the_marking_path(case_name, row_sparse):
    l1, p1 = _epoch_runs(case_name, True, row_sparse=row_sparse)
    l0, p0 = _epoch_runs(case_name, False, row_sparse=row_sparse)
    np.testing.assert_allclose(l1, l0, rtol=1e-5, atol=1e-6)
    for k in p0:
        # (Adam's first steps are sign-like: a gradient entry whose sign flips between two summation orders moves
        # by 2 lr; everything else agrees to rounding)
        diff = np.abs(p1[k] - p0[k])
        assert (diff > 2e-5).mean() <= 2e-3, (k, float(diff.max()))


@pytest.mark.parametrize("row_sparse", [None])
def test_row_adam_forms_leave_the_same_bits(row_sparse):
    """The fused row Adam on a support as a ONE-SHOT grid (`adam_once` = 1, 2 or 4 list entries per wave; round 6) and as
    the persistent list kernel (`adam_once=0`) run the same fmaf chain per element: three epochs leave bitwise equal
    parameters and moments."""
    from mrgcn_amd import _lib as L
    name = [n for n in _cases() if "smoke" in n and "ft" in n and "b5" in n][0]
    runs = {}
    for once in (0, 1, 2, 4):
        old = L.set_config(adam_once=once)
        try:
            runs[once] = _epoch_runs(name, True, steps=3, row_sparse=row_sparse)
        finally:
            L.set_config(**old)
    l0, p0 = runs[0]
    for once in (1, 2, 4):
        l1, p1 = runs[once]
        assert l1 == l0, (once, l1, l0)
        for k in p0:
            assert np.array_equal(p1[k], p0[k]), (once, k)


def test_the_default_epoch_runs_its_backward_on_supports(monkeypatch):
    """a labelled model's train_step takes the support path in every layer: the C entry points are called, the
    per-epoch marking product is not"""
    import mrgcn_amd.functional as Fn
    from mrgcn_amd import _lib as L
    assert Fn._SUPPORT
    lib = L.load()
    seen = []

    class _Spy:
        def __init__(self, real):
            self._real = real

        def __getattr__(self, name):
            fn = getattr(self._real, name)
            if name.startswith("mrgcn_support_") or name.startswith("mrgcn_spmm_transposed_live"):
                def wrapped(*a, **k):
                    seen.append(name)
                    return fn(*a, **k)
                return wrapped
            return fn
    monkeypatch.setattr(L, "_lib", _Spy(lib))
    name = [n for n in _cases() if "smoke" in n][0]
    _epoch_runs(name, True, steps=2)
    assert "mrgcn_support_spmm_t_f32" in seen
    assert not any(n.startswith("mrgcn_spmm_transposed_live") for n in seen), seen


# ---- forward supports: a mini-batch layer as a masked pass over the full plan (csrc/masked.hip) -------------------
def _dense(rows, cols, vals, N, R):
    A = np.zeros((N, R * N), dtype=np.float64)
    np.add.at(A, (rows, cols), vals.astype(np.float64))
    return A


@pytest.mark.parametrize("seed,N,R,nnz,hubs,sample", [(0, 300, 7, 2500, 0, 12), (1, 1500, 11, 20000, 3, 40),
                                                      (2, 200, 3, 600, 1, 200), (3, 4000, 9, 30000, 2, 1)])
def test_forward_support_arrays_and_products(seed, N, R, nnz, hubs, sample):
    """FROW / FPTR / FCOL / FVAL = A[sample] over live columns (bit-exact against numpy); the forward and transposed
    products through the C ABI, with the stored values and with the all-ones slice, against dense float64."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphSupport
    lib = L.load()
    rng = np.random.default_rng(seed)
    rows, cols, vals = _random_graph(rng, N, N, R, nnz, hub_rows=hubs, hub_len=min(400, N), hub_cols=hubs)
    plan = _plan_from_coo(rows, cols, vals, N, N, R)
    ref = util.numpy_plan(rows, cols, vals, N, N, R)
    flags = np.zeros(N, dtype=np.uint8)
    flags[rng.choice(N, sample, replace=False)] = 1
    sup = GraphSupport(plan, torch.from_numpy(flags).cuda(), forward=True)
    want = _numpy_support(ref, flags, N, R)
    frow = np.nonzero(flags)[0]
    assert (sup.NR, sup.L, sup.NL) == (len(frow), len(want["lcol"]), len(want["lnode"]))
    np.testing.assert_array_equal(sup.export(L.SUP_FROW), frow)
    rank = np.full(N, -1, dtype=np.int32)
    rank[frow] = np.arange(len(frow))
    np.testing.assert_array_equal(sup.export(L.SUP_ROWRANK), rank)
    # the flagged rows of the plan's CSR, columns by live number
    lpos = np.cumsum(want["col_flags"]) - want["col_flags"]
    rowptr, ccol, val = ref["rowptr"], ref["ccol"], ref["val"]
    fptr, fcol, fval = [0], [], []
    for i in frow:
        fcol.extend(lpos[ccol[rowptr[i]:rowptr[i + 1]]])
        fval.extend(val[rowptr[i]:rowptr[i + 1]])
        fptr.append(len(fcol))
    np.testing.assert_array_equal(sup.export(L.SUP_FPTR), np.asarray(fptr, dtype=np.int32))
    np.testing.assert_array_equal(sup.export(L.SUP_FCOL), np.asarray(fcol, dtype=np.int32))
    np.testing.assert_array_equal(sup.export(L.SUP_FVAL), np.asarray(fval, dtype=np.float32))
    node_of_col = ref["unode"][want["lcol"]]
    np.testing.assert_array_equal(sup.export(L.SUP_LNODE_ORD), np.searchsorted(want["lnode"], node_of_col))
    assert sup.E == len(fcol)
    # products
    A = _dense(rows, cols, vals, N, R)
    gcols = ref["urel"][want["lcol"]].astype(np.int64) * N + node_of_col
    As = A[np.ix_(frow, gcols)]                       # [NR, L] stored values
    Aones = (np.zeros_like(As))
    for q, i in enumerate(frow):                      # the entries (incl. stored zeros) as ones
        np.add.at(Aones[q], lpos[ccol[rowptr[i]:rowptr[i + 1]]], 1.0)
    s = torch.cuda.current_stream().cuda_stream
    for F in (10, 16, 3):
        ld = (F + 3) // 4 * 4
        D = rng.standard_normal((max(sup.L, 1), ld)).astype(np.float32)
        Dt = torch.from_numpy(D).cuda()
        bias = rng.standard_normal(F).astype(np.float32)
        bt = torch.from_numpy(bias).cuda()
        for use_values, M in ((1, As), (0, Aones)):
            Y = torch.full((sup.NR, F), 7.0, device="cuda")
            L.check(lib.mrgcn_support_spmm_fwd_f32(sup.handle, use_values, Dt.data_ptr(), ld, F, Y.data_ptr(), F,
                                                   bt.data_ptr(), 1, s))
            want_Y = np.maximum(M @ D[:sup.L, :F].astype(np.float64) + bias, 0.0)
            np.testing.assert_allclose(Y.cpu().numpy(), want_Y, rtol=1e-4, atol=1e-4)
            dY = rng.standard_normal((sup.NR, F)).astype(np.float32)
            dM = torch.full((max(sup.L, 1), ld), 7.0, device="cuda")
            L.check(lib.mrgcn_support_spmm_t_compact_f32(sup.handle, use_values, torch.from_numpy(dY).cuda().data_ptr(), F,
                                                         F, dM.data_ptr(), ld, s))
            np.testing.assert_allclose(dM.cpu().numpy()[:sup.L, :F], M.T @ dY.astype(np.float64), rtol=1e-4, atol=1e-4)
    sup.close()


@pytest.mark.parametrize("B,F,K", [(3, 10, 12), (40, 10, 155), (2, 16, 5)])
def test_forward_support_mix_and_transform(B, F, K):
    """mix_fwd (live columns of weight_I x comp), rel_transform_fwd (X by live-node rank) and the compact backward of
    the transform against float64."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphSupport
    lib = L.load()
    rng = np.random.default_rng(B)
    N, R = 1200, 9
    rows, cols, vals = _random_graph(rng, N, N, R, 15000, hub_rows=2, hub_len=300, hub_cols=2)
    plan = _plan_from_coo(rows, cols, vals, N, N, R)
    ref = util.numpy_plan(rows, cols, vals, N, N, R)
    flags = np.zeros(N, dtype=np.uint8)
    flags[rng.choice(N, 50, replace=False)] = 1
    sup = GraphSupport(plan, torch.from_numpy(flags).cuda(), forward=True)
    want = _numpy_support(ref, flags, N, R)
    lrel, lnode = want["lrel"].astype(np.int64), want["lnode"]
    node_of_col = ref["unode"][want["lcol"]].astype(np.int64)
    ordn = np.searchsorted(lnode, node_of_col)
    s = torch.cuda.current_stream().cuda_stream
    ld = (F + 3) // 4 * 4
    V = rng.standard_normal((N, B, F)).astype(np.float32)
    comp = rng.standard_normal((R, B)).astype(np.float32)
    M = torch.zeros((sup.L, ld), device="cuda")
    L.check(lib.mrgcn_support_mix_fwd_f32(sup.handle, torch.from_numpy(V).cuda().data_ptr(),
                                          torch.from_numpy(comp).cuda().data_ptr(), B, F, M.data_ptr(), ld, s))
    want_M = np.einsum("kb,kbf->kf", comp[lrel].astype(np.float64), V[node_of_col].astype(np.float64))
    np.testing.assert_allclose(M.cpu().numpy()[:, :F], want_M, rtol=1e-4, atol=1e-4)
    need_dX = True   # (wide inputs: the dX pass runs in slices of 64 output columns)
    assert lib.mrgcn_support_rel_transform_supported(sup.handle, K, F, int(need_dX))
    assert not lib.mrgcn_support_rel_transform_supported(sup.handle, 300, F, 1)
    X = rng.standard_normal((sup.NL, K)).astype(np.float32)
    W = rng.standard_normal((R, K, F)).astype(np.float32)
    Xt, Wt = torch.from_numpy(X).cuda(), torch.from_numpy(W).cuda()
    T = torch.zeros((sup.L, ld), device="cuda")
    L.check(lib.mrgcn_support_rel_transform_fwd_f32(sup.handle, Xt.data_ptr(), K, 0, K, Wt.data_ptr(), F, T.data_ptr(), ld, s))
    want_T = np.einsum("kc,kcf->kf", X[ordn].astype(np.float64), W[lrel].astype(np.float64))
    np.testing.assert_allclose(T.cpu().numpy()[:, :F], want_T, rtol=1e-4, atol=1e-4)
    # the whole feature matrix, one row per node: the same rows picked inside the transform
    Xfull = np.zeros((N, K), dtype=np.float32)
    Xfull[lnode] = X
    Xf = torch.from_numpy(Xfull).cuda()
    T2 = torch.zeros((sup.L, ld), device="cuda")
    L.check(lib.mrgcn_support_rel_transform_fwd_f32(sup.handle, Xf.data_ptr(), K, 1, K, Wt.data_ptr(), F, T2.data_ptr(), ld, s))
    assert torch.equal(T2[:, :F], T[:, :F])
    dT = rng.standard_normal((sup.L, ld)).astype(np.float32)
    nws = int(lib.mrgcn_support_rel_transform_bwd_workspace(sup.handle, K, F, int(need_dX), 1))
    ws = torch.empty(max(nws, 2), device="cuda")
    dX = torch.full((sup.NL, K), 7.0, device="cuda")
    dW = torch.full((R, K, F), 7.0, device="cuda")
    L.check(lib.mrgcn_support_rel_transform_bwd_compact_f32(
        sup.handle, torch.from_numpy(dT).cuda().data_ptr(), ld, Xt.data_ptr(), K, 0, K, Wt.data_ptr(), F,
        dX.data_ptr() if need_dX else 0, K, dW.data_ptr(), ws.data_ptr(), ws.numel(), 0, s))
    want_dW = np.zeros((R, K, F))
    np.add.at(want_dW, lrel, np.einsum("kc,kf->kcf", X[ordn].astype(np.float64), dT[:, :F].astype(np.float64)))
    want_dX = np.zeros((sup.NL, K))
    np.add.at(want_dX, ordn, np.einsum("kf,kcf->kc", dT[:, :F].astype(np.float64), W[lrel].astype(np.float64)))
    np.testing.assert_allclose(dW.cpu().numpy(), want_dW, rtol=1e-3, atol=1e-3)
    if need_dX:
        np.testing.assert_allclose(dX.cpu().numpy(), want_dX, rtol=1e-3, atol=1e-3)
    dW2 = torch.full((R, K, F), 7.0, device="cuda")
    L.check(lib.mrgcn_support_rel_transform_bwd_compact_f32(
        sup.handle, torch.from_numpy(dT).cuda().data_ptr(), ld, Xf.data_ptr(), K, 1, K, Wt.data_ptr(), F, 0, K,
        dW2.data_ptr(), ws.data_ptr(), ws.numel(), 0, s))
    assert torch.equal(dW2, dW)
    sup.close()


def test_support_chain_equals_supports_built_one_by_one():
    """mrgcn_support_create_chain (one host wait for all levels): level i + 1 is the support of level i's NODE_FLAGS,
    array for array what building them one after the other gives."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphSupport
    rng = np.random.default_rng(11)
    N, R = 2500, 6
    rows, cols, vals = _random_graph(rng, N, N, R, 9000, hub_rows=2, hub_len=600, hub_cols=3)
    plan = _plan_from_coo(rows, cols, vals, N, N, R)
    flags = np.zeros(N, dtype=np.uint8)
    flags[rng.choice(N, 5, replace=False)] = 1
    ft = torch.from_numpy(flags).cuda()
    chain = GraphSupport.chain(plan, ft, 3, forward=True)
    f = ft
    for lvl in range(3):
        one = GraphSupport(plan, f, forward=True)
        c = chain[lvl]
        assert (c.NR, c.L, c.E, c.NL) == (one.NR, one.L, one.E, one.NL)
        for which in (L.SUP_COL_FLAGS, L.SUP_NODE_FLAGS, L.SUP_LCOL, L.SUP_LREL, L.SUP_NLPTR, L.SUP_LPTR, L.SUP_LROW,
                      L.SUP_LVAL, L.SUP_LNODE, L.SUP_LPERM, L.SUP_FROW, L.SUP_FPTR, L.SUP_FCOL, L.SUP_FVAL,
                      L.SUP_LNODE_ORD, L.SUP_ROWRANK):
            np.testing.assert_array_equal(c.export(which), one.export(which), err_msg=f"level {lvl} array {which}")
        f = one.node_flags()
    assert chain[0].NR == 5 and chain[1].NR == chain[0].NL and chain[2].NR == chain[1].NL


def test_support_of_an_empty_row_set_and_of_rows_without_entries():
    """No flagged row at all, and flagged rows that hold no entry: every count is zero, the products write nothing
    (or zeros + bias), nothing faults."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphSupport
    lib = L.load()
    rng = np.random.default_rng(5)
    N, R = 400, 3
    rows, cols, vals = _random_graph(rng, N, N, R, 900, hub_rows=0, hub_len=10, hub_cols=0)
    keep = rows >= 20              # rows 0..19 hold no entry
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    plan = _plan_from_coo(rows, cols, vals, N, N, R)
    s = torch.cuda.current_stream().cuda_stream
    for flagged in ([], [3, 7, 11]):
        flags = np.zeros(N, dtype=np.uint8)
        flags[flagged] = 1
        chain = GraphSupport.chain(plan, torch.from_numpy(flags).cuda(), 2, forward=True)
        assert chain[0].NR == len(flagged) and chain[0].L == chain[0].E == chain[0].NL == 0
        assert chain[1].NR == 0 and chain[1].L == 0
        F = 10
        D = torch.zeros((1, 12), device="cuda")
        Y = torch.full((max(len(flagged), 1), F), 7.0, device="cuda")
        bias = torch.arange(F, dtype=torch.float32, device="cuda")
        L.check(lib.mrgcn_support_spmm_fwd_f32(chain[0].handle, 0, D.data_ptr(), 12, F, Y.data_ptr(), F, bias.data_ptr(), 0, s))
        if flagged:
            np.testing.assert_array_equal(Y.cpu().numpy(), np.tile(np.arange(F, dtype=np.float32), (len(flagged), 1)))
        for c in chain:
            c.close()
