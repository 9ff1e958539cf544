"""AM-shaped (BASELINE config 3) parity through size-independent properties and against scipy
at full size: the oracle side (scipy CSR x dense, float64) finishes in seconds."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def am():
    from mrgcn_amd import synth
    from mrgcn_amd.plan import GraphPlan
    g = synth.make_graph("am", seed=0, scale=1.0)
    N, R = g.num_nodes, g.num_relations
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals),
                                (N, R * N)).cuda()
    plan = GraphPlan(A, N, R)
    A_csr = sp.csr_matrix((g.vals.astype(np.float64), (g.rows, g.cols)), shape=(N, R * N))
    return g, plan, A_csr


def test_am_shape_counts(am):
    g, plan, A_csr = am
    assert plan.nnz == g.nnz == 2 * len(g.triples) + g.num_nodes == 13643406
    assert plan.ncols == len(np.unique(g.cols)) and plan.max_row_nnz == np.bincount(g.rows).max()


def test_compact_product_vs_scipy_full_size(am):
    """Y = A.D on the compact operand (operand order `mpos`) against scipy; |err| <= 1e-4 (1 + |ref|):
    rows of up to 117k fp32 terms."""
    from mrgcn_amd import _lib as L
    g, plan, A_csr = am
    rng = np.random.default_rng(1)
    F, ld = 10, 12
    ulcol, mpos = plan.export(L.ARR_ULCOL), plan.export(L.ARR_MPOS)
    Mc = rng.standard_normal((plan.ncols, F)).astype(np.float32)       # compact order
    M = np.zeros((plan.ncols, ld), dtype=np.float32)
    M[mpos, :F] = Mc
    Y = plan.spmm(L.VIEW_COMPACT, torch.from_numpy(M).cuda(), F=F).cpu().numpy()
    # same product through scipy on the touched columns only
    A_touched = A_csr[:, ulcol.astype(np.int64)]
    Y_ref = A_touched @ Mc.astype(np.float64)
    np.testing.assert_allclose(Y, Y_ref, rtol=1e-4, atol=1e-4)
    # reproducible bit for bit (no atomics on this path)
    Y2 = plan.spmm(L.VIEW_COMPACT, torch.from_numpy(M).cuda(), F=F).cpu().numpy()
    assert np.array_equal(Y, Y2)


def test_transposed_is_the_adjoint_full_size(am):
    """<A' M, Y> == <M, A'^T Y> and linearity, at full size (no reference needed)."""
    from mrgcn_amd import _lib as L
    g, plan, A_csr = am
    F = 11
    mpos = torch.from_numpy(plan.export(L.ARR_MPOS).astype(np.int64)).cuda()
    Mc = torch.randn((plan.ncols, F), device="cuda", dtype=torch.float64)
    M = torch.zeros((plan.ncols, 12), device="cuda")
    M[mpos, :F] = Mc.float()
    Yw = torch.randn((g.num_nodes, F), device="cuda")
    AM = plan.spmm(L.VIEW_COMPACT, M, F=F)
    ATy = plan.spmm(L.VIEW_TRANSPOSED, Yw, F=F)          # compact (j, r) order, ld = F
    lhs = float((AM.double() * Yw.double()).sum())
    rhs = float((Mc.float().double() * ATy.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs), abs(rhs)) * 50
    AM2 = plan.spmm(L.VIEW_COMPACT, 2.5 * M, F=F)
    torch.testing.assert_close(AM2, 2.5 * AM, rtol=1e-5, atol=1e-5)


def test_fused_engine_equals_literal_engine_am_quarter():
    """The fused layer (no (R*N) x out intermediates) and the op-for-op literal layer give the same
    logits and gradients on an AM/4-shaped graph with the AM model (155 -> 10 -> 11, 40 bases)."""
    from mrgcn_amd import synth
    from mrgcn_amd.models.rgcn import RGCN
    g = synth.make_graph("am", seed=2, scale=0.25)
    N, R = g.num_nodes, g.num_relations
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals),
                                (N, R * N)).cuda()
    torch.manual_seed(0)
    model = RGCN([(155, 10, "mrgcn", torch.nn.ReLU()), (10, 11, "mrgcn", None)], R, N, 40, 0.0, False,
                 True, False).cuda()
    X = torch.randn((N, 155), device="cuda")
    w = torch.randn((N, 11), device="cuda")
    res = {}
    for engine in ("fused", "literal"):
        model.set_engine(engine)
        model.zero_grad(set_to_none=True)
        Y = model(X, A)
        (Y * w).sum().backward()
        res[engine] = (Y.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters()})
    torch.testing.assert_close(res["fused"][0], res["literal"][0], rtol=1e-4, atol=1e-4)
    for n in res["fused"][1]:
        a, b = res["fused"][1][n], res["literal"][1][n]
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-4 * scale + 1e-6, n


def test_int8_boundary_cast_pruned_equals_unpruned():
    """`ref_int8` graphs: dropping the entries the int8 cast zeroed (MRGCN_PLAN_PRUNE_ZEROS) does
    not change the product."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd import synth
    from mrgcn_amd.plan import GraphPlan
    g = synth.make_graph("mutag", seed=1, value_mode="ref_int8")
    N, R = g.num_nodes, g.num_relations
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals),
                                (N, R * N)).cuda()
    assert A.dtype == torch.int8
    full, pruned = GraphPlan(A, N, R), GraphPlan(A, N, R, prune_zeros=True)
    assert pruned.nnz == int((g.vals != 0).sum()) < full.nnz
    D = torch.randn((R * N, 16), device="cuda")
    torch.testing.assert_close(full.spmm(L.VIEW_LITERAL, D), pruned.spmm(L.VIEW_LITERAL, D),
                               rtol=1e-6, atol=1e-6)


# ---- the benchmarked epoch at the benchmarked size -----------------------------------------------
def _am_model_and_data(am, seed=0):
    from mrgcn_amd import synth
    from mrgcn_amd.models.rgcn import RGCN
    g, plan, A_csr = am
    N, R = g.num_nodes, g.num_relations
    A = plan.as_adjacency_handle()
    dims = synth.layer_dims("am")
    B = synth.SHAPES["am"]["bases"]
    torch.manual_seed(seed)
    model = RGCN([(dims[0][0], dims[0][1], "mrgcn", torch.nn.ReLU()), (dims[1][0], dims[1][1], "mrgcn", None)],
                 R, N, B, 0.0, False, True, False).cuda()
    X = torch.randn((N, dims[0][0]), device="cuda", generator=torch.Generator("cuda").manual_seed(seed + 1))
    idx, y = synth.make_labels("am", N, seed=0)
    return model, A, X, torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda(), dims, B


def _oracle_rows(am, model, X, rows, dims, B):
    from oracle import rgcn_oracle as O
    g, plan, A_csr = am
    N, R = g.num_nodes, g.num_relations
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    cfgs = O.rgcn_cfgs(dims, R, N, B, True, False)
    return O.rgcn_forward_at_rows(cfgs, O.split_params(state, len(cfgs)), X.cpu().numpy(), A_csr, rows)


def test_am_epoch_logits_against_the_float64_oracle_at_sampled_rows(am):
    """The path bench.py times (fused engine, live-column backward, row-sparse weight_I gradient and
    Adam, hipGraph replay) at N = 1.67 M: logits of 200 sampled rows — labelled nodes, the largest hubs
    and random ones — against the float64 oracle evaluated on their 2-hop receptive field
    (oracle.rgcn_forward_at_rows, graph.py:62-102 / rgcn.py:69-89), before training and after four
    epochs (three of them replayed).  Tolerance 1e-4, the north_star's."""
    from mrgcn_amd.train import ClipAdam, GraphedTrainStep
    g, plan, A_csr = am
    model, A, X, idx, y, dims, B = _am_model_and_data(am)
    deg = np.bincount(g.rows, minlength=g.num_nodes)
    rng = np.random.default_rng(4)
    rows = np.unique(np.concatenate([idx.cpu().numpy()[:80], np.argsort(deg)[-20:], rng.choice(g.num_nodes, 100)]))
    with torch.no_grad():
        got = model(X, A)[torch.from_numpy(rows).cuda()].cpu().numpy()
    ref = _oracle_rows(am, model, X, rows, dims, B)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0, capturable=True)
    step = GraphedTrainStep(model, lambda: model(X, A), idx, y, opt, warmup=1)
    losses = [float(step()) for _ in range(3)]
    assert losses[-1] < losses[0]
    with torch.no_grad():
        got = model(X, A)[torch.from_numpy(rows).cuda()].cpu().numpy()
    ref = _oracle_rows(am, model, X, rows, dims, B)  # the oracle on the TRAINED parameters
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)


def _compare_runs(names, sd_a, sd_b, mom_a, osd_b, steps):
    for k, st in osd_b["state"].items():
        n = names[k]
        ma, va = mom_a[n]
        assert ma.shape == st["exp_avg"].shape, n
        ms, vs = float(st["exp_avg"].abs().max()), float(st["exp_avg_sq"].abs().max())
        assert float((ma - st["exp_avg"]).abs().max()) <= 2e-3 * ms + 1e-9, n
        assert float((va - st["exp_avg_sq"]).abs().max()) <= 4e-3 * vs + 1e-12, n
    for n, p in sd_b.items():
        d = (sd_a[n] - p).abs()
        # Adam's first steps move every element by ~lr * sign(g): elements whose gradient is at fp32
        # rounding level may differ by a step; everything else agrees tightly
        assert float((d > 2e-5).float().mean()) < 2e-3, n
        assert float(d.max()) <= 0.021 * steps, n


def _default_run(model, A, X, idx, y):
    from mrgcn_amd.train import ClipAdam, GraphedTrainStep
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0, capturable=True)
    step = GraphedTrainStep(model, lambda: model(X, A), idx, y, opt, warmup=1)  # one eager epoch ...
    la = [float(step()), float(step())]                                          # ... and two replayed
    with torch.no_grad():
        logits_a = model(X, A).clone()
    sd_a = {k: v.detach().clone() for k, v in model.state_dict().items()}
    osd = opt.state_dict()
    names = [n for n, _ in model.named_parameters()]
    mom_a = {names[k]: (st["exp_avg"].clone(), st["exp_avg_sq"].clone()) for k, st in osd["state"].items()}
    assert all(int(st["step"]) == 3 for st in osd["state"].values())
    return la, logits_a, sd_a, mom_a, names


def test_am_default_epoch_equals_the_epoch_without_shortcuts(am):
    """Three epochs at N = 1.67 M two ways from the same start: (a) the default path — live-column backward,
    row-sparse weight_I gradient and Adam, captured and replayed from a hipGraph; (b) the same kernels with every
    gradient-sparsity shortcut off (general transposed product, dense gradient in .grad, plain dense Adam kernel),
    launched eagerly.  Losses, logits, parameters and Adam moments agree.  (The op-for-op literal engine is
    compared at a size its 4.45 G-element library GEMM can hold: next test.)"""
    from mrgcn_amd import functional as Fn
    from mrgcn_amd.train import ClipAdam, train_step
    g, plan, A_csr = am
    model, A, X, idx, y, dims, B = _am_model_and_data(am, seed=5)
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    la, logits_a, sd_a, mom_a, names = _default_run(model, A, X, idx, y)
    model.load_state_dict(init)
    opt_b = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0)
    prev, Fn._LIVE_COLS = Fn._LIVE_COLS, False
    try:
        lb = [float(train_step(model, lambda: model(X, A), idx, y, opt_b, row_sparse=False)) for _ in range(3)]
    finally:
        Fn._LIVE_COLS = prev
    np.testing.assert_allclose(la, lb[1:], rtol=2e-5, atol=2e-6)
    with torch.no_grad():
        logits_b = model(X, A)
    assert float((logits_a - logits_b).abs().max()) <= 1e-4 * max(1.0, float(logits_b.abs().max()))
    _compare_runs(names, sd_a, model.state_dict(), mom_a, opt_b.state_dict(), 3)


def test_am_default_epoch_equals_literal_engine_with_dense_adam(am):
    """The same comparison against the op-for-op literal engine at N = 1.67 M: materialised (R*N) x out
    operands (17.8 GB each) and dense gradients (graph.py:70-75,:93-95), plain dense Adam, eager launches.
    (The literal engine issues its library GEMMs in pieces: with results beyond 4 GiB the ROCm BLAS path
    returned a few wrong rows — layers/graph.py::_MAX_GEMM_OUT.)"""
    from mrgcn_amd.train import ClipAdam, train_step
    g, plan, A_csr = am
    model, A, X, idx, y, dims, B = _am_model_and_data(am, seed=9)
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    la, logits_a, sd_a, mom_a, names = _default_run(model, A, X, idx, y)
    model.load_state_dict(init)
    model.set_engine("literal")
    opt_b = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0)
    lb = [float(train_step(model, lambda: model(X, A), idx, y, opt_b, row_sparse=False)) for _ in range(3)]
    np.testing.assert_allclose(la, lb[1:], rtol=2e-4, atol=2e-5)
    with torch.no_grad():
        logits_b = model(X, A)
    assert float((logits_a - logits_b).abs().max()) <= 1e-4 * max(1.0, float(logits_b.abs().max()))
    _compare_runs(names, sd_a, model.state_dict(), mom_a, opt_b.state_dict(), 3)
    del opt_b
    torch.cuda.empty_cache()


def test_am_quarter_bf16_operand_logits_within_the_stated_tolerance():
    """Config 3 names bf16: the compact operand stored in bf16 (fp32 accumulation) on an AM/4 graph
    with the AM model — logits within 2e-2 of the fp32 run (relative to the largest logit)."""
    from mrgcn_amd import synth
    from mrgcn_amd.models.rgcn import RGCN
    g = synth.make_graph("am", seed=2, scale=0.25)
    N, R = g.num_nodes, g.num_relations
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals),
                                (N, R * N)).cuda()
    torch.manual_seed(0)
    model = RGCN([(155, 10, "mrgcn", torch.nn.ReLU()), (10, 11, "mrgcn", None)], R, N, 40, 0.0, False,
                 True, False).cuda()
    X = torch.randn((N, 155), device="cuda")
    with torch.no_grad():
        ref = model(X, A)
        model.set_operand_dtype("bf16")
        got = model(X, A)
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert 0 < err <= 2e-2 * scale, (err, scale)


def _sample_rows(g, idx, rng_seed=4, n_lab=80, n_hub=20, n_rand=100):
    deg = np.bincount(g.rows, minlength=g.num_nodes)
    rng = np.random.default_rng(rng_seed)
    return np.unique(np.concatenate([idx.cpu().numpy()[:n_lab], np.argsort(deg)[-n_hub:],
                                     rng.choice(g.num_nodes, n_rand)]))


def test_am_bf16_operand_against_the_float64_oracle_at_full_size(am):
    """BASELINE config 3 names bf16.  The compact operand stored in bf16 (fp32 values, accumulation and outputs) at
    N = 1.67 M against the float64 oracle on the receptive field of 200 sampled rows (labelled nodes, the largest
    hubs, random ones) — before training and after three epochs on the bf16 path, the oracle then evaluated on the
    TRAINED parameters.  Tolerance: 2e-2 of the largest logit (SURVEY §8d's stated bf16 tolerance; the reference
    has no bf16 anywhere)."""
    from mrgcn_amd.train import ClipAdam, train_step
    g, plan, A_csr = am
    model, A, X, idx, y, dims, B = _am_model_and_data(am, seed=2)
    model.set_operand_dtype("bf16")
    rows = _sample_rows(g, idx)
    sel = torch.from_numpy(rows).cuda()
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0)
    for phase in range(2):
        with torch.no_grad():
            got = model(X, A)[sel].cpu().numpy()
        ref = _oracle_rows(am, model, X, rows, dims, B)
        scale = float(np.abs(ref).max())
        err = float(np.abs(got - ref).max())
        assert 0 < err <= 2e-2 * scale, (phase, err, scale)
        if phase == 0:
            losses = [float(train_step(model, lambda: model(X, A), idx, y, opt)) for _ in range(3)]
            assert losses[-1] < losses[0]


@pytest.mark.parametrize("prune", [False, True])
def test_am_epoch_with_the_int8_boundary_cast_against_the_oracle(prune):
    """The reference's real value mode at the benchmarked size: `FullBatch.as_tensors_` truncates the normalised
    adjacency to int8 (batch.py:144-149: only entries that are exactly 1.0 survive, the rest are stored zeros).
    AM shape in `ref_int8`, with the stored zeros kept and with MRGCN_PLAN_PRUNE_ZEROS: logits of sampled rows against
    the float64 oracle on the same int8 values, before training and after three epochs (one eager, two replayed
    from a hipGraph); the pruned and the unpruned plan give the same logits."""
    from mrgcn_amd import synth
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.plan import GraphPlan
    from mrgcn_amd.train import ClipAdam, GraphedTrainStep
    from oracle import rgcn_oracle as O
    g = synth.make_graph("am", seed=0, scale=1.0, value_mode="ref_int8")
    N, R = g.num_nodes, g.num_relations
    assert g.vals.dtype == np.int8 and 0 < int((g.vals != 0).sum()) < g.nnz
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals),
                                (N, R * N)).cuda()
    assert A.dtype == torch.int8
    dims, B = synth.layer_dims("am"), synth.SHAPES["am"]["bases"]
    torch.manual_seed(1)
    model = RGCN([(dims[0][0], dims[0][1], "mrgcn", torch.nn.ReLU()), (dims[1][0], dims[1][1], "mrgcn", None)],
                 R, N, B, 0.0, False, True, False).cuda()
    plan = GraphPlan(A, N, R, prune_zeros=prune, operand_row_bytes=model.operand_row_bytes())
    assert plan.nnz == (int((g.vals != 0).sum()) if prune else g.nnz)
    Ah = plan.as_adjacency_handle()
    X = torch.randn((N, dims[0][0]), device="cuda", generator=torch.Generator("cuda").manual_seed(2))
    idx_np, y_np = synth.make_labels("am", N, seed=0)
    idx, y = torch.from_numpy(idx_np).cuda(), torch.from_numpy(y_np).cuda()
    A_csr = sp.csr_matrix((g.vals.astype(np.float64), (g.rows, g.cols)), shape=(N, R * N))
    rows = _sample_rows(g, idx)
    sel = torch.from_numpy(rows).cuda()
    cfgs = O.rgcn_cfgs(dims, R, N, B, True, False)

    def check():
        with torch.no_grad():
            got = model(X, Ah)[sel].cpu().numpy()
        state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        ref = O.rgcn_forward_at_rows(cfgs, O.split_params(state, len(cfgs)), X.cpu().numpy(), A_csr, rows)
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
        return got

    before = check()
    if prune:  # the same logits as on the plan that keeps the stored zeros
        with torch.no_grad():
            full = model(X, A)[sel].cpu().numpy()
        np.testing.assert_allclose(before, full, rtol=1e-6, atol=1e-6)
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0, capturable=True)
    step = GraphedTrainStep(model, lambda: model(X, Ah), idx, y, opt, warmup=1)
    losses = [float(step()) for _ in range(2)]
    assert np.isfinite(losses).all()
    check()
