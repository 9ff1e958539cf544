"""GPU parity of the fused kernels, the layer / model mirrors and the epoch step against the
golden vectors captured from the reference (tests/golden) and against the numpy oracle."""
import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

# logits tolerance of BASELINE.json's north_star: 1e-4 (fp32)
TOL = dict(rtol=1e-4, atol=1e-4)


def _adjacency(c, name):
    g, A = util.load_graph(util.graph_of_case(name))
    return util.coo_tensor(A, str(c["value_mode"]), "cuda")


@pytest.mark.parametrize("engine", ["fused", "literal"])
@pytest.mark.parametrize("name", util.rgcn_cases())
def test_rgcn_forward_backward_vs_reference_goldens(name, engine):
    c = util.load_case(name)
    model, dims = util.build_rgcn_from_case(c, "cuda")
    util.load_state_from_case(model, c)
    model = model.cuda()
    model.set_engine(engine)
    A = _adjacency(c, name)
    fl = bool(c["meta.featureless"])
    X = None if fl else torch.from_numpy(c["X"]).cuda().requires_grad_(True)

    # per-layer activations
    with torch.no_grad():
        H = X
        for li, (key, layer) in enumerate(model.layers.items()):
            H = layer(H, A)
            np.testing.assert_allclose(H.cpu().numpy(), c[f"act.pre_{li}"], err_msg=f"pre_{li}", **TOL)
            act = model.activations[key]
            if act is not None:
                H = act(H)

    logits = model(X, A)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), c["logits"], **TOL)
    # what the caller sees is dense like the reference's output; only hidden layers of the fused engine keep their
    # output in rows padded to whole 16-byte pieces (plan.spmm)
    assert logits.is_contiguous()
    assert [bool(getattr(l, "padded_output", False)) for l in model.layers.values()] == \
        [True] * (len(model.layers) - 1) + [False]
    idx = torch.from_numpy(c["labels_idx"]).cuda()
    tgt = torch.from_numpy(c["labels_y"]).cuda()
    from mrgcn_amd.train import categorical_crossentropy
    loss = categorical_crossentropy(logits, idx, tgt)
    np.testing.assert_allclose(float(loss), float(c["loss"]), rtol=1e-5, atol=1e-5)
    loss.backward()
    for n, p in model.named_parameters():
        if n == "relations":
            assert p.grad is None
            continue
        np.testing.assert_allclose(util.ref_layout(p.grad, n).cpu().numpy(), c["grad." + n], rtol=1e-3, atol=1e-5,
                                   err_msg=n)
    if not fl:
        np.testing.assert_allclose(X.grad.cpu().numpy(), c["grad.X"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("row_sparse", [None, False])
@pytest.mark.parametrize("name", util.rgcn_cases())
def test_epoch_steps_vs_reference_goldens(name, row_sparse):
    """zero_grad / backward / clip 1.0 / Adam driven for n_adam epochs
    (node_classification.py:166-193), with the row-sparse weight_I gradient (default) and the dense one."""
    from mrgcn_amd.train import ClipAdam, train_step
    _epoch_steps(name, ClipAdam, train_step, row_sparse)


def _epoch_steps(name, ClipAdam, train_step, row_sparse=None):
    c = util.load_case(name)
    model, dims = util.build_rgcn_from_case(c, "cuda")
    util.load_state_from_case(model, c)
    model = model.cuda()
    A = _adjacency(c, name)
    fl = bool(c["meta.featureless"])
    X = None if fl else torch.from_numpy(c["X"]).cuda()
    idx = torch.from_numpy(c["labels_idx"]).cuda()
    tgt = torch.from_numpy(c["labels_y"]).cuda()
    opt = ClipAdam([p for n, p in model.named_parameters()], lr=0.01, weight_decay=0.0, max_norm=1.0)
    n_adam = int(c["meta.n_adam"])
    for step in range(1, n_adam + 1):
        loss = train_step(model, lambda: model(X, A), idx, tgt, opt, row_sparse=row_sparse)
        np.testing.assert_allclose(float(loss), float(c[f"loss_step{step}"]), rtol=2e-4, atol=2e-5)
        if step == 1:
            np.testing.assert_allclose(opt.last_grad_norm(), float(c["grad_norm"]), rtol=1e-4)
        if step in (1, n_adam):
            sd = model.state_dict()
            coef = min(1.0, 1.0 / (float(c["grad_norm"]) + 1e-6))
            for k in c.files:
                if k.startswith(f"adam{step}."):
                    key = k[len(f"adam{step}."):]
                    diff = np.abs(sd[key].cpu().numpy() - c[k])
                    assert diff.max() <= 0.021 * step, k
                    if step == 1 and "grad." + key in c.files:
                        # Adam's first step moves an element by lr * g / (|g| + eps): a whole lr-sized step may differ
                        # only where the reference's own (clipped) gradient is rounding noise; every other element —
                        # every node block of weight_I among them — lands on the reference's value
                        g = np.abs(c["grad." + key]) * coef
                        bad = (diff > 2e-5) & (g > 1e-6)
                        assert not bad.any(), (k, int(bad.sum()), np.argwhere(bad)[:5].tolist())
                    else:   # (later steps: the goldens hold no gradient to tell noise from signal)
                        assert (diff > 2e-5).mean() < 2e-3, (k, float((diff > 2e-5).mean()))


@pytest.mark.parametrize("name", ["mrgcn_small_featureless_b0", "mrgcn_small_featureless_b3",
                                  "mrgcn_small_encoders_b3"])
def test_mrgcn_through_fullbatch_boundary(name):
    """MRGCN(FullBatch) with the reference's int8 boundary cast: logits, loss, accuracy, grads
    and three optimizer steps."""
    import scipy.sparse as sp
    from mrgcn_amd.data.batch import FullBatch
    from mrgcn_amd.models.mrgcn import MRGCN
    from mrgcn_amd.train import ClipAdam, categorical_accuracy, categorical_crossentropy
    c = util.load_case(name)
    g, A = util.load_graph("graph_small")
    N, R = int(c["meta.num_nodes"]), int(c["meta.R"])
    enc = bool(c["meta.with_encoders"])
    X = [np.empty((N, 0), dtype=float)]
    modules_config = []
    if enc:
        ia, ib = c["enc.numeric_idx"], c["enc.boolean_idx"]
        X.append(["xsd.numeric", [[c["enc.numeric"], ia, np.ones(len(ia), dtype=int)]], False])
        X.append(["xsd.boolean", [[c["enc.boolean"], ib, np.ones(len(ib), dtype=int)]], False])
        modules_config = sorted([("xsd.numeric", (4, 4, 0.0), False), ("xsd.boolean", (1, 2, 0.0), False)],
                                key=lambda t: t[0])
    xw = 6 if enc else 0
    modules = [(xw, int(c["meta.hidden"]), "mrgcn", torch.nn.ReLU()),
               (int(c["meta.hidden"]), int(c["meta.num_classes"]), "mrgcn", None)]
    model = MRGCN(modules, modules_config, R, N, num_bases=int(c["meta.num_bases"]), p_dropout=0.0,
                  featureless=not enc, bias=False, gcn_gpu_acceleration=True)
    model.load_state_dict({k[5:]: torch.from_numpy(np.array(c[k])) for k in c.files if k.startswith("init.")})
    assert model.devices["relational"].type == "cuda"
    batch = FullBatch(A, X, np.arange(N))
    batch.pad_(); batch.to_dense_(); batch.as_tensors_(); batch.to(model.devices)
    assert batch.A.is_cuda and batch.A.dtype == torch.int8

    idx = torch.from_numpy(c["labels_idx"]).cuda()
    tgt = torch.from_numpy(c["labels_y"]).cuda()
    opt = ClipAdam(list(model.parameters()), lr=0.01, max_norm=1.0)
    for step in (1, 2, 3):
        Y_hat = model(batch)
        loss = categorical_crossentropy(Y_hat, idx, tgt)
        opt.zero_grad()
        loss.backward()
        if step == 1:
            np.testing.assert_allclose(Y_hat.detach().cpu().numpy(), c["logits"], **TOL)
            np.testing.assert_allclose(float(loss), float(c["loss"]), rtol=1e-5, atol=1e-5)
            acc = categorical_accuracy(Y_hat.detach(), idx, tgt)[0]
            np.testing.assert_allclose(float(acc), float(c["accuracy"]), atol=1e-6)
            for n, p in model.named_parameters():
                np.testing.assert_allclose(util.ref_layout(p.grad, n).cpu().numpy(), c["grad." + n], rtol=1e-3,
                                           atol=1e-5, err_msg=n)
        opt.step()
        sd = model.state_dict()
        for k in c.files:
            if k.startswith(f"adam{step}."):
                diff = np.abs(sd[k[len(f"adam{step}."):]].cpu().numpy() - c[k])
                assert (diff > 2e-5).mean() < 5e-3 and diff.max() <= 0.021 * step, k


@pytest.mark.parametrize("N,R,B,F,bias,hub", [(400, 5, 2, 200, False, 150), (300, 7, 3, 130, True, 150),
                                              (200, 4, 0, 96, True, 150), (900, 6, 4, 64, False, 700),
                                              (700, 5, 1, 256, True, 600), (500, 3, 2, 20, False, 400)])
def test_featureless_wide_layer_vs_oracle(N, R, B, F, bias, hub):
    """FB15k-237-style encoder layer: featureless input layer with a wide hidden size
    (configs/fb15k-237.toml: 2 bases, hidden 200) — forward and every gradient.  B <= 4 with F % 4 == 0 takes the
    backward that never forms dM (csrc/wide_input.hip); hubs of several hundred entries per source node span several
    of its units (float atomics on their dV blocks)."""
    from oracle import rgcn_oracle as O
    from mrgcn_amd.layers.graph import GraphConvolution
    from mrgcn_amd.plan import plan_of
    rng = np.random.default_rng(F)
    rows, cols, vals, A = _oracle_layer_case(rng, N, R, max(B, 1), 1, F, 8 * N, hub)
    At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals),
                                 (N, R * N)).cuda()
    torch.manual_seed(F)
    layer = GraphConvolution(0, F, R, N, num_bases=B, bias=bias, input_layer=True, featureless=True).cuda()
    Y = layer._forward_fused(None, plan_of(At, N, R), relu=True)
    w = torch.randn_like(Y)
    (Y * w).sum().backward()
    cfg = O.LayerCfg(0, F, R, N, B, bias=bias, input_layer=True, featureless=True)
    p = {k: v.detach().cpu().numpy() for k, v in layer.state_dict().items()}  # the reference's shapes
    pre, cache = O.layer_forward(cfg, p, None, A)
    np.testing.assert_allclose(Y.detach().cpu().numpy(), np.maximum(pre, 0), rtol=1e-4, atol=1e-4)
    grads, _ = O.layer_backward(cfg, p, None, A, w.cpu().numpy().astype(np.float64) * (pre > 0), cache)
    for k, v in grads.items():
        got = util.ref_layout(getattr(layer, k).grad, k).cpu().numpy()
        np.testing.assert_allclose(got, v, rtol=2e-4, atol=2e-5 * (np.abs(v).max() + 1e-12) + 1e-6, err_msg=k)
    # the default backward (units of 64 entries, hub partials summed in unit order, no float atomics) gives the same
    # BITS every run; the round-5 form (256-entry units, atomics on hub blocks: MRGCN_WIDE_DET=0) the same values
    import mrgcn_amd
    from mrgcn_amd import functional as Fn
    first = {k: getattr(layer, k).grad.clone() for k in grads}

    def again():
        layer.zero_grad(set_to_none=True)
        Y2 = layer._forward_fused(None, plan_of(At, N, R), relu=True)
        (Y2 * w).sum().backward()
        return {k: getattr(layer, k).grad.clone() for k in grads}
    mrgcn_amd.reset_stats()
    second = again()
    wide = mrgcn_amd.stats().get("backward.wide_input") == 1
    assert wide == (1 <= B <= 4 and F % 4 == 0 and F > 16)
    for k in grads:   # (the other shapes' general path sums dcomp with float atomics: equal to rounding only)
        if wide:
            assert torch.equal(first[k], second[k]), k
        else:
            torch.testing.assert_close(second[k], first[k], rtol=1e-4, atol=1e-5 * float(first[k].abs().max()) + 1e-7)
    prev, Fn._WIDE_DET = Fn._WIDE_DET, False
    try:
        third = again()
    finally:
        Fn._WIDE_DET = prev
    for k in grads:
        torch.testing.assert_close(third[k], first[k], rtol=1e-4, atol=1e-5 * float(first[k].abs().max()) + 1e-7)


def _oracle_layer_case(rng, N, R, B, K, F, nnz, hub):
    """Random layer problem + float64 oracle results for the fused kernels."""
    import scipy.sparse as sp
    from oracle import rgcn_oracle as O
    RN = R * N
    rows = rng.integers(0, N, nnz); cols = rng.integers(0, RN, nnz)
    if hub:
        rows = np.concatenate([rows, np.full(hub, 3)]); cols = np.concatenate([cols, rng.choice(RN, hub, replace=False)])
        rows = np.concatenate([rows, rng.choice(N, min(hub, N), replace=False)]); cols = np.concatenate([cols, np.full(min(hub, N), 7)])
    ident = np.arange(N)
    rows = np.concatenate([rows, ident]); cols = np.concatenate([cols, (R - 1) * N + ident])
    key = np.unique(rows.astype(np.int64) * RN + cols)
    rows, cols = key // RN, key % RN
    vals = rng.uniform(0.1, 1.0, len(rows)).astype(np.float32)
    A = sp.csr_matrix((vals.astype(np.float64), (rows, cols)), shape=(N, RN))
    return rows, cols, vals, A


@pytest.mark.parametrize("N,R,B,K,F,hub", [(300, 5, 3, 7, 6, 0), (1500, 9, 40, 155, 10, 600),
                                           (700, 4, 2, 10, 11, 300), (500, 6, 70, 3, 16, 0),
                                           (400, 3, 5, 200, 33, 0)])
def test_fused_layer_vs_oracle(N, R, B, K, F, hub):
    """Input layer with bases + features, bias and ReLU: forward, every gradient."""
    import scipy.sparse as sp
    from oracle import rgcn_oracle as O
    from mrgcn_amd.layers.graph import GraphConvolution
    from mrgcn_amd.plan import plan_of
    rng = np.random.default_rng(N + R + B)
    rows, cols, vals, A = _oracle_layer_case(rng, N, R, B, K, F, 6 * N, hub)
    At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals),
                                 (N, R * N)).cuda()
    torch.manual_seed(N)
    layer = GraphConvolution(K, F, R, N, num_bases=B, bias=True, input_layer=True, featureless=False).cuda()
    with torch.no_grad():
        layer.b.copy_(torch.randn(F) * 0.1)
    X = torch.randn(N, K, device="cuda", requires_grad=True)
    plan = plan_of(At, N, R)
    Y = layer._forward_fused(X, plan, relu=True)
    w = torch.randn_like(Y)
    (Y * w).sum().backward()

    cfg = O.LayerCfg(K, F, R, N, B, bias=True, input_layer=True, featureless=False)
    p = {k: v.detach().cpu().numpy() for k, v in layer.state_dict().items()}  # the reference's shapes
    pre, cache = O.layer_forward(cfg, p, X.detach().cpu().numpy(), A)
    np.testing.assert_allclose(Y.detach().cpu().numpy(), np.maximum(pre, 0), rtol=1e-4, atol=1e-4)
    dPre = w.cpu().numpy().astype(np.float64) * (pre > 0)
    grads, dX = O.layer_backward(cfg, p, X.detach().cpu().numpy(), A, dPre, cache)
    scale = {k: np.abs(v).max() + 1e-12 for k, v in grads.items()}
    for k, v in grads.items():
        got = util.ref_layout(getattr(layer, k).grad, k).cpu().numpy()
        np.testing.assert_allclose(got, v, rtol=2e-4, atol=2e-5 * scale[k] + 1e-6, err_msg=k)
    np.testing.assert_allclose(X.grad.cpu().numpy(), dX, rtol=2e-4, atol=1e-5 * np.abs(dX).max())


@pytest.mark.parametrize("name", ["rgcn_smoke_ft_b5_norm_f32", "rgcn_small_fl_b0_bias_ref_int8"])
def test_graph_captured_epoch_equals_eager(name):
    """hipGraph-captured train step (GraphedTrainStep, ClipAdam(capturable=True)): three warm-up
    steps + three replays give the parameters of six eager steps (Adam's step counter and bias
    corrections live on the device, so the replayed graph advances them)."""
    from mrgcn_amd.train import ClipAdam, GraphedTrainStep, train_step
    c = util.load_case(name)
    A = _adjacency(c, name)
    fl = bool(c["meta.featureless"])
    X = None if fl else torch.from_numpy(c["X"]).cuda()
    idx = torch.from_numpy(c["labels_idx"]).cuda()
    tgt = torch.from_numpy(c["labels_y"]).cuda()
    states, losses = [], []
    for graphed in (False, True):
        model, _ = util.build_rgcn_from_case(c, "cuda")
        util.load_state_from_case(model, c)
        model = model.cuda()
        opt = ClipAdam(list(model.parameters()), lr=0.01, max_norm=1.0, capturable=graphed)
        if graphed:
            step = GraphedTrainStep(model, lambda: model(X, A), idx, tgt, opt, warmup=3)
            for _ in range(3):
                loss = step()
        else:
            for _ in range(6):
                loss = train_step(model, lambda: model(X, A), idx, tgt, opt)
        torch.cuda.synchronize()
        states.append({k: v.clone() for k, v in model.state_dict().items()})
        losses.append(float(loss))
    assert abs(losses[0] - losses[1]) <= 1e-5 * max(1.0, abs(losses[0]))
    for k in states[0]:
        torch.testing.assert_close(states[0][k], states[1][k], rtol=1e-5, atol=1e-6, msg=k)


@pytest.mark.parametrize("with_addend", [False, True])
@pytest.mark.parametrize("B,F", [(4, 4), (16, 16), (20, 8), (32, 16), (40, 10), (48, 4), (48, 16), (33, 12),
                                 (64, 4), (64, 8), (64, 12), (64, 16), (7, 6), (70, 8)])
def test_basis_mix_forward_shapes_through_the_c_abi(B, F, with_addend):
    """M[MPOS[c]] = addend[c] + comp[r_c] . V[j_c] for every instantiation of the matrix-core kernel (K steps
    ceil(B / 16) x 16-byte pieces ceil(B F / 256)), the scalar kernels beyond its limits ((7, 6): B F not a
    multiple of 4; (70, 8): B > 64), fp32 and bf16 operand rows, with a node of more than 64 columns (its step
    leaves the prefetched range), nodes of 17..64 columns (more than one tile) and nodes without any column."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphPlan
    N, R = 700, 90
    rng = np.random.default_rng(B * 100 + F)
    rows, cols, vals, _ = _oracle_layer_case(rng, N, R, B, 1, F, 6 * N, 400)
    # node 2 gets 85 columns, nodes 3..10 get 20..55 each; nodes 11..40 none at all
    sizes = {2: 85, **{j: 20 + 5 * (j - 3) for j in range(3, 11)}}
    extra_r = np.concatenate([rng.choice(R, size=n, replace=False) for n in sizes.values()])
    extra_j = np.concatenate([np.full(n, j) for j, n in sizes.items()])
    keep = ~np.isin(cols % N, np.arange(11, 41))
    rows, cols, vals = rows[keep], cols[keep], vals[keep]
    rows = np.concatenate([rows, rng.integers(0, N, size=extra_r.size)])
    cols = np.concatenate([cols, extra_r * N + extra_j])
    vals = np.concatenate([vals, np.ones(extra_r.size, dtype=vals.dtype)])
    At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals),
                                 (N, R * N)).coalesce().cuda()
    plan = GraphPlan(At, N, R)
    urel, unode = plan.export(L.ARR_UREL).astype(np.int64), plan.export(L.ARR_UNODE).astype(np.int64)
    mpos = plan.export(L.ARR_MPOS).astype(np.int64)
    per_node = np.bincount(unode, minlength=N)
    assert per_node.max() > 64 and ((per_node > 16) & (per_node <= 64)).sum() >= 8 and (per_node == 0).sum() >= 30
    nc = plan.ncols
    V = rng.standard_normal((N, B, F)).astype(np.float32)
    comp = rng.standard_normal((R, B)).astype(np.float32)
    ldA = (F + 3) // 4 * 4
    add = rng.standard_normal((nc, ldA)).astype(np.float32)
    want = np.einsum("cb,cbf->cf", comp.astype(np.float64)[urel], V.astype(np.float64)[unode])
    if with_addend:
        want = want + add[:, :F]
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    Vt, ct, at = (torch.from_numpy(x).cuda() for x in (V, comp, add))
    for sfx, dt, ld, tol in (("f32", torch.float32, ldA, 2e-5), ("bf16", torch.bfloat16, (F + 3) // 4 * 4, 1e-2)):
        M = torch.full((plan.nop, ld), 7.0, dtype=dt, device="cuda")
        rc = getattr(lib, "mrgcn_basis_mix_fwd_" + sfx)(
            plan.handle, Vt.data_ptr(), ct.data_ptr(), B, F, at.data_ptr() if with_addend else 0, ldA, M.data_ptr(),
            ld, s)
        if sfx == "bf16" and B > 64:   # (a second pass over the bases would round the operand twice)
            assert rc != 0
            continue
        L.check(rc)
        got = M.float().cpu().numpy()
        scale = np.abs(want).max()
        np.testing.assert_allclose(got[mpos][:, :F], want, rtol=tol, atol=tol * scale, err_msg=sfx)
        assert (got[:, F:] == 7.0).all() or (got[:, F:] == 0.0).all()   # padding: untouched (or zeroed)
        if with_addend:
            # the addend through LDS as 16-byte pieces (the default where its rows allow it) and as 4-byte loads per
            # node are the same arithmetic: equal bits
            old = L.set_config(mix_add_vec=0)
            try:
                M2 = torch.full((plan.nop, ld), 7.0, dtype=dt, device="cuda")
                L.check(getattr(lib, "mrgcn_basis_mix_fwd_" + sfx)(
                    plan.handle, Vt.data_ptr(), ct.data_ptr(), B, F, at.data_ptr(), ldA, M2.data_ptr(), ld, s))
            finally:
                L.set_config(**old)
            assert torch.equal(M.view(torch.int16 if dt == torch.bfloat16 else torch.int32),
                               M2.view(torch.int16 if dt == torch.bfloat16 else torch.int32)), sfx


def test_constant_wide_input_is_read_from_line_aligned_rows():
    """A wide fp32 feature matrix without gradient whose rows are not whole 128-byte lines (K = 155: 620 bytes) is
    copied ONCE into rows padded to whole lines and read from there (functional.line_aligned_rows): the same bits as
    reading it in place, one copy however many epochs, a new copy when the tensor changes in place, none for an input
    that takes a gradient."""
    import mrgcn_amd
    from mrgcn_amd import functional as Fn
    from mrgcn_amd.layers.graph import GraphConvolution
    from mrgcn_amd.plan import plan_of
    rng = np.random.default_rng(5)
    N, R, B, K, F = 700, 5, 4, 155, 10
    rows, cols, vals, _ = _oracle_layer_case(rng, N, R, B, K, F, 6 * N, 100)
    At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    torch.manual_seed(3)
    layer = GraphConvolution(K, F, R, N, num_bases=B, bias=True, input_layer=True, featureless=False).cuda()
    X = torch.randn(N, K, device="cuda")
    plan = plan_of(At, N, R)

    def run():
        layer.zero_grad(set_to_none=True)
        Y = layer._forward_fused(X, plan, relu=True)
        Y.sum().backward()
        return Y.detach().clone(), layer.weight_F.grad.clone()
    prev, Fn._XPAD = Fn._XPAD, False
    try:
        y0, g0 = run()
    finally:
        Fn._XPAD = prev
    mrgcn_amd.reset_stats()
    y1, g1 = run()
    y2, g2 = run()
    assert mrgcn_amd.stats().get("x_line_rows.copy") == 1
    assert torch.equal(y1, y0) and torch.equal(y2, y0)
    for g in (g1, g2):   # (this small layer's dW adds its chunks with float atomics: equal to rounding)
        torch.testing.assert_close(g, g0, rtol=1e-5, atol=1e-5 * float(g0.abs().max()))
    X.mul_(2.0)
    y3, _ = run()
    assert mrgcn_amd.stats().get("x_line_rows.copy") == 2 and not torch.equal(y3, y0)
    Xg = X.detach().clone().requires_grad_(True)
    mrgcn_amd.reset_stats()
    layer._forward_fused(Xg, plan, relu=True).sum().backward()
    assert mrgcn_amd.stats().get("x_line_rows.copy") is None and Xg.grad is not None


@pytest.mark.parametrize("B,F", [(40, 10), (5, 11), (64, 16)])
def test_mix_forward_in_ticket_order_equals_the_strided_walk(B, F):
    """`mix_tickets` (round 6: the waves of the resident mix forward draw tiles of steps in order from ticket counters)
    changes which wave computes a node, not what it computes: a graph large enough for a full grid (>= 8 192 nodes), with
    hub nodes of several tiles, f32 and bf16 operands, with and without the feature term, several counter / tile
    settings — the same bits as the strided walk (`mix_tickets=0`)."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphPlan
    rng = np.random.default_rng(B * F)
    N, R = 20000, 9
    rows, cols, vals, _ = _oracle_layer_case(rng, N, R, B, 1, F, 6 * N, 900)
    At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    plan = GraphPlan(At, N, R)
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator("cuda").manual_seed(F)
    V = torch.randn((N, B, F), device="cuda", generator=g)
    comp = torch.randn((R, B), device="cuda", generator=g)
    ldA = (F + 3) // 4 * 4
    add = torch.randn((plan.ncols, ldA), device="cuda", generator=g)

    def run(sfx, dt, with_add):
        M = torch.full((plan.nop, ldA), 7.0, dtype=dt, device="cuda")
        L.check(getattr(lib, "mrgcn_basis_mix_fwd_" + sfx)(plan.handle, V.data_ptr(), comp.data_ptr(), B, F,
                                                           add.data_ptr() if with_add else 0, ldA, M.data_ptr(), ldA, s))
        torch.cuda.synchronize()
        return M.view(torch.int16 if dt == torch.bfloat16 else torch.int32).clone()
    for sfx, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        for with_add in (False, True):
            old = L.set_config(mix_tickets=0)
            try:
                want = run(sfx, dt, with_add)
                for ctr, tile in ((12, 4), (1, 2), (64, 1), (8, 7)):
                    L.set_config(mix_tickets=ctr, mix_ticket_tile=tile)
                    assert torch.equal(run(sfx, dt, with_add), want), (sfx, with_add, ctr, tile)
            finally:
                L.set_config(**old)


@pytest.mark.parametrize("zero_frac", [0.0, 0.5, 0.93, 1.0])
@pytest.mark.parametrize("N,R,B,F,hub", [(900, 7, 40, 10, 500), (333, 5, 3, 11, 0), (640, 9, 64, 16, 200),
                                         (500, 6, 70, 12, 0)])  # the last one: B > 64, two-kernel fallback
def test_basis_mix_backward_with_rows_without_gradient(N, R, B, F, hub, zero_frac):
    """dV / dcomp through the C ABI (node-major V / dV) when most rows of dM carry no gradient — the state of
    a semi-supervised epoch (only columns within two hops of a label receive gradient).  Includes a node with
    more than 64 columns, NaN-poisoned dead rows, the squared norm that clip_grad_norm_ consumes, and the
    row-sparse form that leaves the blocks of nodes without gradient unwritten."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphPlan
    rng = np.random.default_rng(N + B + int(zero_frac * 100))
    rows, cols, vals, _ = _oracle_layer_case(rng, N, R, B, 1, F, 5 * N, hub)
    At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals),
                                 (N, R * N)).cuda()
    plan = GraphPlan(At, N, R)
    urel, unode = plan.export(L.ARR_UREL).astype(np.int64), plan.export(L.ARR_UNODE).astype(np.int64)
    nc, ld = plan.ncols, (F + 3) // 4 * 4
    dM = rng.standard_normal((nc, ld)).astype(np.float32)
    dead = rng.random(nc) < zero_frac
    dM[dead] = 0.0
    V = rng.standard_normal((N, B, F)).astype(np.float32)      # node-major
    comp = rng.standard_normal((R, B)).astype(np.float32)

    d64 = dM[:, :F].astype(np.float64)
    want_dV = np.zeros((N, B, F))
    np.add.at(want_dV, unode, comp.astype(np.float64)[urel][:, :, None] * d64[:, None, :])
    want_dc = np.zeros((R, B))
    np.add.at(want_dc, urel, np.einsum("cf,cbf->cb", d64, V.astype(np.float64)[unode]))
    node_live = np.zeros(N, dtype=bool)
    node_live[unode[~dead]] = True

    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    dMt, Vt, ct = (torch.from_numpy(x).cuda() for x in (dM, V, comp))
    # (a) no flags: every column counts, every block is written
    dV = torch.full((N, B, F), 7.0, device="cuda")
    dc = torch.full((R, B), 7.0, device="cuda")
    sq = torch.zeros((), dtype=torch.float64, device="cuda")
    L.check(lib.mrgcn_basis_mix_bwd_f32(plan.handle, dMt.data_ptr(), ld, 0, Vt.data_ptr(), ct.data_ptr(), B, F,
                                        dV.data_ptr(), 0, dc.data_ptr(), sq.data_ptr(), s))
    np.testing.assert_allclose(dV.cpu().numpy(), want_dV, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dc.cpu().numpy(), want_dc, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(float(sq), (want_dV ** 2).sum(), rtol=1e-4, atol=1e-6)
    if zero_frac == 1.0:
        assert not dV.any() and not dc.any() and float(sq) == 0.0
    # (b) with the producer's flags the dead rows are never read: poison them; dense dV (zeros for dead nodes)
    liveg = torch.from_numpy((~dead).astype(np.uint8)).cuda()
    dMp = dMt.clone()
    dMp[torch.from_numpy(dead).cuda()] = float("nan")
    dV.fill_(7.0); dc.fill_(7.0); sq.zero_()
    L.check(lib.mrgcn_basis_mix_bwd_f32(plan.handle, dMp.clone().data_ptr(), ld, liveg.data_ptr(), Vt.data_ptr(),
                                        ct.data_ptr(), B, F, dV.data_ptr(), 0, dc.data_ptr(), sq.data_ptr(), s))
    np.testing.assert_allclose(dV.cpu().numpy(), want_dV, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dc.cpu().numpy(), want_dc, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(float(sq), (want_dV ** 2).sum(), rtol=1e-4, atol=1e-6)
    # (c) row-sparse: blocks of nodes without a live column stay as they were, the flags say which
    dV.fill_(7.0); dc.fill_(7.0); sq.zero_()
    cur = torch.full((N,), 9, dtype=torch.uint8, device="cuda")
    L.check(lib.mrgcn_basis_mix_bwd_f32(plan.handle, dMp.clone().data_ptr(), ld, liveg.data_ptr(), Vt.data_ptr(),
                                        ct.data_ptr(), B, F, dV.data_ptr(), cur.data_ptr(), dc.data_ptr(),
                                        sq.data_ptr(), s))
    curh = cur.cpu().numpy()
    got = dV.cpu().numpy()
    assert set(np.unique(curh)) <= {0, 1}
    assert np.array_equal(curh.astype(bool) | ~node_live, np.ones(N, dtype=bool))  # every live node is written
    np.testing.assert_allclose(got[curh == 1], want_dV[curh == 1], rtol=1e-4, atol=1e-5)
    if B <= 64:  # the wave-per-node kernel skips exactly the nodes without a live column
        assert np.array_equal(curh.astype(bool), node_live)
        assert (got[curh == 0] == 7.0).all()
    np.testing.assert_allclose(dc.cpu().numpy(), want_dc, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(float(sq), (want_dV ** 2).sum(), rtol=1e-4, atol=1e-6)


def test_backward_is_the_same_on_the_sparse_and_the_general_transposed_product(monkeypatch):
    """The layer's backward picks mrgcn_spmm_transposed_live_f32 while few rows of dY are live
    and the general product otherwise (functional._LiveGauge); both must give the same gradients.
    Dense dY: the first call takes the sparse path (nothing known yet), the second the general one;
    sparse dY (few 'labelled' rows): both take the sparse path.  (The per-epoch marking path: the look-up that would
    send a plain dense gradient to a gradient support — functional._discovered_rows — is switched off here.)"""
    from mrgcn_amd import functional as Fn
    monkeypatch.setattr(Fn, "_DISCOVER", False)
    from mrgcn_amd.layers.graph import GraphConvolution
    from mrgcn_amd.plan import plan_of
    N, R, B, K, F = 1200, 6, 5, 9, 10
    rng = np.random.default_rng(5)
    rows, cols, vals, _ = _oracle_layer_case(rng, N, R, B, K, F, 6 * N, 400)
    At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    plan = plan_of(At, N, R)
    torch.manual_seed(0)
    layer = GraphConvolution(K, F, R, N, num_bases=B, bias=True, input_layer=True, featureless=False).cuda()
    X = torch.randn(N, K, device="cuda", requires_grad=True)
    for labelled in (N, 7):
        w = torch.zeros(N, F, device="cuda")
        w[torch.randperm(N, device="cuda")[:labelled]] = torch.randn(labelled, F, device="cuda")
        grads, paths = [], []
        plan.__dict__.pop("_gauges", None)  # a fresh start: once on the general product the count is only refreshed every 32 calls
        for _ in range(3):
            layer.zero_grad(); X.grad = None
            Y = layer._forward_fused(X, plan, relu=False)
            gauge = Fn._live_gauge(plan, F, False, X.device)
            before = int(gauge.host[0])
            paths.append(before < 0 or before <= 0.25 * N)
            (Y * w).sum().backward()
            torch.cuda.synchronize()
            grads.append([p.grad.clone() for p in layer.parameters() if p.grad is not None] + [X.grad.clone()])
        # the choice lags one step behind the data
        assert paths == ([True, False, False] if labelled == N else [True, True, True]), paths
        for g in grads[1:]:
            for a, b in zip(grads[0], g):
                torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["rgcn_smoke_ft_b5_norm_f32", "rgcn_small_fl_b0_bias_ref_int8",
                                  "rgcn_small_ft_b3_bias_norm_f32", "rgcn_small_fl_b3_bias_norm_f32"])
def test_epoch_steps_never_read_unwritten_gradient_rows(name):
    """Rows of dM without gradient are left unwritten by the transposed product; with dM
    pre-filled with NaNs the golden epochs must still come out (nothing may read those rows)."""
    from mrgcn_amd import functional as Fn
    if name not in util.rgcn_cases():
        pytest.skip("case not in the golden set")
    Fn._POISON_DEAD = True
    try:
        test_epoch_steps_vs_reference_goldens(name, None)
    finally:
        Fn._POISON_DEAD = False


def _sparse_label_problem(N=6000, R=3, seed=3, labelled=6):
    """A graph with low in-degree and a handful of labels: most of the node table never gets gradient."""
    rng = np.random.default_rng(seed)
    nnz = N
    rows = rng.integers(0, N, nnz); cols = rng.integers(0, (R - 1) * N, nnz)
    ident = np.arange(N)
    rows = np.concatenate([rows, ident]); cols = np.concatenate([cols, (R - 1) * N + ident])
    key = np.unique(rows.astype(np.int64) * (R * N) + cols)
    rows, cols = key // (R * N), key % (R * N)
    vals = rng.uniform(0.2, 1.0, len(rows)).astype(np.float32)
    idx = rng.choice(N, labelled, replace=False).astype(np.int64)
    y = rng.integers(0, 4, labelled).astype(np.int64)
    return rows, cols, vals, idx, y


def _train_rgcn(rows, cols, vals, N, R, idx, y, steps, row_sparse, graphed=False, seed=0, bases=5, opt_state=None,
                model_state=None, return_opt=False):
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.train import ClipAdam, GraphedTrainStep, train_step
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    torch.manual_seed(seed)
    model = RGCN([(6, 10, "mrgcn", torch.nn.ReLU()), (10, 4, "mrgcn", None)], R, N, bases, 0.0, False, True, False).cuda()
    if model_state is not None:
        model.load_state_dict(model_state)
    X = torch.randn((N, 6), device="cuda", generator=torch.Generator("cuda").manual_seed(seed + 1))
    ig, yg = torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0, capturable=graphed)
    if opt_state is not None:
        opt.load_state_dict(opt_state)
    losses = []
    if graphed:
        step = GraphedTrainStep(model, lambda: model(X, A), ig, yg, opt, warmup=1, row_sparse=row_sparse)
        losses = [float(step()) for _ in range(steps - 1)]
    else:
        for _ in range(steps):
            losses.append(float(train_step(model, lambda: model(X, A), ig, yg, opt, row_sparse=row_sparse)))
            wI = model.layers["layer_0"].weight_I
            assert (wI.grad is None) == (row_sparse is not False)
    torch.cuda.synchronize()
    out = ({k: v.clone() for k, v in model.state_dict().items()}, losses)
    return out + (opt, model) if return_opt else out


def test_row_sparse_weight_gradient_trains_exactly_like_the_dense_one():
    """train_step's default (the blocks of weight_I's gradient without any live node are neither written nor
    fed to Adam: mrgcn_adam_step_rows_f32) against row_sparse=False (dense gradient in .grad, plain Adam):
    same parameters after 4 epochs — eager and replayed from a hipGraph."""
    N, R = 6000, 3
    rows, cols, vals, idx, y = _sparse_label_problem(N, R)
    dense, ld = _train_rgcn(rows, cols, vals, N, R, idx, y, 4, False)
    sparse, ls, opt, model = _train_rgcn(rows, cols, vals, N, R, idx, y, 4, None, return_opt=True)
    ent = model.layers["layer_0"].weight_I._mrgcn_rows
    frac = float(ent["ever"].float().mean())
    assert 0.0 < frac < 0.5, frac          # most of the node table never gets gradient
    np.testing.assert_allclose(ls, ld, rtol=1e-6, atol=1e-7)
    for k in dense:
        torch.testing.assert_close(sparse[k], dense[k], rtol=1e-6, atol=1e-7, msg=k)
    graphed, lg = _train_rgcn(rows, cols, vals, N, R, idx, y, 4, None, graphed=True)
    np.testing.assert_allclose(lg, ld[1:], rtol=1e-5, atol=1e-6)
    for k in dense:
        torch.testing.assert_close(graphed[k], dense[k], rtol=1e-5, atol=1e-6, msg=k)


def test_row_sparse_adam_respects_moments_it_did_not_build():
    """Two dense epochs with labels L1, then the state goes — through state_dict()s in the reference's layout —
    to a fresh model / optimizer that trains row-sparse with labels L2: the moments the first phase built at
    nodes that L2 never reaches must keep moving those parameters (rows with non-zero moments count as `ever`)."""
    N, R = 4000, 3
    rows, cols, vals, idx1, y1 = _sparse_label_problem(N, R, seed=4, labelled=8)
    _, _, _, idx2, y2 = _sparse_label_problem(N, R, seed=9, labelled=5)
    res = []
    for second_sparse in (False, None):
        sd1, _, opt1, model1 = _train_rgcn(rows, cols, vals, N, R, idx1, y1, 2, False, return_opt=True)
        osd = opt1.state_dict()
        assert osd["state"][list(n for n, _ in model1.named_parameters()).index("layers.layer_0.weight_I")][
            "exp_avg"].shape == sd1["layers.layer_0.weight_I"].shape  # reference layout (B*N, out)
        sd2, l2 = _train_rgcn(rows, cols, vals, N, R, idx2, y2, 3, second_sparse, opt_state=osd, model_state=sd1)
        res.append((sd2, l2))
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-6, atol=1e-7)
    for k in res[0][0]:
        torch.testing.assert_close(res[0][0][k], res[1][0][k], rtol=1e-6, atol=1e-7, msg=k)


def test_capturable_optimizer_state_resumes_like_an_eager_run():
    """ClipAdam(capturable=True): the step counter lives on the device.  One eager + two replayed epochs, a
    checkpoint (model + optimizer state_dict()), and two more epochs resumed (a) captured again and (b) eagerly
    without capturable give the parameters of five eager epochs: the saved `step` is the true one and a
    resumed run takes its bias corrections from it."""
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.train import ClipAdam, GraphedTrainStep, train_step
    N, R = 3000, 3
    rows, cols, vals, idx, y = _sparse_label_problem(N, R, seed=6)
    want, lw = _train_rgcn(rows, cols, vals, N, R, idx, y, 5, None)
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    X = torch.randn((N, 6), device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    ig, yg = torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()
    mods = [(6, 10, "mrgcn", torch.nn.ReLU()), (10, 4, "mrgcn", None)]
    torch.manual_seed(0)
    m1 = RGCN(mods, R, N, 5, 0.0, False, True, False).cuda()
    o1 = ClipAdam(m1.parameters(), lr=0.01, max_norm=1.0, capturable=True)
    g1 = GraphedTrainStep(m1, lambda: m1(X, A), ig, yg, o1, warmup=1)
    g1(); g1()
    import copy
    msd, osd = {k: v.clone() for k, v in m1.state_dict().items()}, copy.deepcopy(o1.state_dict())
    assert all(int(st["step"]) == 3 for st in osd["state"].values())
    for capt in (True, False):
        m2 = RGCN(mods, R, N, 5, 0.0, False, True, False).cuda()
        m2.load_state_dict(msd)
        o2 = ClipAdam(m2.parameters(), lr=0.01, max_norm=1.0, capturable=capt)
        o2.load_state_dict(copy.deepcopy(osd))  # (torch keeps references to the tensors it is handed)
        if capt:
            g2 = GraphedTrainStep(m2, lambda: m2(X, A), ig, yg, o2, warmup=1)
            last = float(g2())
        else:
            for _ in range(2):
                last = float(train_step(m2, lambda: m2(X, A), ig, yg, o2))
        np.testing.assert_allclose(last, lw[4], rtol=1e-5, atol=1e-6)
        got = m2.state_dict()
        for k in want:
            torch.testing.assert_close(got[k], want[k], rtol=1e-5, atol=1e-6, msg=f"{k} capturable={capt}")
        assert all(int(st["step"]) == 5 for st in o2.state_dict()["state"].values())


def test_layer_used_twice_in_one_row_sparse_step_is_an_error():
    """The row-sparse gradient lives on the parameter and is overwritten by every backward: a forward_fn that
    runs the layer twice must not lose a contribution silently."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.train import ClipAdam, train_step
    N, R = 500, 3
    rows, cols, vals, idx, y = _sparse_label_problem(N, R, seed=2)
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    model = RGCN([(6, 10, "mrgcn", torch.nn.ReLU()), (10, 4, "mrgcn", None)], R, N, 5, 0.0, False, True, False).cuda()
    X = torch.randn((N, 6), device="cuda")
    opt = ClipAdam(model.parameters(), lr=0.01)
    ig, yg = torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()
    with pytest.raises(L.MrgcnError):
        train_step(model, lambda: model(X, A) + model(X, A), ig, yg, opt)
    loss = train_step(model, lambda: model(X, A) + model(X, A), ig, yg, opt, row_sparse=False)  # dense: fine
    assert np.isfinite(float(loss))


def test_renumbering_nodes_permutes_the_logits():
    """mrgcn_amd.data.reorder: renumbering the nodes (labelled neighbourhoods first) is a pure relabelling — the
    logits of the renumbered graph are the permuted logits of the original."""
    from mrgcn_amd.data import reorder
    from mrgcn_amd.models.rgcn import RGCN
    N, R = 3000, 4
    rows, cols, vals, idx, y = _sparse_label_problem(N, R, seed=8)
    order, inv = reorder.label_reach_order(rows, cols, N, R, idx, hops=2)
    rows2, cols2 = reorder.relabel_coo(rows, cols, N, inv)
    torch.manual_seed(0)
    mods = [(6, 10, "mrgcn", torch.nn.ReLU()), (10, 4, "mrgcn", None)]
    m1 = RGCN(mods, R, N, 5, 0.0, False, True, False).cuda()
    m2 = RGCN(mods, R, N, 5, 0.0, False, True, False).cuda()
    sd = m1.state_dict()
    B, F = 5, 10
    wI = sd["layers.layer_0.weight_I"].view(B, N, F)
    sd2 = dict(sd)
    sd2["layers.layer_0.weight_I"] = wI[:, torch.from_numpy(order).cuda()].reshape(B * N, F)
    m2.load_state_dict(sd2)
    X = torch.randn((N, 6), device="cuda")
    A1 = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    A2 = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows2, cols2])), torch.from_numpy(vals), (N, R * N)).cuda()
    with torch.no_grad():
        y1 = m1(X, A1)
        y2 = m2(X[torch.from_numpy(order).cuda()], A2)
    torch.testing.assert_close(y2, y1[torch.from_numpy(order).cuda()], rtol=1e-5, atol=1e-5)


def test_row_sparse_adam_through_the_c_abi():
    """mrgcn_adam_step_rows_f32 against mrgcn_adam_step_f32 on the rows it may not skip: rows that never had
    gradient are left alone bit for bit, rows with moments but no gradient this step take g = 0 without their
    (NaN-poisoned) gradient being read, `ever` picks up `cur`.  Several row lengths, eager and device-side step."""
    from mrgcn_amd import _lib as L
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator("cuda").manual_seed(3)
    for nrows, rowlen in ((1000, 400), (257, 16), (64, 1600), (5, 4), (300, 50), (7, 3)):
        p0 = torch.randn((nrows, rowlen), device="cuda", generator=gen)
        g = torch.randn((nrows, rowlen), device="cuda", generator=gen)
        m0 = torch.randn((nrows, rowlen), device="cuda", generator=gen) * 0.1
        v0 = torch.rand((nrows, rowlen), device="cuda", generator=gen) * 0.01
        cur = (torch.rand(nrows, device="cuda", generator=gen) < 0.4).to(torch.uint8)
        ever = ((torch.rand(nrows, device="cuda", generator=gen) < 0.5).to(torch.uint8) | 0)
        untouched = (cur == 0) & (ever == 0)
        m0[untouched] = 0; v0[untouched] = 0
        gz = g.clone(); gz[cur == 0] = 0
        coef = torch.tensor(0.7, device="cuda")
        # dense reference on copies
        pr, mr, vr = p0.clone(), m0.clone(), v0.clone()
        L.check(lib.mrgcn_adam_step_f32(pr.data_ptr(), gz.data_ptr(), mr.data_ptr(), vr.data_ptr(), pr.numel(), 0.01,
                                        0.9, 0.999, 1e-8, 0.0, 3, coef.data_ptr(), s))
        gp = g.clone(); gp[cur == 0] = float("nan")
        p, m, v, ev = p0.clone(), m0.clone(), v0.clone(), ever.clone()
        L.check(lib.mrgcn_adam_step_rows_f32(p.data_ptr(), gp.data_ptr(), m.data_ptr(), v.data_ptr(), nrows, rowlen,
                                             cur.data_ptr(), ev.data_ptr(), 0.01, 0.9, 0.999, 1e-8, 3, 0,
                                             coef.data_ptr(), s))
        touched = ~untouched
        assert torch.equal(p[untouched], p0[untouched]) and torch.equal(m[untouched], m0[untouched])
        assert torch.equal(p[touched], pr[touched]) and torch.equal(m[touched], mr[touched])
        assert torch.equal(v[touched], vr[touched])
        assert torch.equal(ev, ever | cur)


@pytest.mark.parametrize("N,R,B,F,hub,zero_frac", [(900, 7, 40, 10, 500, 0.6), (640, 9, 64, 16, 200, 0.9),
                                                   (500, 5, 16, 4, 0, 0.0), (333, 6, 48, 12, 100, 0.97)])
def test_adam_with_the_gradient_formed_on_the_fly_through_the_c_abi(N, R, B, F, hub, zero_frac):
    """mrgcn_basis_mix_bwd_f32(dV = NULL) + mrgcn_adam_step_rows_fused_f32 against the stored-gradient pair
    (mrgcn_basis_mix_bwd_f32 with dV + mrgcn_adam_step_rows_f32): the same flags, dcomp and squared norm, and
    bit-identical parameters / moments / `ever` — untouched nodes stay bit for bit, nodes with moments but no
    gradient this step decay.  NaN-poisoned dead rows of dM are never read; the coefficients are the snapshot."""
    import os
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphPlan
    if os.environ.get("MRGCN_FUSED_ADAM") == "0" or os.environ.get("MRGCN_MIX_NODE") == "0":
        pytest.skip("the fused update is switched off by the environment")
    rng = np.random.default_rng(N + B)
    rows, cols, vals, _ = _oracle_layer_case(rng, N, R, B, 1, F, 5 * N, hub)
    At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    plan = GraphPlan(At, N, R)
    lib = L.load()
    assert lib.mrgcn_adam_rows_fused_supported(plan.handle, B, F) == 1
    assert lib.mrgcn_adam_rows_fused_supported(plan.handle, 7, 3) == 0      # B F not a multiple of 4
    nc, ld = plan.ncols, (F + 3) // 4 * 4
    s = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator("cuda").manual_seed(N)
    dM = torch.randn((nc, ld), device="cuda", generator=gen)
    dead = torch.rand(nc, device="cuda", generator=gen) < zero_frac
    live = (~dead).to(torch.uint8)
    dM[dead] = float("nan")
    V = torch.randn((N, B, F), device="cuda", generator=gen)
    comp = torch.randn((R, B), device="cuda", generator=gen)
    m0 = torch.randn((N, B, F), device="cuda", generator=gen) * 0.1
    v0 = torch.rand((N, B, F), device="cuda", generator=gen) * 0.01
    ever0 = (torch.rand(N, device="cuda", generator=gen) < 0.3).to(torch.uint8)
    coef = torch.tensor(0.83, device="cuda")
    # stored-gradient reference
    g = torch.full((N, B, F), float("nan"), device="cuda")
    cur_r = torch.full((N,), 9, dtype=torch.uint8, device="cuda")
    dc_r = torch.empty((R, B), device="cuda")
    sq_r = torch.zeros((), dtype=torch.float64, device="cuda")
    L.check(lib.mrgcn_basis_mix_bwd_f32(plan.handle, dM.clone().data_ptr(), ld, live.data_ptr(), V.data_ptr(),
                                        comp.data_ptr(), B, F, g.data_ptr(), cur_r.data_ptr(), dc_r.data_ptr(),
                                        sq_r.data_ptr(), s))
    untouched = (cur_r == 0) & (ever0 == 0)
    m0[untouched] = 0; v0[untouched] = 0
    pr, mr, vr, er = V.clone(), m0.clone(), v0.clone(), ever0.clone()
    L.check(lib.mrgcn_adam_step_rows_f32(pr.data_ptr(), g.data_ptr(), mr.data_ptr(), vr.data_ptr(), N, B * F,
                                         cur_r.data_ptr(), er.data_ptr(), 0.01, 0.9, 0.999, 1e-8, 5, 0,
                                         coef.data_ptr(), s))
    # on the fly
    cur = torch.full((N,), 9, dtype=torch.uint8, device="cuda")
    dc = torch.empty((R, B), device="cuda")
    sq = torch.zeros((), dtype=torch.float64, device="cuda")
    L.check(lib.mrgcn_basis_mix_bwd_f32(plan.handle, dM.data_ptr(), ld, live.data_ptr(), V.data_ptr(), comp.data_ptr(),
                                        B, F, 0, cur.data_ptr(), dc.data_ptr(), sq.data_ptr(), s))
    assert torch.equal(cur, cur_r)
    np.testing.assert_allclose(dc.cpu().numpy(), dc_r.cpu().numpy(), rtol=1e-5, atol=1e-4)   # (LDS atomics: order)
    np.testing.assert_allclose(float(sq), float(sq_r), rtol=1e-6)
    p, m, v, ev = V.clone(), m0.clone(), v0.clone(), ever0.clone()
    L.check(lib.mrgcn_adam_step_rows_fused_f32(plan.handle, dM.data_ptr(), ld, live.data_ptr(), comp.data_ptr(), B, F,
                                               p.data_ptr(), m.data_ptr(), v.data_ptr(), cur.data_ptr(), ev.data_ptr(),
                                               0.01, 0.9, 0.999, 1e-8, 5, 0, coef.data_ptr(), s))
    assert torch.equal(ev, er)
    assert torch.equal(p, pr) and torch.equal(m, mr) and torch.equal(v, vr)
    assert torch.equal(p[untouched], V[untouched])
    if zero_frac > 0:
        assert int(untouched.sum()) > 0 and int(((cur == 0) & (ever0 == 1)).sum()) > 0


def test_layers_on_plans_with_many_narrow_and_wide_node_bands():
    """The relation-major orders of the transforms (common.hpp: RelOrder — wide node bands for wide inputs, narrow
    ones for inputs of <= 32 floats per row) only come apart on graphs of more than 32 768 nodes.  Here the band sizes
    are shrunk through the environment (wide 512 nodes, narrow 64) and the golden / oracle layer tests run again in a
    child process: forward, backward and epoch parity on plans where layer 0 (wide input) and layer 1 (narrow input)
    walk different orders with many bands each."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MRGCN_NODE_BAND="512", MRGCN_NODE_BAND_NARROW="64")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "tests/test_gpu_layers.py", "-k",
                        "forward_backward_vs_reference or fused_layer_vs_oracle or epoch_steps_vs_reference or "
                        "transform_backward_over_live"], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout


def test_input_gradient_handed_down_masked_and_flagged_equals_the_plain_hand_off():
    """A hidden layer's backward hands its input gradient to the layer below already multiplied by that layer's
    ReLU mask and with a byte per row (functional._grad_meta) — the layer below then skips its own masking pass and
    the scan for live rows.  Three stacked layers (two hand-offs), few labelled nodes: every gradient must be
    what the plain hand-off gives (the note ignored).  And when a hidden activation has TWO consumers
    autograd sums their gradients: the note of the first must not be trusted (version check) — again equal to the
    plain path."""
    from mrgcn_amd import functional as Fn
    from mrgcn_amd.layers.graph import GraphConvolution
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.plan import plan_of
    from mrgcn_amd.train import categorical_crossentropy
    N, R = 5000, 3
    rows, cols, vals, idx, y = _sparse_label_problem(N, R, labelled=8)
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    ig, yg = torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()
    torch.manual_seed(2)
    model = RGCN([(6, 10, "mrgcn", torch.nn.ReLU()), (10, 7, "mrgcn", torch.nn.ReLU()), (7, 4, "mrgcn", None)],
                 R, N, 4, 0.0, False, True, False).cuda()
    extra = GraphConvolution(10, 7, R, N, num_bases=4, bias=True).cuda()   # a second consumer of layer 0's output
    X = torch.randn((N, 6), device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    params = list(model.parameters()) + list(extra.parameters())

    def run(two_consumers, trust):
        for p in params:
            p.grad = None
        seen = []
        orig = Fn._grad_meta
        Fn._grad_meta = (lambda t: seen.append(orig(t)) or seen[-1]) if trust else (lambda t: None)
        try:
            if two_consumers:
                plan = plan_of(A, N, R)
                H = model.layers["layer_0"]._forward_fused(X, plan, relu=True)
                H2 = model.layers["layer_1"]._forward_fused(H, plan, relu=True) + Fn.rgcn_layer(plan, extra, H, relu=True)
                out = model.layers["layer_2"](H2, A)
            else:
                out = model(X, A)
            categorical_crossentropy(out, ig, yg).backward()
        finally:
            Fn._grad_meta = orig
        return [None if p.grad is None else p.grad.clone() for p in params], seen

    plain, _ = run(False, False)
    fast, seen = run(False, True)
    assert sum(m is not None and m["relu_applied"] for m in seen) == 2      # both hand-offs carried the note
    def same(xs, ys):   # (dcomp and the clip norm are summed with float atomics: equal up to their order)
        for a, b in zip(xs, ys):
            assert (a is None) == (b is None)
            if a is not None:
                torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-9)

    same(plain, fast)
    plain2, _ = run(True, False)
    fast2, seen2 = run(True, True)
    assert any(m is None for m in seen2)                                    # the summed gradient lost its note
    same(plain2, fast2)


@pytest.mark.parametrize("R,B,K,F", [(267, 40, 155, 10), (5, 3, 7, 4), (47, 30, 8, 16), (475, 2, 1, 200)])
def test_basis_contraction_of_weight_F_through_the_c_abi(R, B, K, F):
    """graph.py:83-85: W_F[r] = sum_b comp[r, b] V_F[b] and its backward on this package's kernels
    (mrgcn_basis_contract_f32 / _bwd_f32) against float64 torch; reproducible bit for bit."""
    from mrgcn_amd import functional as Fn
    g = torch.Generator("cuda").manual_seed(R)
    comp = torch.randn((R, B), device="cuda", generator=g, requires_grad=True)
    V = torch.randn((B, K, F), device="cuda", generator=g, requires_grad=True)
    w = torch.randn((R, K, F), device="cuda", generator=g)
    W = Fn._BasisContract.apply(comp, V)
    (W * w).sum().backward()
    c64, v64 = comp.detach().double().requires_grad_(True), V.detach().double().requires_grad_(True)
    W64 = (c64 @ v64.reshape(B, -1)).view(R, K, F)
    (W64 * w.double()).sum().backward()
    torch.testing.assert_close(W.detach().double(), W64.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(comp.grad.double(), c64.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(V.grad.double(), v64.grad, rtol=1e-4, atol=1e-4)
    assert torch.equal(Fn._BasisContract.apply(comp, V), W)


@pytest.mark.parametrize("N,C,n,scale", [(5000, 11, 1000, 1.0), (300, 2, 340, 0.25), (100000, 4, 40000, 3.0)])
def test_cross_entropy_with_compact_gradient_matches_torch(N, C, n, scale):
    """categorical_crossentropy (node_classification.py:439-444): loss and d loss / d logits against torch's
    CrossEntropyLoss in float64 — labelled nodes listed twice among them (n > N case: `Y.nonzero()` never does that,
    the kernel must still add), an upstream gradient other than 1, and the row flags that travel with the gradient."""
    from mrgcn_amd import functional as Fn
    from mrgcn_amd.train import categorical_crossentropy
    g = torch.Generator("cuda").manual_seed(N)
    buf = torch.randn((N, C + 1), device="cuda", generator=g)
    logits = buf[:, :C].clone().requires_grad_(True)           # dense; the strided form below
    idx = torch.randint(0, N, (n,), device="cuda", generator=g)
    tgt = torch.randint(0, C, (n,), device="cuda", generator=g)
    seen = {}
    loss = categorical_crossentropy(logits, idx, tgt)
    hook = logits.register_hook(lambda gr: seen.update(meta=Fn._grad_meta(gr)))
    (loss * scale).backward()
    hook.remove()
    l64 = logits.detach().double().requires_grad_(True)
    ref = torch.nn.CrossEntropyLoss()(l64[idx], tgt)
    (ref * scale).backward()
    torch.testing.assert_close(loss.double(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(logits.grad.double(), l64.grad, rtol=1e-4, atol=1e-7)
    flags = seen["meta"]["row_live"]
    want = torch.zeros(N, dtype=torch.uint8, device="cuda")
    want[idx] = 1
    assert torch.equal(flags, want) and not seen["meta"]["relu_applied"]
    strided = buf[:, :C].detach().requires_grad_(True)          # rows with a pad (a layer's padded output)
    torch.testing.assert_close(categorical_crossentropy(strided, idx, tgt), loss.detach(), rtol=1e-6, atol=1e-7)


def test_a_plain_dense_gradient_finds_its_gradient_support():
    """A loss built with torch's own ops hands the layer a dense gradient without a note.  The layer looks up its live
    rows (functional._discovered_rows), keeps their union as a structural row set and runs on the gradient support of
    that set: same gradients as the per-epoch marking path, also when a row of the set holds zeros in some epoch and
    when the set grows."""
    from mrgcn_amd import functional as Fn
    from mrgcn_amd.layers.graph import GraphConvolution
    from mrgcn_amd.plan import plan_of
    N, R, B, K, F = 1500, 5, 4, 7, 10
    rng = np.random.default_rng(11)
    rows, cols, vals, _ = _oracle_layer_case(rng, N, R, B, K, F, 6 * N, 300)
    At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    plan = plan_of(At, N, R)
    torch.manual_seed(0)
    layer = GraphConvolution(K, F, R, N, num_bases=B, bias=False, input_layer=True, featureless=False).cuda()
    X = torch.randn(N, K, device="cuda", requires_grad=True)
    idx = torch.randperm(N, device="cuda")[:40]
    sets = [idx, idx, idx[:30], torch.cat([idx, torch.tensor([3, 5], device="cuda")]), idx]   # steady, shrinks, grows, steady

    def grads(w):
        layer.zero_grad(); X.grad = None
        (layer._forward_fused(X, plan, relu=False) * w).sum().backward()
        return [p.grad.clone() for p in layer.parameters() if p.grad is not None] + [X.grad.clone()]

    taken = []
    orig = Fn._RgcnLayer._backward_on_support

    def spy(ctx, sup, dY, dbias):
        out = orig(ctx, sup, dY, dbias)
        taken.append(out is not None)
        return out
    for rows_ in sets:
        w = torch.zeros(N, F, device="cuda")
        w[rows_] = torch.randn(len(rows_), F, device="cuda")
        Fn._DISCOVER = False
        want = grads(w)
        Fn._DISCOVER = True
        Fn._RgcnLayer._backward_on_support = staticmethod(spy)
        try:
            got = grads(w)
        finally:
            Fn._RgcnLayer._backward_on_support = staticmethod(orig)
        for a, b in zip(want, got):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)   # (another summation order)
    assert taken == [True] * len(sets), taken
    assert int(layer.__dict__["_mrgcn_found_rows"].sum()) == 42   # the union of every set seen


@pytest.mark.gpu
@pytest.mark.parametrize("M,F,ld", [(1, 1, 1), (257, 3, 4), (5000, 10, 10), (5000, 11, 12), (70001, 16, 16), (70001, 7, 9)])
def test_bias_gradient_column_sums_over_flagged_rows_through_the_c_abi(M, F, ld):
    """mrgcn_colsum_rows_f32 = `dY.sum(0)` (the gradient of `+ self.b`, graph.py:98-101): every row, or the flagged rows
    only — unflagged rows are never read (they hold NaN here, like the unwritten rows of the loss's
    labelled-rows-only gradient); two calls give the same bits (fixed summation order)."""
    from mrgcn_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(M + F)
    X = rng.standard_normal((M, ld)).astype(np.float32)
    flags = (rng.random(M) < 0.3).astype(np.uint8)
    flags[rng.integers(0, M)] = 1
    Xg = torch.from_numpy(X).cuda()
    ws = torch.empty(int(lib.mrgcn_colsum_rows_workspace(F)), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    out = torch.full((F,), 7.0, device="cuda")
    L.check(lib.mrgcn_colsum_rows_f32(Xg.data_ptr(), ld, M, F, 0, out.data_ptr(), ws.data_ptr(), ws.numel(), s))
    np.testing.assert_allclose(out.cpu().numpy(), X[:, :F].astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)
    Xp = X.copy()
    Xp[flags == 0] = np.nan
    Xpg, fg = torch.from_numpy(Xp).cuda(), torch.from_numpy(flags).cuda()
    outs = []
    for _ in range(2):
        o = torch.empty(F, device="cuda")
        L.check(lib.mrgcn_colsum_rows_f32(Xpg.data_ptr(), ld, M, F, fg.data_ptr(), o.data_ptr(), ws.data_ptr(), ws.numel(), s))
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    np.testing.assert_allclose(outs[0].cpu().numpy(), X[flags == 1][:, :F].astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)
    assert lib.mrgcn_colsum_rows_workspace(17) < 0


# ---- the literal operand of a featureless layer without bases: compact-rows gradient (round 6) -------------------
def _train_featureless(rows, cols, vals, N, R, idx, y, steps, row_sparse, graphed=False, seed=0, hidden=16,
                       opt_state=None, model_state=None, weight_decay=0.0):
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.train import ClipAdam, GraphedTrainStep, train_step
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    torch.manual_seed(seed)
    model = RGCN([(N, hidden, "mrgcn", torch.nn.ReLU()), (hidden, 4, "mrgcn", None)], R, N, 0, 0.0, True, True,
                 False).cuda()
    if model_state is not None:
        model.load_state_dict(model_state)
    ig, yg = torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0, capturable=graphed, weight_decay=weight_decay)
    if opt_state is not None:
        opt.load_state_dict(opt_state)
    if graphed:
        step = GraphedTrainStep(model, lambda: model(None, A), ig, yg, opt, warmup=1, row_sparse=row_sparse)
        losses = [float(step()) for _ in range(steps - 1)]
    else:
        losses = [float(train_step(model, lambda: model(None, A), ig, yg, opt, row_sparse=row_sparse))
                  for _ in range(steps)]
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in model.state_dict().items()}, losses, opt, model


def test_literal_operand_gradient_in_compact_rows_trains_like_the_dense_one():
    """A featureless layer without bases (AIFB's shape): weight_I is the (R*N) x F operand of A . W itself
    (graph.py:69-75).  train_step's default keeps its gradient as the compact columns' rows and Adam touches those rows
    only (mrgcn_adam_step_index_rows_f32); row_sparse=False builds the dense (R*N) x F gradient and runs the dense
    Adam — same parameters after 4 epochs, eager and replayed; rows that are no column of A never move."""
    import mrgcn_amd
    N, R = 5000, 3
    rows, cols, vals, idx, y = _sparse_label_problem(N, R, labelled=40)
    dense, ld, _, m0 = _train_featureless(rows, cols, vals, N, R, idx, y, 4, False)
    assert m0.layers["layer_0"].weight_I.grad is not None
    mrgcn_amd.reset_stats()
    sparse, ls, opt, model = _train_featureless(rows, cols, vals, N, R, idx, y, 4, None)
    st = mrgcn_amd.stats()
    assert st.get("weight_I.index_rows") == 4 and st.get("adam.index_rows") == 4, st
    wI = model.layers["layer_0"].weight_I
    assert wI.grad is None and wI._mrgcn_rows["kind"] == "index"
    assert wI._mrgcn_rows["g"].shape[0] < 0.7 * wI.shape[0]     # (a good part of the table is no column of A)
    np.testing.assert_allclose(ls, ld, rtol=1e-6, atol=1e-7)
    for k in dense:
        torch.testing.assert_close(sparse[k], dense[k], rtol=1e-6, atol=1e-7, msg=k)
    # rows outside the compact columns: exactly the initial values, zero moments
    torch.manual_seed(0)
    from mrgcn_amd.models.rgcn import RGCN
    init = RGCN([(N, 16, "mrgcn", torch.nn.ReLU()), (16, 4, "mrgcn", None)], R, N, 0, 0.0, True, True, False).cuda()
    outside = torch.ones(wI.shape[0], dtype=torch.bool, device="cuda")
    outside[wI._mrgcn_rows["index"]] = False
    assert torch.equal(wI.detach()[outside], init.layers["layer_0"].weight_I.detach()[outside])
    assert not bool(opt.state[wI]["exp_avg"][outside].any())
    graphed, lg, _, _ = _train_featureless(rows, cols, vals, N, R, idx, y, 4, None, graphed=True)
    np.testing.assert_allclose(lg, ld[1:], rtol=1e-5, atol=1e-6)
    for k in dense:
        torch.testing.assert_close(graphed[k], dense[k], rtol=1e-5, atol=1e-6, msg=k)


def test_literal_compact_rows_give_way_to_moments_outside_their_columns():
    """An optimizer state with moments on a row that is no column of A (built elsewhere: a weight-decayed run, another
    graph): skipping that row would freeze a parameter that Adam keeps moving — the compact form steps aside, once, and
    the run equals the dense one.  With weight_decay the gradient is dense from the start."""
    import copy
    import mrgcn_amd
    N, R = 3000, 3
    rows, cols, vals, idx, y = _sparse_label_problem(N, R, seed=5, labelled=20)
    sd1, _, opt1, model1 = _train_featureless(rows, cols, vals, N, R, idx, y, 2, False)
    osd = copy.deepcopy(opt1.state_dict())
    names = [n for n, _ in model1.named_parameters()]
    k = names.index("layers.layer_0.weight_I")
    used = set((cols.astype(np.int64)).tolist())
    free = next(r for r in range(R * N) if r not in used)
    osd["state"][k]["exp_avg"][free] += 0.25
    osd["state"][k]["exp_avg_sq"][free] += 0.5
    res = []
    for rs in (False, None):
        mrgcn_amd.reset_stats()
        sd2, l2, _, m2 = _train_featureless(rows, cols, vals, N, R, idx, y, 3, rs, opt_state=copy.deepcopy(osd),
                                            model_state=sd1)
        if rs is None:
            st = mrgcn_amd.stats()
            assert st.get("adam.index_rows") is None and st.get("weight_I.index_rows") == 1, st
            assert m2.layers["layer_0"].weight_I._mrgcn_rows["dense_only"]
        res.append((sd2, l2))
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-6, atol=1e-7)
    for key in res[0][0]:
        torch.testing.assert_close(res[0][0][key], res[1][0][key], rtol=1e-6, atol=1e-7, msg=key)
    assert not torch.equal(res[1][0]["layers.layer_0.weight_I"][free], sd1["layers.layer_0.weight_I"][free])
    a = _train_featureless(rows, cols, vals, N, R, idx, y, 3, False, weight_decay=0.01)
    mrgcn_amd.reset_stats()
    b = _train_featureless(rows, cols, vals, N, R, idx, y, 3, None, weight_decay=0.01)
    assert mrgcn_amd.stats().get("adam.index_rows") is None
    for key in a[0]:
        torch.testing.assert_close(a[0][key], b[0][key], rtol=1e-6, atol=1e-7, msg=key)


def test_adam_on_indexed_rows_equals_the_dense_step_on_those_rows():
    """mrgcn_adam_step_index_rows_f32 against mrgcn_adam_step_f32 fed the scattered gradient: the indexed rows bit for
    bit, every other row untouched (also with a padded gradient row stride and a clip coefficient)."""
    from mrgcn_amd import _lib as L
    lib = L.load()
    gen = torch.Generator("cuda").manual_seed(3)
    nrows, F, n = 5000, 12, 700
    index = torch.randperm(nrows, device="cuda", generator=gen)[:n].to(torch.int32)
    p0 = torch.randn((nrows, F), device="cuda", generator=gen)
    m0 = torch.randn((nrows, F), device="cuda", generator=gen) * 0.1
    v0 = torch.rand((nrows, F), device="cuda", generator=gen) * 0.1
    gbuf = torch.randn((n, 16), device="cuda", generator=gen)
    coef = torch.tensor(0.37, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for ld, g in ((F, gbuf[:, :F].contiguous()), (16, gbuf)):
        dense_g = torch.zeros_like(p0)
        dense_g[index.long()] = g[:, :F]
        pd, md, vd = p0.clone(), m0.clone(), v0.clone()
        L.check(lib.mrgcn_adam_step_f32(pd.data_ptr(), dense_g.data_ptr(), md.data_ptr(), vd.data_ptr(), pd.numel(),
                                        0.01, 0.9, 0.999, 1e-8, 0.0, 3, coef.data_ptr(), s))
        pi, mi, vi = p0.clone(), m0.clone(), v0.clone()
        L.check(lib.mrgcn_adam_step_index_rows_f32(pi.data_ptr(), g.data_ptr(), ld, mi.data_ptr(), vi.data_ptr(),
                                                   index.data_ptr(), n, F, 0.01, 0.9, 0.999, 1e-8, 3, 0,
                                                   coef.data_ptr(), s))
        torch.cuda.synchronize()
        ii = index.long()
        assert torch.equal(pi[ii], pd[ii]) and torch.equal(mi[ii], md[ii]) and torch.equal(vi[ii], vd[ii])
        out = torch.ones(nrows, dtype=torch.bool, device="cuda")
        out[ii] = False
        assert torch.equal(pi[out], p0[out]) and torch.equal(mi[out], m0[out]) and torch.equal(vi[out], v0[out])
    with pytest.raises(L.MrgcnError):
        L.check(lib.mrgcn_adam_step_index_rows_f32(p0.data_ptr(), gbuf.data_ptr(), 16, m0.data_ptr(), v0.data_ptr(),
                                                   index.data_ptr(), n, 10, 0.01, 0.9, 0.999, 1e-8, 3, 0, 0, s))
