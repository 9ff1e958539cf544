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
        np.testing.assert_allclose(p.grad.cpu().numpy(), c["grad." + n], rtol=1e-3, atol=1e-5, err_msg=n)
    if not fl:
        np.testing.assert_allclose(X.grad.cpu().numpy(), c["grad.X"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("name", util.rgcn_cases())
def test_epoch_steps_vs_reference_goldens(name, defer=False):
    """zero_grad / backward / clip 1.0 / Adam driven for n_adam epochs
    (node_classification.py:166-193)."""
    from mrgcn_amd import functional as Fn
    from mrgcn_amd.train import ClipAdam, train_step
    prev = Fn.defer_input_grad(defer)
    try:
        _epoch_steps(name, ClipAdam, train_step, defer)
    finally:
        Fn.defer_input_grad(prev)


@pytest.mark.parametrize("name", [n for n in util.rgcn_cases() if "_b0_" not in n])
def test_epoch_steps_with_deferred_weight_I_update(name):
    """Same goldens with weight_I's gradient never stored: ||dV||^2 in the backward, Adam applied
    inside the kernel that recomputes dV (mrgcn_basis_mix_bwd_adam_f32)."""
    test_epoch_steps_vs_reference_goldens(name, defer=True)


def test_deferred_update_equals_the_stored_gradient_update():
    """The two update paths share their arithmetic (only the double-precision atomics that sum
    the gradient norm are unordered): parameters and Adam state after 3 steps agree to fp32
    rounding, with and without weight decay."""
    from mrgcn_amd import functional as Fn
    from mrgcn_amd.train import ClipAdam, train_step
    name = "rgcn_smoke_ft_b5_norm_f32"
    c = util.load_case(name)
    A = _adjacency(c, name)
    X = torch.from_numpy(c["X"]).cuda()
    idx = torch.from_numpy(c["labels_idx"]).cuda()
    tgt = torch.from_numpy(c["labels_y"]).cuda()
    for wd in (0.0, 0.05):
        out = []
        for defer in (False, True):
            model, _ = util.build_rgcn_from_case(c, "cuda")
            util.load_state_from_case(model, c)
            model = model.cuda()
            opt = ClipAdam(list(model.parameters()), lr=0.01, weight_decay=wd, max_norm=1.0)
            prev = Fn.defer_input_grad(defer)
            prev_nm, Fn._NODE_MAJOR = Fn._NODE_MAJOR, False  # "stored" = the dense gradient tensor
            try:
                for _ in range(3):
                    train_step(model, lambda: model(X, A), idx, tgt, opt)
                    wI = model.layers["layer_0"].weight_I
                    assert (wI.grad is None) == defer
            finally:
                Fn.defer_input_grad(prev)
                Fn._NODE_MAJOR = prev_nm
            st = opt.state[model.layers["layer_0"].weight_I]
            out.append(({k: v.clone() for k, v in model.state_dict().items()}, st["exp_avg"].clone(),
                        st["exp_avg_sq"].clone(), opt.last_grad_norm()))
        (sa, ma, va, na), (sb, mb, vb, nb) = out
        assert abs(na - nb) <= 1e-6 * na
        for k in sa:
            torch.testing.assert_close(sa[k], sb[k], rtol=1e-6, atol=1e-7, msg=k)
        torch.testing.assert_close(ma, mb, rtol=1e-6, atol=1e-9)
        torch.testing.assert_close(va, vb, rtol=1e-6, atol=1e-12)


def _epoch_steps(name, ClipAdam, train_step, defer):
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
        loss = train_step(model, lambda: model(X, A), idx, tgt, opt)
        np.testing.assert_allclose(float(loss), float(c[f"loss_step{step}"]), rtol=2e-4, atol=2e-5)
        if step == 1:
            np.testing.assert_allclose(opt.last_grad_norm(), float(c["grad_norm"]), rtol=1e-4)
        if step in (1, n_adam):
            sd = model.state_dict()
            for k in c.files:
                if k.startswith(f"adam{step}."):
                    key = k[len(f"adam{step}."):]
                    diff = np.abs(sd[key].cpu().numpy() - c[k])
                    # Adam's first steps move every element by ~lr*sign(g): elements whose fp32
                    # gradient is at rounding-noise level may differ by up to 2*lr per step
                    assert (diff > 2e-5).mean() < 2e-3, (k, float((diff > 2e-5).mean()))
                    assert diff.max() <= 0.021 * step, k


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
                np.testing.assert_allclose(p.grad.cpu().numpy(), c["grad." + n], rtol=1e-3, atol=1e-5,
                                           err_msg=n)
        opt.step()
        sd = model.state_dict()
        for k in c.files:
            if k.startswith(f"adam{step}."):
                diff = np.abs(sd[k[len(f"adam{step}."):]].cpu().numpy() - c[k])
                assert (diff > 2e-5).mean() < 5e-3 and diff.max() <= 0.021 * step, k


@pytest.mark.parametrize("N,R,B,F,bias", [(400, 5, 2, 200, False), (300, 7, 3, 130, True), (200, 4, 0, 96, True)])
def test_featureless_wide_layer_vs_oracle(N, R, B, F, bias):
    """FB15k-237-style encoder layer: featureless input layer with a wide hidden size
    (configs/fb15k-237.toml: 2 bases, hidden 200) — forward and every gradient."""
    from oracle import rgcn_oracle as O
    from mrgcn_amd.layers.graph import GraphConvolution
    from mrgcn_amd.plan import plan_of
    rng = np.random.default_rng(F)
    rows, cols, vals, A = _oracle_layer_case(rng, N, R, max(B, 1), 1, F, 8 * N, 150)
    At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals),
                                 (N, R * N)).cuda()
    torch.manual_seed(F)
    layer = GraphConvolution(0, F, R, N, num_bases=B, bias=bias, input_layer=True, featureless=True).cuda()
    Y = layer._forward_fused(None, plan_of(At, N, R), relu=True)
    w = torch.randn_like(Y)
    (Y * w).sum().backward()
    cfg = O.LayerCfg(0, F, R, N, B, bias=bias, input_layer=True, featureless=True)
    p = {k: v.detach().cpu().numpy() for k, v in layer.named_parameters()}
    pre, cache = O.layer_forward(cfg, p, None, A)
    np.testing.assert_allclose(Y.detach().cpu().numpy(), np.maximum(pre, 0), rtol=1e-4, atol=1e-4)
    grads, _ = O.layer_backward(cfg, p, None, A, w.cpu().numpy().astype(np.float64) * (pre > 0), cache)
    for k, v in grads.items():
        got = getattr(layer, k).grad.cpu().numpy()
        np.testing.assert_allclose(got, v, rtol=2e-4, atol=2e-5 * (np.abs(v).max() + 1e-12) + 1e-6, err_msg=k)


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
    p = {k: v.detach().cpu().numpy() for k, v in layer.named_parameters()}
    pre, cache = O.layer_forward(cfg, p, X.detach().cpu().numpy(), A)
    np.testing.assert_allclose(Y.detach().cpu().numpy(), np.maximum(pre, 0), rtol=1e-4, atol=1e-4)
    dPre = w.cpu().numpy().astype(np.float64) * (pre > 0)
    grads, dX = O.layer_backward(cfg, p, X.detach().cpu().numpy(), A, dPre, cache)
    scale = {k: np.abs(v).max() + 1e-12 for k, v in grads.items()}
    for k, v in grads.items():
        got = getattr(layer, k).grad.cpu().numpy()
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


@pytest.mark.parametrize("zero_frac", [0.0, 0.5, 0.93, 1.0])
@pytest.mark.parametrize("N,R,B,F,hub", [(900, 7, 40, 10, 500), (333, 5, 3, 11, 0), (640, 9, 64, 16, 200),
                                         (500, 6, 70, 12, 0)])  # the last one: B > 64, two-kernel fallback
def test_basis_mix_backward_with_rows_without_gradient(N, R, B, F, hub, zero_frac):
    """dV / dcomp through the C ABI when most rows of dM are exact zeros — the state of a
    semi-supervised epoch (only columns within two hops of a label receive gradient), which the
    wave-per-node kernel detects and skips.  Includes a node with more than 64 columns, negative
    zeros, and the squared norm that clip_grad_norm_ consumes."""
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
    dM[dead & (rng.random(nc) < 0.3)] = -0.0
    V = rng.standard_normal((B, N, F)).astype(np.float32)
    comp = rng.standard_normal((R, B)).astype(np.float32)

    d64 = dM[:, :F].astype(np.float64)
    want_dV = np.zeros((B, N, F))
    np.add.at(want_dV, (slice(None), unode), comp.astype(np.float64)[urel].T[:, :, None] * d64[None])
    want_dc = np.zeros((R, B))
    np.add.at(want_dc, urel, np.einsum("cf,bcf->cb", d64, V.astype(np.float64)[:, unode]))

    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    dMt, Vt, ct = (torch.from_numpy(x).cuda() for x in (dM, V, comp))
    dV = torch.full((B, N, F), 7.0, device="cuda")
    dc = torch.full((R, B), 7.0, device="cuda")
    sq = torch.zeros((), dtype=torch.float64, device="cuda")
    L.check(lib.mrgcn_basis_mix_bwd_f32(plan.handle, dMt.data_ptr(), ld, Vt.data_ptr(), ct.data_ptr(), B, F,
                                        dV.data_ptr(), dc.data_ptr(), sq.data_ptr(), s))
    np.testing.assert_allclose(dV.cpu().numpy(), want_dV, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dc.cpu().numpy(), want_dc, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(float(sq), (want_dV ** 2).sum(), rtol=1e-4, atol=1e-6)
    if zero_frac == 1.0:
        assert not dV.any() and not dc.any() and float(sq) == 0.0

    # norm-only form (deferred update): same dcomp and norm, no dV
    dc2 = torch.full((R, B), 7.0, device="cuda")
    sq.zero_()
    L.check(lib.mrgcn_basis_mix_bwd_f32(plan.handle, dMt.data_ptr(), ld, Vt.data_ptr(), ct.data_ptr(), B, F,
                                        0, dc2.data_ptr(), sq.data_ptr(), s))
    np.testing.assert_allclose(dc2.cpu().numpy(), want_dc, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(float(sq), (want_dV ** 2).sum(), rtol=1e-4, atol=1e-6)

    # with the producer's flags the dead rows are never read: poison them
    liveg = torch.from_numpy((~dead).astype(np.uint8)).cuda()
    dMp = dMt.clone()
    dMp[torch.from_numpy(dead).cuda()] = float("nan")
    for dv_ptr in (dV.data_ptr(), 0):
        sq.zero_()
        L.check(lib.mrgcn_basis_mix_bwd_live_f32(plan.handle, dMp.data_ptr(), ld, liveg.data_ptr(), 0, Vt.data_ptr(),
                                                 ct.data_ptr(), B, F, dv_ptr, dc.data_ptr(), sq.data_ptr(), s))
        np.testing.assert_allclose(dc.cpu().numpy(), want_dc, rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(float(sq), (want_dV ** 2).sum(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(dV.cpu().numpy(), want_dV, rtol=1e-4, atol=1e-5)

    # a NaN anywhere in a row keeps that row alive
    if zero_frac > 0 and dead.any():
        c = int(np.flatnonzero(dead)[0])
        dMt[c, F - 1] = float("nan")
        L.check(lib.mrgcn_basis_mix_bwd_f32(plan.handle, dMt.data_ptr(), ld, Vt.data_ptr(), ct.data_ptr(), B, F,
                                            dV.data_ptr(), dc.data_ptr(), 0, s))
        assert torch.isnan(dV[:, int(unode[c]), F - 1]).all()
        assert torch.isnan(dc[int(urel[c])]).all()


def test_backward_is_the_same_on_the_sparse_and_the_general_transposed_product():
    """The layer's backward picks mrgcn_spmm_transposed_live_f32 while few rows of dY are live
    and the general product otherwise (functional._LiveGauge); both must give the same gradients.
    Dense dY: the first call takes the sparse path (nothing known yet), the second the general one;
    sparse dY (few 'labelled' rows): both take the sparse path."""
    from mrgcn_amd import functional as Fn
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
        Fn._GAUGES.clear()  # a fresh start: once on the general product the count is only refreshed every 32 calls
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
        test_epoch_steps_vs_reference_goldens(name)
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


def _train_rgcn(rows, cols, vals, N, R, idx, y, steps, sparse, graphed=False, seed=0, node_major=False, bases=5):
    from mrgcn_amd import functional as Fn
    from mrgcn_amd import train as T
    from mrgcn_amd.models.rgcn import RGCN
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    torch.manual_seed(seed)
    dims = [(N, 10), (10, 4)]
    modules = [(i, o, "mrgcn", torch.nn.ReLU() if li == 0 else None) for li, (i, o) in enumerate(dims)]
    model = RGCN(modules, R, N, bases, 0.0, True, False, False).cuda()
    opt = T.ClipAdam(model.parameters(), lr=0.01, weight_decay=0.0, max_norm=1.0, capturable=graphed)
    it, tg = torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()
    prev, prev_nm = T._SPARSE_WGRAD_DEFAULT, Fn._NODE_MAJOR
    T._SPARSE_WGRAD_DEFAULT, Fn._NODE_MAJOR = sparse, node_major
    try:
        losses = []
        if graphed:
            step = T.GraphedTrainStep(model, lambda: model(None, A), it, tg, opt, warmup=2)
            losses = [float(step()) for _ in range(steps - 2)]  # the two warm-up steps count
        else:
            losses = [float(T.train_step(model, lambda: model(None, A), it, tg, opt)) for _ in range(steps)]
        with torch.no_grad():
            logits = model(None, A).cpu().numpy()
    finally:
        T._SPARSE_WGRAD_DEFAULT, Fn._NODE_MAJOR = prev, prev_nm
    return losses, logits, {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, opt


@pytest.mark.parametrize("reordered", [False, True])
def test_chunk_sparse_weight_gradient_trains_exactly_like_the_dense_one(reordered):
    """functional.sparse_weight_grad: chunks of weight_I's gradient without any live node are neither
    written nor read, Adam never touches chunks that never had gradient — the parameters after
    several epochs are bit for bit those of the dense path; also through a captured hipGraph, and
    on a graph renumbered with data.reorder (logits permute with the nodes)."""
    from mrgcn_amd import functional as Fn
    from mrgcn_amd.data import reorder
    N, R = 6000, 3
    rows, cols, vals, idx, y = _sparse_label_problem(N, R)
    if reordered:
        order, inv = reorder.label_reach_order(rows, cols, N, R, idx, hops=2)
        rows, cols = reorder.relabel_coo(rows, cols, N, inv)
        idx = inv[idx]
    dense = _train_rgcn(rows, cols, vals, N, R, idx, y, 5, sparse=False)
    Fn._WCHUNKS.clear()
    sparse = _train_rgcn(rows, cols, vals, N, R, idx, y, 5, sparse=True)
    ent = next(iter(Fn._WCHUNKS.values()))
    ever = ent["ever"].cpu().numpy()
    assert 0 < ever.sum() < len(ever) // 2, "the test graph must leave most chunks without gradient"
    if reordered:
        assert ever[: ever.sum()].all(), "reachable nodes first: the live chunks are a prefix"
    # (dcomp is summed with float atomics: the clip norm, hence every update, may differ in the last
    # bits between any two runs — the tolerance is for that, not for the skipping)
    tol = dict(rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(np.asarray(sparse[0]), np.asarray(dense[0]), **tol)
    for k in dense[2]:
        np.testing.assert_allclose(sparse[2][k], dense[2][k], err_msg=k, **tol)
    for (pd, sd), (ps, ss) in zip(dense[3].state.items(), sparse[3].state.items()):
        for key in ("exp_avg", "exp_avg_sq"):
            torch.testing.assert_close(ss[key], sd[key], **tol)
        if ps.numel() == ent["numel"]:  # weight_I: chunks that never had gradient were never touched
            dead = torch.from_numpy(np.repeat(ever == 0, 1024)[: ent["slab"]]).cuda()
            for key in ("exp_avg", "exp_avg_sq"):
                assert not ss[key].view(ent["B"], -1)[:, dead].any()
                assert not sd[key].view(ent["B"], -1)[:, dead].any()  # ... and the dense run agrees: all zeros
    Fn._WCHUNKS.clear()
    graphed = _train_rgcn(rows, cols, vals, N, R, idx, y, 5, sparse=True, graphed=True)
    for k in dense[2]:
        np.testing.assert_allclose(graphed[2][k], dense[2][k], err_msg=k, **tol)


def test_renumbering_nodes_permutes_the_logits():
    from mrgcn_amd.data import reorder
    N, R = 6000, 3
    rows, cols, vals, idx, y = _sparse_label_problem(N, R)
    base = _train_rgcn(rows, cols, vals, N, R, idx, y, 1, sparse=False)
    order, inv = reorder.label_reach_order(rows, cols, N, R, idx, hops=2)
    assert sorted(order.tolist()) == list(range(N)) and (order[inv] == np.arange(N)).all()
    r2, c2 = reorder.relabel_coo(rows, cols, N, inv)
    # the same model on the renumbered graph: permute the node table accordingly
    from mrgcn_amd import train as T
    from mrgcn_amd.models.rgcn import RGCN
    A2 = torch.sparse_coo_tensor(torch.from_numpy(np.stack([r2, c2])), torch.from_numpy(vals), (N, R * N)).cuda()
    torch.manual_seed(0)
    dims = [(N, 10), (10, 4)]
    modules = [(i, o, "mrgcn", torch.nn.ReLU() if li == 0 else None) for li, (i, o) in enumerate(dims)]
    model = RGCN(modules, R, N, 5, 0.0, True, False, False).cuda()
    with torch.no_grad():
        first = next(iter(model.layers.values()))
        w = first.weight_I.view(5, N, 10)
        first.weight_I.copy_(w[:, torch.from_numpy(order).cuda(), :].reshape(5 * N, 10).clone())
        logits2 = model(None, A2).cpu().numpy()
    # `base` took one training step before its logits were read: compare the untrained forward instead
    torch.manual_seed(0)
    model0 = RGCN(modules, R, N, 5, 0.0, True, False, False).cuda()
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    with torch.no_grad():
        logits0 = model0(None, A).cpu().numpy()
    np.testing.assert_allclose(logits2[inv], logits0, rtol=1e-5, atol=1e-6)


def test_chunked_adam_and_chunk_flags_through_the_c_abi():
    """mrgcn_weight_chunks_live marks the 1024-float chunks of a basis slab that hold a node with a live
    column; mrgcn_basis_mix_bwd_live_f32 leaves dV untouched in the other chunks;
    mrgcn_adam_step_chunked_f32 == mrgcn_adam_step_f32 given that the gradient is zero outside the
    `cur` chunks and the moments are zero outside the `ever` chunks — and it touches nothing there."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphPlan
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(9)
    N, R, B, F = 3000, 4, 6, 10
    rows, cols, vals, _ = _oracle_layer_case(rng, N, R, B, 1, F, 3 * N, 0)
    At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    plan = GraphPlan(At, N, R)
    unode = plan.export(L.ARR_UNODE).astype(np.int64)
    live_nodes = np.zeros(N, bool); live_nodes[rng.choice(N, 40, replace=False)] = True
    col_live = live_nodes[unode] & (rng.random(plan.ncols) < 0.7)
    node_has = np.zeros(N, bool); node_has[unode[col_live]] = True
    nch = int(lib.mrgcn_weight_chunks(plan.handle, F))
    assert nch == (N * F + 1023) // 1024
    want = np.zeros(nch, np.uint8)
    for j in np.flatnonzero(node_has):
        want[(j * F) // 1024:(j * F + F - 1) // 1024 + 1] = 1
    cur = torch.full((nch,), 9, dtype=torch.uint8, device="cuda")
    ever = torch.zeros(nch, dtype=torch.uint8, device="cuda"); ever[0] = 1
    clg = torch.from_numpy(col_live.astype(np.uint8)).cuda()
    L.check(lib.mrgcn_weight_chunks_live(plan.handle, clg.data_ptr(), F, cur.data_ptr(), ever.data_ptr(), s))
    np.testing.assert_array_equal(cur.cpu().numpy(), want)
    w_ever = want.copy(); w_ever[0] = 1
    np.testing.assert_array_equal(ever.cpu().numpy(), w_ever)

    # dV stays untouched (NaN here) in dead chunks, equals the dense kernel's in live ones
    ld = 12
    dM = rng.standard_normal((plan.ncols, ld)).astype(np.float32); dM[~col_live] = np.nan
    V = torch.from_numpy(rng.standard_normal((B * N, F)).astype(np.float32)).cuda()
    comp = torch.from_numpy(rng.standard_normal((R, B)).astype(np.float32)).cuda()
    dMg = torch.from_numpy(dM).cuda()
    outs = []
    for chunk_ptr in (0, cur.data_ptr()):
        dV = torch.full((B * N, F), float("nan"), device="cuda")
        dc = torch.empty((R, B), device="cuda")
        sq = torch.zeros((), dtype=torch.float64, device="cuda")
        L.check(lib.mrgcn_basis_mix_bwd_live_f32(plan.handle, dMg.data_ptr(), ld, clg.data_ptr(), chunk_ptr,
                                                 V.data_ptr(), comp.data_ptr(), B, F, dV.data_ptr(), dc.data_ptr(),
                                                 sq.data_ptr(), s))
        outs.append((dV.view(B, N * F).cpu().numpy(), dc.cpu().numpy(), float(sq)))
    elem_live = np.repeat(want.astype(bool), 1024)[: N * F]
    assert not np.isnan(outs[0][0]).any()
    np.testing.assert_array_equal(outs[1][0][:, elem_live], outs[0][0][:, elem_live])
    dead_part = outs[1][0][:, ~elem_live]  # untouched, or zeros where a 4-node group straddles a live chunk
    assert (np.isnan(dead_part) | (dead_part == 0)).all() and np.isnan(dead_part).mean() > 0.9
    assert not outs[0][0][:, ~elem_live].any()
    np.testing.assert_allclose(outs[1][1], outs[0][1], rtol=1e-4, atol=1e-5 * np.abs(outs[0][1]).max() + 1e-6)
    np.testing.assert_allclose(outs[1][2], outs[0][2], rtol=1e-6)

    # Adam: three steps, `cur` changing, against the plain kernel
    n = B * N * F
    p0 = torch.randn(n, device="cuda")
    pa, pb = p0.clone(), p0.clone()
    ma, va, mb, vb = (torch.zeros(n, device="cuda") for _ in range(4))
    coef = torch.full((), 0.5, device="cuda")
    ever_t = torch.zeros(nch, dtype=torch.uint8, device="cuda")
    for step in (1, 2, 3):
        cur_np = (rng.random(nch) < 0.3).astype(np.uint8)
        cur_t = torch.from_numpy(cur_np).cuda()
        ever_t |= cur_t
        mask = torch.from_numpy(np.repeat(cur_np.astype(bool), 1024)[: N * F]).cuda()
        g = torch.randn(B, N * F, device="cuda") * mask
        g_sparse = torch.where(mask, g, torch.full_like(g, float("nan")))  # unwritten where not `cur`
        L.check(lib.mrgcn_adam_step_f32(pa.data_ptr(), g.data_ptr(), ma.data_ptr(), va.data_ptr(), n, 0.01, 0.9, 0.999,
                                        1e-8, 0.0, step, coef.data_ptr(), s))
        L.check(lib.mrgcn_adam_step_chunked_f32(pb.data_ptr(), g_sparse.data_ptr(), mb.data_ptr(), vb.data_ptr(), N * F, B,
                                                cur_t.data_ptr(), ever_t.data_ptr(), 0.01, 0.9, 0.999, 1e-8, step, 0,
                                                coef.data_ptr(), s))
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    never = torch.from_numpy(np.repeat(ever_t.cpu().numpy() == 0, 1024)[: N * F]).cuda()
    assert torch.equal(pb.view(B, -1)[:, never], p0.view(B, -1)[:, never])


def test_chunk_sparse_adam_respects_moments_it_did_not_build():
    """Steps on the plain path first (moments everywhere the labels of that phase reached), then the
    chunk-sparse path with other labels: chunks that hold non-zero moments keep being updated."""
    from mrgcn_amd import functional as Fn
    from mrgcn_amd import train as T
    from mrgcn_amd.models.rgcn import RGCN
    N, R = 6000, 3
    rows, cols, vals, idx, y = _sparse_label_problem(N, R)
    _, _, _, idx2, y2 = _sparse_label_problem(N, R, seed=17)
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    dims = [(N, 10), (10, 4)]
    modules = [(i, o, "mrgcn", torch.nn.ReLU() if li == 0 else None) for li, (i, o) in enumerate(dims)]
    results = []
    for second_phase_sparse in (False, True):
        Fn._WCHUNKS.clear()
        torch.manual_seed(1)
        model = RGCN(modules, R, N, 5, 0.0, True, False, False).cuda()
        opt = T.ClipAdam(model.parameters(), lr=0.01, weight_decay=0.0, max_norm=1.0)
        prev = T._SPARSE_WGRAD_DEFAULT
        try:
            T._SPARSE_WGRAD_DEFAULT = False
            for _ in range(2):
                T.train_step(model, lambda: model(None, A), torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda(), opt)
            T._SPARSE_WGRAD_DEFAULT = second_phase_sparse
            for _ in range(3):
                T.train_step(model, lambda: model(None, A), torch.from_numpy(idx2).cuda(), torch.from_numpy(y2).cuda(), opt)
        finally:
            T._SPARSE_WGRAD_DEFAULT = prev
        results.append({k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    assert Fn._WCHUNKS, "the second phase of the second run must have used the masks"
    for k in results[0]:
        np.testing.assert_allclose(results[1][k], results[0][k], rtol=1e-5, atol=1e-8, err_msg=k)


@pytest.mark.parametrize("N,R,B,F", [(3000, 4, 6, 10), (1037 * 4, 5, 40, 10), (2048, 3, 8, 16), (1000, 3, 3, 4)])
def test_node_major_gradient_and_adam_through_the_c_abi(N, R, B, F):
    """mrgcn_basis_mix_bwd_nodemajor_f32 writes dV as [N][B][F] for the nodes with a live column only
    (flags in node_cur) — the same numbers as the dense [B][N][F] kernel; mrgcn_adam_step_nodemajor_f32
    on node-major gradient / moments == mrgcn_adam_step_f32 on the dense ones, nodes that never had
    gradient untouched, blocks of nodes without gradient this step never read."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphPlan
    lib = L.load()
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(N + B)
    rows, cols, vals, _ = _oracle_layer_case(rng, N, R, B, 1, F, 3 * N, 0)
    At = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    plan = GraphPlan(At, N, R)
    assert lib.mrgcn_nodemajor_supported(plan.handle, B, F) == 1
    unode = plan.export(L.ARR_UNODE).astype(np.int64)
    live_nodes = rng.random(N) < 0.3
    col_live = live_nodes[unode] & (rng.random(plan.ncols) < 0.7)
    node_has = np.zeros(N, bool); node_has[unode[col_live]] = True
    ld = (F + 3) // 4 * 4
    dM = rng.standard_normal((plan.ncols, ld)).astype(np.float32); dM[~col_live] = np.nan
    V = torch.from_numpy(rng.standard_normal((B * N, F)).astype(np.float32)).cuda()
    comp = torch.from_numpy(rng.standard_normal((R, B)).astype(np.float32)).cuda()
    dMg = torch.from_numpy(dM).cuda()
    clg = torch.from_numpy(col_live.astype(np.uint8)).cuda()
    # dense reference
    dV = torch.empty((B * N, F), device="cuda"); dc = torch.empty((R, B), device="cuda")
    sq = torch.zeros((), dtype=torch.float64, device="cuda")
    L.check(lib.mrgcn_basis_mix_bwd_live_f32(plan.handle, dMg.data_ptr(), ld, clg.data_ptr(), 0, V.data_ptr(),
                                             comp.data_ptr(), B, F, dV.data_ptr(), dc.data_ptr(), sq.data_ptr(), s))
    # node-major
    dVn = torch.full((N, B, F), float("nan"), device="cuda"); dc2 = torch.empty((R, B), device="cuda")
    cur = torch.full((N,), 9, dtype=torch.uint8, device="cuda")
    sq2 = torch.zeros((), dtype=torch.float64, device="cuda")
    L.check(lib.mrgcn_basis_mix_bwd_nodemajor_f32(plan.handle, dMg.data_ptr(), ld, clg.data_ptr(), V.data_ptr(),
                                                  comp.data_ptr(), B, F, dVn.data_ptr(), cur.data_ptr(),
                                                  dc2.data_ptr(), sq2.data_ptr(), s))
    np.testing.assert_array_equal(cur.cpu().numpy(), node_has.astype(np.uint8))
    want = dV.view(B, N, F).permute(1, 0, 2)
    nh = torch.from_numpy(node_has).cuda()
    assert torch.equal(dVn[nh], want[nh])
    assert torch.isnan(dVn[~nh]).all() and not want[~nh].any()
    dcn = dc.cpu().numpy()  # summed with float atomics: the order, hence the last bits, differ from run to run
    np.testing.assert_allclose(dc2.cpu().numpy(), dcn, rtol=1e-4, atol=1e-5 * np.abs(dcn).max() + 1e-6)
    np.testing.assert_allclose(float(sq2), float(sq), rtol=1e-6)

    # Adam: three steps with changing `cur`
    n = B * N * F
    p0 = torch.randn(n, device="cuda")
    pa, pb = p0.clone(), p0.clone()
    ma, va = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    mb, vb = torch.zeros((N, B, F), device="cuda"), torch.zeros((N, B, F), device="cuda")
    coef = torch.full((), 0.5, device="cuda")
    ever = torch.zeros(N, dtype=torch.uint8, device="cuda")
    for step in (1, 2, 3):
        cur_t = (torch.rand(N, device="cuda") < 0.3).to(torch.uint8)
        ever |= cur_t
        g = torch.randn(B, N, F, device="cuda") * cur_t.view(1, N, 1)
        g_nm = torch.where(cur_t.bool().view(N, 1, 1), g.permute(1, 0, 2), torch.full((N, B, F), float("nan"), device="cuda")).contiguous()
        L.check(lib.mrgcn_adam_step_f32(pa.data_ptr(), g.data_ptr(), ma.data_ptr(), va.data_ptr(), n, 0.01, 0.9, 0.999,
                                        1e-8, 0.0, step, coef.data_ptr(), s))
        L.check(lib.mrgcn_adam_step_nodemajor_f32(pb.data_ptr(), g_nm.data_ptr(), mb.data_ptr(), vb.data_ptr(), N, B, F,
                                                  cur_t.data_ptr(), ever.data_ptr(), 0.01, 0.9, 0.999, 1e-8, step, 0,
                                                  coef.data_ptr(), s))
        assert torch.equal(pa, pb)
        assert torch.equal(ma.view(B, N, F).permute(1, 0, 2), mb) and torch.equal(va.view(B, N, F).permute(1, 0, 2), vb)
    never = ever == 0
    assert torch.equal(pb.view(B, N, F)[:, never], p0.view(B, N, F)[:, never])


def test_node_major_optimizer_space_trains_exactly_like_the_dense_path():
    """functional._NODE_MAJOR (default): weight_I's gradient goes to ClipAdam as [N][B][F] blocks of the
    nodes with gradient, the moments live in the same layout, weight_I.grad stays None.  Parameters,
    losses and the optimizer's state_dict() (handed out in the reference layout) after several epochs
    equal the dense path's — eager, captured, and when the moments were first built on the plain path."""
    from mrgcn_amd import functional as Fn
    from mrgcn_amd import train as T
    N, R = 6000, 3
    rows, cols, vals, idx, y = _sparse_label_problem(N, R)
    tol = dict(rtol=1e-5, atol=1e-8)
    dense = _train_rgcn(rows, cols, vals, N, R, idx, y, 5, sparse=False, bases=6)  # B*F must be a multiple of 4
    Fn._NODEMAJOR.clear()
    nm = _train_rgcn(rows, cols, vals, N, R, idx, y, 5, sparse=True, node_major=True, bases=6)
    ent = next(iter(Fn._NODEMAJOR.values()))
    ever = ent["ever"].cpu().numpy()
    assert 0 < ever.sum() < N // 2
    np.testing.assert_allclose(np.asarray(nm[0]), np.asarray(dense[0]), **tol)
    for k in dense[2]:
        np.testing.assert_allclose(nm[2][k], dense[2][k], err_msg=k, **tol)
    sd_d, sd_n = dense[3].state_dict(), nm[3].state_dict()
    assert sd_d["state"].keys() == sd_n["state"].keys()
    for k in sd_d["state"]:
        assert "node_major" not in sd_n["state"][k]
        for key in ("exp_avg", "exp_avg_sq"):
            assert sd_n["state"][k][key].shape == sd_d["state"][k][key].shape
            torch.testing.assert_close(sd_n["state"][k][key], sd_d["state"][k][key], **tol)
    Fn._NODEMAJOR.clear()
    graphed = _train_rgcn(rows, cols, vals, N, R, idx, y, 5, sparse=True, node_major=True, graphed=True, bases=6)
    for k in dense[2]:
        np.testing.assert_allclose(graphed[2][k], dense[2][k], err_msg=k, **tol)
    # the optimizer state round-trips through state_dict / load_state_dict into a new optimizer,
    # and plain steps may follow node-major ones (and the other way round)
    from mrgcn_amd.models.rgcn import RGCN
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    dims = [(N, 10), (10, 4)]
    modules = [(i, o, "mrgcn", torch.nn.ReLU() if li == 0 else None) for li, (i, o) in enumerate(dims)]
    it, tg = torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()
    finals = []
    for schedule in ([False] * 6, [True, True, False, False, True, True]):
        Fn._NODEMAJOR.clear()
        torch.manual_seed(2)
        model = RGCN(modules, R, N, 6, 0.0, True, False, False).cuda()
        opt = T.ClipAdam(model.parameters(), lr=0.01, weight_decay=0.0, max_norm=1.0)
        prev, prev_nm = T._SPARSE_WGRAD_DEFAULT, Fn._NODE_MAJOR
        try:
            for k, sparse in enumerate(schedule):
                T._SPARSE_WGRAD_DEFAULT, Fn._NODE_MAJOR = sparse, True
                T.train_step(model, lambda: model(None, A), it, tg, opt)
                if k == 3:  # hand the state over to a fresh optimizer
                    sd = opt.state_dict()
                    opt = T.ClipAdam(model.parameters(), lr=0.01, weight_decay=0.0, max_norm=1.0)
                    opt.load_state_dict(sd)
        finally:
            T._SPARSE_WGRAD_DEFAULT, Fn._NODE_MAJOR = prev, prev_nm
        finals.append({k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    for k in finals[0]:
        np.testing.assert_allclose(finals[1][k], finals[0][k], err_msg=k, **tol)
