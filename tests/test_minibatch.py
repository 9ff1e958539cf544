"""Mini-batch path (reference mrgcn/data/batch.py:150-316, mrgcn/models/rgcn.py:
91-128, mrgcn/layers/graph.py:62-102 with A_idx) against goldens captured from the reference
(tests/golden/make_minibatch_goldens.py): batch structure bit-exact on the host, logits / loss /
gradients of both engines on the GPU."""
import os

import numpy as np
import pytest
import torch

from tests import util

GOLD = os.path.join(os.path.dirname(__file__), "golden", "minibatch_small.npz")
TAGS = ["ft_b3", "fl_b0", "ft_b0_l3", "fl_b2_l1"]


def _batch(g, tag, value_mode="ref_int8"):
    from mrgcn_amd.data import batch as mb
    _, A = util.load_graph("graph_small")
    meta = g[tag + ".meta"]
    fl, nl = bool(meta[0]), int(meta[3])
    X = None if fl else [g[tag + ".X_full"]]
    return mb.MiniBatch(A, X, g["batch_idx"], nl, value_mode=value_mode), A


@pytest.mark.parametrize("tag", TAGS)
def test_batch_structure_bit_exact(tag):
    from mrgcn_amd.data import batch as mb
    g = np.load(GOLD)
    b, A = _batch(g, tag)
    nl = int(g[tag + ".meta"][3])
    for i in range(nl):
        assert np.array_equal(b.A.neighbours[i], g[f"{tag}.neighbours_{i}"])
        assert list(b.A.row[i].shape) == list(g[f"{tag}.row_{i}.shape"])
    b.as_tensors_()
    for i in range(nl):  # the int8 COO of every row slice: indices and truncated values
        assert np.array_equal(b.A.row[i]._indices().numpy(), g[f"{tag}.row_{i}.indices"])
        assert np.array_equal(b.A.row[i]._values().numpy(), g[f"{tag}.row_{i}.values"])
        assert b.A.row[i].dtype == torch.int8
    if not bool(g[tag + ".meta"][0]):
        assert np.allclose(b.X[0].numpy(), g[tag + ".X_sub"])
        N = A.shape[0]
        idx = mb.getAdjacencyNodeColumnIdx(b.A.neighbours[0], N, A.shape[1] // N)
        assert np.array_equal(idx.numpy(), g[tag + ".A_idx_0"])
        sl = mb.sliceSparseCOO(b.A.row[0], idx)
        assert np.array_equal(sl._indices().numpy(), g[tag + ".sliced_0.indices"])
        assert np.array_equal(sl._values().numpy(), g[tag + ".sliced_0.values"])  # all ones (quirk A-3)
        assert list(sl.shape) == list(g[tag + ".sliced_0.shape"])


def test_neighbours_edge_cases():
    import scipy.sparse as sp
    from mrgcn_amd.data import batch as mb
    A = sp.csr_matrix(np.array([[0, 1, 0, 0, 0, 1], [0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0]], dtype=np.float32))
    assert mb.getNeighboursSparse(A, [1]).tolist() == []                 # isolated row
    assert mb.getNeighboursSparse(A, [0]).tolist() == [1, 2]             # columns 1 and 5 -> nodes 1, 2
    assert mb.getNeighboursSparse(A, [0, 2, 0]).tolist() == [0, 1, 2]    # duplicates in the sample
    t = torch.sparse_coo_tensor(torch.tensor([[0, 0, 1], [1, 5, 3]]), torch.tensor([0, 1, 1], dtype=torch.int8), (2, 6))
    sl = mb.sliceSparseCOO(t, torch.tensor([1, 3]))
    assert sl._indices().tolist() == [[0, 1], [0, 1]] and sl._values().tolist() == [1.0, 1.0]
    assert mb.sliceSparseCOO(t, torch.zeros(0, dtype=torch.long)).shape == (2, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["fused", "literal"])
@pytest.mark.parametrize("tag", TAGS)
def test_minibatch_model_vs_reference(tag, engine):
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.train import categorical_crossentropy
    g = np.load(GOLD)
    fl, B, bias, nl, hidden, classes, xw = [int(v) for v in g[tag + ".meta"]]
    _, A = util.load_graph("graph_small")
    N = A.shape[0]
    R = A.shape[1] // N
    dims = [(xw if li == 0 else hidden, hidden if li < nl - 1 else classes) for li in range(nl)]
    modules = [(i, o, "mrgcn", torch.nn.ReLU() if li < nl - 1 else None) for li, (i, o) in enumerate(dims)]
    model = RGCN(modules, R, N, B, 0.0, bool(fl), bool(bias), False)
    model.load_state_dict({k[len(tag) + 6:]: torch.from_numpy(np.array(g[k])) for k in g.files
                           if k.startswith(tag + ".init.")})
    model = model.cuda()
    model.set_engine(engine)
    b, _ = _batch(g, tag)
    b.as_tensors_()
    b.A.to(torch.device("cuda"))
    X = None if fl else b.X[0].float().cuda().requires_grad_(True)
    for epoch in range(2):  # the second pass reuses the cached plans / slices
        model.zero_grad()
        if X is not None:
            X.grad = None
        logits = model(X, b.A)
        np.testing.assert_allclose(logits.detach().cpu().numpy(), g[tag + ".logits"], rtol=1e-4, atol=1e-4)
        idx = torch.arange(len(g["batch_idx"]), device="cuda")
        loss = categorical_crossentropy(logits, idx, torch.from_numpy(g[tag + ".y"]).cuda())
        assert abs(float(loss.detach()) - float(g[tag + ".loss"])) < 1e-5
        loss.backward()
        for n, p in model.named_parameters():
            np.testing.assert_allclose(util.ref_layout(p.grad, n).cpu().numpy(), g[f"{tag}.grad.{n}"], rtol=1e-3, atol=1e-5,
                                       err_msg=n)
        if X is not None:
            np.testing.assert_allclose(X.grad.cpu().numpy(), g[tag + ".grad.X"], rtol=1e-3, atol=1e-5)


@pytest.mark.gpu
def test_mrgcn_minibatch_boundary():
    """MRGCN(MiniBatch) featureless: same logits as RGCN on the A_Batch, batch object moved with
    `to(devices)` as the reference's run loop does."""
    from mrgcn_amd.data import batch as mb
    from mrgcn_amd.models.mrgcn import MRGCN
    g = np.load(GOLD)
    tag = "fl_b0"
    fl, B, bias, nl, hidden, classes, xw = [int(v) for v in g[tag + ".meta"]]
    _, A = util.load_graph("graph_small")
    N = A.shape[0]
    R = A.shape[1] // N
    modules = [(0, hidden, "mrgcn", torch.nn.ReLU()), (hidden, classes, "mrgcn", None)]
    model = MRGCN(modules, [], R, N, num_bases=B, p_dropout=0.0, featureless=True, bias=False,
                  gcn_gpu_acceleration=True)
    model.load_state_dict({"rgcn." + k[len(tag) + 6:]: torch.from_numpy(np.array(g[k])) for k in g.files
                           if k.startswith(tag + ".init.")}, strict=False)
    batch = mb.MiniBatch(A, [np.empty((N, 0))], g["batch_idx"], nl)
    batch.as_tensors_()
    batch.to(model.devices)
    logits = model(batch)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g[tag + ".logits"], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_mrgcn_minibatch_boundary_on_the_full_plan():
    """The same boundary with the batch as a masked batch on the full graph's plan: MiniBatch(plan=...)."""
    from mrgcn_amd.data import batch as mb
    from mrgcn_amd.data.batch import scipy_sparse_to_pytorch_sparse
    from mrgcn_amd.models.mrgcn import MRGCN
    from mrgcn_amd.plan import plan_of
    g = np.load(GOLD)
    tag = "fl_b0"
    fl, B, bias, nl, hidden, classes, xw = [int(v) for v in g[tag + ".meta"]]
    _, A = util.load_graph("graph_small")
    N = A.shape[0]
    R = A.shape[1] // N
    modules = [(0, hidden, "mrgcn", torch.nn.ReLU()), (hidden, classes, "mrgcn", None)]
    model = MRGCN(modules, [], R, N, num_bases=B, p_dropout=0.0, featureless=True, bias=False,
                  gcn_gpu_acceleration=True)
    model.load_state_dict({"rgcn." + k[len(tag) + 6:]: torch.from_numpy(np.array(g[k])) for k in g.files
                           if k.startswith(tag + ".init.")}, strict=False)
    plan = plan_of(scipy_sparse_to_pytorch_sparse(A, dtype=torch.int8).cuda(), N, R)
    batch = mb.MiniBatch(None, [np.empty((N, 0))], g["batch_idx"], nl, plan=plan)
    batch.as_tensors_()
    batch.to(model.devices)
    logits = model(batch)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g[tag + ".logits"], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["ft_b3", "ft_b0_l3"])
def test_device_built_batch_equals_host_built(tag):
    """A_BatchDevice (gathers + torch.unique on the GPU) against the reference's goldens: same
    neighbour sets, same COO of every row slice (indices, int8 values), same model output."""
    from mrgcn_amd.data import batch as mb
    from mrgcn_amd.models.rgcn import RGCN
    g = np.load(GOLD)
    _, A = util.load_graph("graph_small")
    fl, B, bias, nl, hidden, classes, xw = [int(v) for v in g[tag + ".meta"]]
    ab = mb.A_BatchDevice(mb.DeviceCSR(A), g["batch_idx"], nl)
    for i in range(nl):
        assert np.array_equal(ab.neighbours[i].cpu().numpy(), g[f"{tag}.neighbours_{i}"])
        assert np.array_equal(ab.row[i]._indices().cpu().numpy(), g[f"{tag}.row_{i}.indices"])
        assert np.array_equal(ab.row[i]._values().cpu().numpy(), g[f"{tag}.row_{i}.values"])
        assert ab.row[i].dtype == torch.int8
    # the column-compacted slice the frontier kernel emits next to the row slice = the reference's
    # sliceSparseCOO(row_0, A_idx_0)
    a_idx, sl = ab.row[0]._mrgcn_slice
    assert a_idx is ab._a_idx[0] and np.array_equal(a_idx.cpu().numpy(), g[tag + ".A_idx_0"])
    assert np.array_equal(sl._indices().cpu().numpy(), g[tag + ".sliced_0.indices"])
    assert np.array_equal(sl._values().cpu().numpy(), g[tag + ".sliced_0.values"])
    assert list(sl.shape) == list(g[tag + ".sliced_0.shape"])
    N = A.shape[0]
    R = A.shape[1] // N
    dims = [(xw if li == 0 else hidden, hidden if li < nl - 1 else classes) for li in range(nl)]
    modules = [(i, o, "mrgcn", torch.nn.ReLU() if li < nl - 1 else None) for li, (i, o) in enumerate(dims)]
    model = RGCN(modules, R, N, B, 0.0, bool(fl), bool(bias), False)
    model.load_state_dict({k[len(tag) + 6:]: torch.from_numpy(np.array(g[k])) for k in g.files
                           if k.startswith(tag + ".init.")})
    model = model.cuda()
    X = torch.from_numpy(g[tag + ".X_full"]).cuda()[ab.neighbours[-1]]
    logits = model(X, ab)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g[tag + ".logits"], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["ft_b3", "ft_b0_l3", "fl_b0"])
def test_short_lived_batch_on_lean_plans_equals_the_goldens(tag):
    """A batch built for one step (A_BatchDevice(short_lived=True): its slices take the quick plan build,
    MRGCN_PLAN_LEAN) gives the reference's logits, loss and gradients like a batch on full plans."""
    from mrgcn_amd.data import batch as mb
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.train import categorical_crossentropy
    g = np.load(GOLD)
    _, A = util.load_graph("graph_small")
    fl, B, bias, nl, hidden, classes, xw = [int(v) for v in g[tag + ".meta"]]
    N = A.shape[0]
    R = A.shape[1] // N
    dims = [(xw if li == 0 else hidden, hidden if li < nl - 1 else classes) for li in range(nl)]
    modules = [(i, o, "mrgcn", torch.nn.ReLU() if li < nl - 1 else None) for li, (i, o) in enumerate(dims)]
    model = RGCN(modules, R, N, B, 0.0, bool(fl), bool(bias), False)
    model.load_state_dict({k[len(tag) + 6:]: torch.from_numpy(np.array(g[k])) for k in g.files
                           if k.startswith(tag + ".init.")})
    model = model.cuda()
    dcsr = mb.DeviceCSR(A)
    for _ in range(2):  # a fresh batch object (fresh plans) each time
        ab = mb.A_BatchDevice(dcsr, g["batch_idx"], nl, short_lived=True)
        X = None if fl else torch.from_numpy(g[tag + ".X_full"]).cuda()[ab.neighbours[-1]]
        model.zero_grad()
        logits = model(X, ab)
        plans = [t._mrgcn_plan for a in ab.row for t in (a, a._mrgcn_slice[1]) if getattr(t, "_mrgcn_plan", None)]
        assert plans and all(p.lean for p in plans)
        np.testing.assert_allclose(logits.detach().cpu().numpy(), g[tag + ".logits"], rtol=1e-4, atol=1e-4)
        idx = torch.arange(len(g["batch_idx"]), device="cuda")
        loss = categorical_crossentropy(logits, idx, torch.from_numpy(g[tag + ".y"]).cuda())
        assert abs(float(loss.detach()) - float(g[tag + ".loss"])) < 1e-5
        loss.backward()
        for n, p in model.named_parameters():
            np.testing.assert_allclose(util.ref_layout(p.grad, n).cpu().numpy(), g[f"{tag}.grad.{n}"], rtol=1e-3, atol=1e-5,
                                       err_msg=n)


def _golden_model(g, tag, A):
    from mrgcn_amd.models.rgcn import RGCN
    fl, B, bias, nl, hidden, classes, xw = [int(v) for v in g[tag + ".meta"]]
    N = A.shape[0]
    R = A.shape[1] // N
    dims = [(xw if li == 0 else hidden, hidden if li < nl - 1 else classes) for li in range(nl)]
    modules = [(i, o, "mrgcn", torch.nn.ReLU() if li < nl - 1 else None) for li, (i, o) in enumerate(dims)]
    model = RGCN(modules, R, N, B, 0.0, bool(fl), bool(bias), False)
    model.load_state_dict({k[len(tag) + 6:]: torch.from_numpy(np.array(g[k])) for k in g.files
                           if k.startswith(tag + ".init.")})
    return model.cuda(), fl, nl, N, R


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_masked_batch_on_the_full_plan_equals_the_goldens(tag):
    """A_BatchMasked: the batch as row sets on the FULL graph's plan (no slices, no per-batch plans; csrc/masked.hip) —
    the reference's neighbour sets, logits, loss and gradients."""
    from mrgcn_amd.data import batch as mb
    from mrgcn_amd.data.batch import scipy_sparse_to_pytorch_sparse
    from mrgcn_amd.plan import plan_of
    from mrgcn_amd.train import categorical_crossentropy
    g = np.load(GOLD)
    _, A = util.load_graph("graph_small")
    model, fl, nl, N, R = _golden_model(g, tag, A)
    plan = plan_of(scipy_sparse_to_pytorch_sparse(A, dtype=torch.int8).cuda(), N, R)   # the reference's boundary cast
    Xfull = None if fl else torch.from_numpy(g[tag + ".X_full"]).cuda()
    for _ in range(2):  # a fresh batch object each time, the one plan
        ab = mb.A_BatchMasked(plan, g["batch_idx"], nl)
        for i in range(nl):
            assert np.array_equal(ab.neighbours[i].cpu().numpy(), g[f"{tag}.neighbours_{i}"])
        X = None if fl else Xfull[ab.neighbours[-1]].requires_grad_(True)
        model.zero_grad()
        logits = model(X, ab)
        np.testing.assert_allclose(logits.detach().cpu().numpy(), g[tag + ".logits"], rtol=1e-4, atol=1e-4)
        idx = torch.arange(len(g["batch_idx"]), device="cuda")
        loss = categorical_crossentropy(logits, idx, torch.from_numpy(g[tag + ".y"]).cuda())
        assert abs(float(loss.detach()) - float(g[tag + ".loss"])) < 1e-5
        loss.backward()
        for n, p in model.named_parameters():
            np.testing.assert_allclose(util.ref_layout(p.grad, n).cpu().numpy(), g[f"{tag}.grad.{n}"], rtol=1e-3, atol=1e-5,
                                       err_msg=n)
        if X is not None:
            np.testing.assert_allclose(X.grad.cpu().numpy(), g[tag + ".grad.X"], rtol=1e-3, atol=1e-5)
            # the whole feature matrix instead of the neighbours' rows: the transform picks the rows itself
            Xw = Xfull.clone().requires_grad_(True)
            logits2 = model(Xw, ab)
            assert torch.equal(logits2, logits)
            logits2.sum().backward()
            gx = Xw.grad
            assert bool((gx[ab.neighbours[-1]].abs().sum() > 0)) and float(gx.abs().sum()) == pytest.approx(
                float(gx[ab.neighbours[-1]].abs().sum()), rel=1e-6)
        ab.close()


@pytest.mark.gpu
def test_masked_batch_rows_follow_the_batch_order_with_repeats():
    """Output rows follow batch_idx (unsorted, with a repeated node) exactly like A[sample] does on the slice path."""
    from mrgcn_amd.data import batch as mb
    from mrgcn_amd.data.batch import scipy_sparse_to_pytorch_sparse
    from mrgcn_amd.plan import plan_of
    g = np.load(GOLD)
    tag = "ft_b3"
    _, A = util.load_graph("graph_small")
    model, fl, nl, N, R = _golden_model(g, tag, A)
    idx = np.asarray(g["batch_idx"])
    shuffled = np.concatenate([idx[::-1], idx[:2]])
    plan = plan_of(scipy_sparse_to_pytorch_sparse(A, dtype=torch.int8).cuda(), N, R)
    ab = mb.A_BatchMasked(plan, shuffled, nl)
    X = torch.from_numpy(g[tag + ".X_full"]).cuda()[ab.neighbours[-1]]
    logits = model(X, ab).detach().cpu().numpy()
    want = np.concatenate([g[tag + ".logits"][::-1], g[tag + ".logits"][:2]])
    np.testing.assert_allclose(logits, want, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_masked_batches_train_like_slice_batches():
    """Re-sampled batches through ClipAdam (row-sparse weight_I update on the support's node flags): the same losses
    and parameters as the slice path with per-batch plans."""
    from mrgcn_amd.data import batch as mb
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.plan import GraphPlan
    from mrgcn_amd.train import ClipAdam, categorical_crossentropy
    import scipy.sparse as sp
    from mrgcn_amd import synth
    g = synth.make_graph("aifb", seed=1, scale=0.2)
    N, R = g.num_nodes, g.num_relations
    A = sp.csr_matrix((g.vals, (g.rows, g.cols)), shape=(N, R * N))
    rng = np.random.default_rng(3)
    steps, nb, K, C = 8, 24, 6, 4
    idxs = [np.sort(rng.choice(N, nb, replace=False)) for _ in range(steps)]
    ys = [torch.from_numpy(rng.integers(0, C, nb)).cuda() for _ in range(steps)]
    X = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).cuda()
    dcsr = mb.DeviceCSR(A)
    plan = GraphPlan.from_csr(A, N, R, value_mode="norm_f32")
    rows = torch.arange(nb, device="cuda")

    def run(masked):
        torch.manual_seed(0)
        model = RGCN([(K, 8, "mrgcn", torch.nn.ReLU()), (8, C, "mrgcn", None)], R, N, 3, 0.0, False, True, False).cuda()
        opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0)
        losses = []
        for k, i in enumerate(idxs):
            ab = (mb.A_BatchMasked(plan, i, 2) if masked
                  else mb.A_BatchDevice(dcsr, i, 2, value_mode="norm_f32", short_lived=True))
            loss = categorical_crossentropy(model(X[ab.neighbours[-1]], ab), rows, ys[k])
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        torch.cuda.synchronize()
        return [float(l) for l in losses], {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}

    l0, p0 = run(False)
    l1, p1 = run(True)
    np.testing.assert_allclose(l1, l0, rtol=1e-5, atol=1e-6)
    for n in p0:
        np.testing.assert_allclose(p1[n], p0[n], rtol=1e-4, atol=1e-6, err_msg=n)

    # the package's own step (train_step: weight_I's gradient stays row-sparse on the support's node flags, fused
    # clip + Adam on those rows), X handed over whole
    from mrgcn_amd.train import train_step
    torch.manual_seed(0)
    model = RGCN([(K, 8, "mrgcn", torch.nn.ReLU()), (8, C, "mrgcn", None)], R, N, 3, 0.0, False, True, False).cuda()
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0)
    l2 = []
    for k, i in enumerate(idxs):
        ab = mb.A_BatchMasked(plan, i, 2)
        l2.append(float(train_step(model, lambda: model(X, ab), rows, ys[k], opt)))
        ab.close()
    np.testing.assert_allclose(l2, l0, rtol=1e-5, atol=1e-6)
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), p0[n], rtol=1e-4, atol=1e-6, err_msg=n)

    # the reference's own loop shape (zero_grad / backward / clip_grad_norm_ / step) with the drop-in optimizer pair
    from mrgcn_amd import optim as fast
    torch.manual_seed(0)
    model = RGCN([(K, 8, "mrgcn", torch.nn.ReLU()), (8, C, "mrgcn", None)], R, N, 3, 0.0, False, True, False).cuda()
    opt = fast.RowSparseAdam(model.parameters(), lr=0.01)
    l3 = []
    for k, i in enumerate(idxs):
        ab = mb.A_BatchMasked(plan, i, 2)
        opt.zero_grad()
        loss = categorical_crossentropy(model(X[ab.neighbours[-1]], ab), rows, ys[k])
        loss.backward()
        fast.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        l3.append(float(loss))
        ab.close()
    np.testing.assert_allclose(l3, l0, rtol=1e-5, atol=1e-6)
    for n, p in model.named_parameters():
        np.testing.assert_allclose(p.detach().cpu().numpy(), p0[n], rtol=1e-4, atol=1e-6, err_msg=n)


@pytest.mark.gpu
def test_prefetched_batches_train_like_batches_built_in_line():
    """BatchPrefetcher (next batch + its slice plans built by a worker thread on its own stream, plans of finished
    steps released behind an event): the same losses and parameters as building every batch in line."""
    from mrgcn_amd.data import batch as mb
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.train import ClipAdam, categorical_crossentropy
    import scipy.sparse as sp
    from mrgcn_amd import synth
    g = synth.make_graph("aifb", seed=1, scale=0.2)
    N, R = g.num_nodes, g.num_relations
    A = sp.csr_matrix((g.vals, (g.rows, g.cols)), shape=(N, R * N))
    rng = np.random.default_rng(3)
    steps, nb, K, C = 8, 24, 6, 4
    idxs = [np.sort(rng.choice(N, nb, replace=False)) for _ in range(steps)]
    ys = [torch.from_numpy(rng.integers(0, C, nb)).cuda() for _ in range(steps)]
    X = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).cuda()
    dcsr = mb.DeviceCSR(A)
    rows = torch.arange(nb, device="cuda")

    def run(prefetch):
        torch.manual_seed(0)
        model = RGCN([(K, 8, "mrgcn", torch.nn.ReLU()), (8, C, "mrgcn", None)], R, N, 3, 0.0, False, True, False).cuda()
        opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0)
        it = (mb.BatchPrefetcher(dcsr, iter(idxs), 2, value_mode="norm_f32", model=model) if prefetch
              else (mb.A_BatchDevice(dcsr, i, 2, value_mode="norm_f32", short_lived=True) for i in idxs))
        losses = []
        for k, ab in enumerate(it):
            loss = categorical_crossentropy(model(X[ab.neighbours[-1]], ab), rows, ys[k])
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(loss.detach())
        if prefetch:
            it.close()
        torch.cuda.synchronize()
        return [float(l) for l in losses], {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}

    l0, p0 = run(False)
    l1, p1 = run(True)
    assert len(l1) == steps
    np.testing.assert_allclose(l1, l0, rtol=1e-5, atol=1e-6)
    for n in p0:
        np.testing.assert_allclose(p1[n], p0[n], rtol=1e-4, atol=1e-6, err_msg=n)


@pytest.mark.gpu
def test_prefetcher_edge_cases():
    """No batches at all, a loop left early, a sampler that raises: the iterator ends / closes / re-raises on the
    caller's thread and its workers exit."""
    import scipy.sparse as sp
    from mrgcn_amd import synth
    from mrgcn_amd.data import batch as mb
    g = synth.make_graph("aifb", seed=1, scale=0.1)
    N, R = g.num_nodes, g.num_relations
    dcsr = mb.DeviceCSR(sp.csr_matrix((g.vals, (g.rows, g.cols)), shape=(N, R * N)))
    pf = mb.BatchPrefetcher(dcsr, iter(()), 2, value_mode="norm_f32")
    assert list(pf) == []
    pf.close()
    rng = np.random.default_rng(0)
    pf = mb.BatchPrefetcher(dcsr, (np.sort(rng.choice(N, 8, replace=False)) for _ in range(50)), 2, value_mode="norm_f32")
    for k, ab in enumerate(pf):
        assert len(ab.row) == 2 and int(ab.node_index.numel()) == 8
        if k == 2:
            break
    pf.close()
    assert not any(t.is_alive() for t in pf._threads)

    def bad():
        yield np.arange(4)
        raise RuntimeError("sampler failed")
    pf = mb.BatchPrefetcher(dcsr, bad(), 1, value_mode="norm_f32", workers=1)
    with pytest.raises(RuntimeError, match="sampler failed"):
        for _ in pf:
            pass
    pf.close()


@pytest.mark.gpu
def test_frontier_kernels_edge_cases_through_the_c_abi():
    """mrgcn_frontier_count / _emit against the host functions on a graph with isolated rows, a hub row longer
    than several waves, duplicated sample rows and an empty sample; float32 and int8 (truncating) values."""
    import scipy.sparse as sp
    from mrgcn_amd.data import batch as mb
    rng = np.random.default_rng(5)
    N, R = 300, 7
    rows = rng.integers(0, N, 2500)
    cols = rng.integers(0, R * N, 2500)
    rows = np.concatenate([rows, np.full(700, 17)])                      # hub row
    cols = np.concatenate([cols, rng.choice(R * N, 700, replace=False)])
    keep = ~np.isin(rows, [3, 4, 250])                                    # isolated rows
    rows, cols = rows[keep], cols[keep]
    vals = rng.uniform(-1.9, 1.9, rows.size).astype(np.float32)           # int8 cast truncates toward zero
    A = sp.csr_matrix((vals, (rows, cols)), shape=(N, R * N))
    A.sum_duplicates()
    dcsr = mb.DeviceCSR(A)
    for sample in ([17], [3, 4, 250], [5, 17, 5, 3, 299, 0], list(range(N)), []):
        smp = np.asarray(sample, dtype=np.int64)
        for mode, dt in (("ref_int8", np.int8), ("norm_f32", np.float32)):
            row, col, val, col_sl, nb = dcsr.frontier(torch.from_numpy(smp), mode)
            sub = A[smp] if smp.size else A[:0]
            r_h, c_h = sub.nonzero()
            # scipy's nonzero() drops explicit zeros; the kernels keep every stored entry (as the reference's
            # A[sample] COO does): compare on the stored structure
            sub = sub.tocsr()
            r_h = np.repeat(np.arange(sub.shape[0]), np.diff(sub.indptr))
            c_h = sub.indices
            assert np.array_equal(row.cpu().numpy(), r_h) and np.array_equal(col.cpu().numpy(), c_h)
            want_v = sub.data.astype(np.int32).astype(np.int8) if dt is np.int8 else sub.data
            assert np.array_equal(val.cpu().numpy(), want_v)
            nb_h = mb.getNeighboursSparse(A, smp) if smp.size else np.zeros(0, dtype=np.int64)
            assert np.array_equal(nb.cpu().numpy(), nb_h)
            if nb_h.size:
                pos = np.searchsorted(nb_h, c_h % N)
                assert np.array_equal(col_sl.cpu().numpy(), (c_h // N) * nb_h.size + pos)
            else:
                assert col_sl.numel() == 0


@pytest.mark.gpu
def test_masked_batch_with_node_dropout_and_three_layers_matches_the_slice_path():
    """Three layers (the hidden one takes both the feature term's compact input gradient and the ReLU mask hand-over)
    and the unfused activation path (ReLU applied outside the product when dropout is on, p = 0 draws kept equal by
    seeding): masked pass = slice path, logits and gradients."""
    from mrgcn_amd.data import batch as mb
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.plan import GraphPlan
    import scipy.sparse as sp
    from mrgcn_amd import synth
    g = synth.make_graph("aifb", seed=2, scale=0.2)
    N, R = g.num_nodes, g.num_relations
    A = sp.csr_matrix((g.vals, (g.rows, g.cols)), shape=(N, R * N))
    rng = np.random.default_rng(7)
    K, C = 5, 3
    X = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).cuda()
    idx = rng.choice(N, 17, replace=False)        # unsorted on purpose
    torch.manual_seed(0)
    model = RGCN([(K, 8, "mrgcn", torch.nn.ReLU()), (8, 6, "mrgcn", torch.nn.ReLU()), (6, C, "mrgcn", None)], R, N, 2, 0.0,
                 False, True, False).cuda()
    plan = GraphPlan.from_csr(A, N, R, value_mode="norm_f32")
    outs = []
    for masked in (False, True):
        ab = (mb.A_BatchMasked(plan, idx, 3) if masked
              else mb.A_BatchDevice(mb.DeviceCSR(A), idx, 3, value_mode="norm_f32"))
        model.zero_grad()
        Xb = X[ab.neighbours[-1]].clone().requires_grad_(True)
        logits = model(Xb, ab)
        (logits * torch.arange(1, C + 1, device="cuda")).sum().backward()
        outs.append((logits.detach().cpu().numpy(), Xb.grad.cpu().numpy(),
                     {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters()}))
    np.testing.assert_allclose(outs[1][0], outs[0][0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(outs[1][1], outs[0][1], rtol=1e-3, atol=1e-5)
    for n in outs[0][2]:
        np.testing.assert_allclose(outs[1][2][n], outs[0][2][n], rtol=1e-3, atol=1e-5, err_msg=n)


@pytest.mark.gpu
@pytest.mark.parametrize("bias", [False, True])
def test_one_batch_of_all_labelled_nodes_trains_like_the_full_batch(bias):
    """Mini-batch mode takes ALL neighbours (data/batch.py:185-263, no sampling): one batch that holds every labelled
    node computes the labels' receptive field only.  With the adjacency's values kept for the feature term
    (`full_batch_values`: the reference's own slices drop them, batch.py:258-270) three training steps leave the same
    losses and the same parameters as the full-batch steps (node_classification.py:166-193) — eager and replayed from
    a hipGraph (bench.py: extra.labelled_nodes_as_one_batch); without it (the reference's mini-batch arithmetic) they
    do not."""
    from mrgcn_amd.data import batch as mb
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.plan import GraphPlan
    from mrgcn_amd.train import ClipAdam, GraphedTrainStep, train_step
    import scipy.sparse as sp
    from mrgcn_amd import synth
    g = synth.make_graph("am", seed=2, scale=0.03)
    N, R = g.num_nodes, g.num_relations
    idx_np, y_np = synth.make_labels("am", N, 2, 0.03)
    rng = np.random.default_rng(0)
    K, H, C, B = 20, 10, int(y_np.max()) + 1, 5
    X = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).cuda()
    A_csr = sp.csr_matrix((g.vals, (g.rows, g.cols)), shape=(N, R * N))
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals), (N, R * N)).cuda()
    plan = GraphPlan.from_csr(A_csr, N, R, value_mode="norm_f32")
    mods = [(K, H, "mrgcn", torch.nn.ReLU()), (H, C, "mrgcn", None)]
    idx, tgt = torch.from_numpy(idx_np).cuda(), torch.from_numpy(y_np).cuda()
    order = np.argsort(idx_np, kind="stable")
    nodes, ys = idx_np[order], torch.from_numpy(y_np[order]).cuda()
    rows = torch.arange(len(nodes), device="cuda")

    def run(kind):
        torch.manual_seed(0)
        model = RGCN(mods, R, N, B, 0.0, False, bias, False).cuda()
        opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0, capturable=(kind == "graph"))
        losses = []
        if kind == "full":
            for _ in range(3):
                losses.append(float(train_step(model, lambda: model(X, A), idx, tgt, opt)))
        else:
            am = mb.A_BatchMasked(plan, nodes, 2, full_batch_values=(kind != "reference-slices"))
            if kind == "graph":
                # (the constructor runs ONE eager warm-up step before the capture: it is step 1 of the three)
                step = GraphedTrainStep(model, lambda: model(X, am), rows, ys, opt, warmup=1)
                losses = [float("nan"), float(step()), float(step())]
            else:
                for _ in range(3):
                    losses.append(float(train_step(model, lambda: model(X, am), rows, ys, opt)))
            torch.cuda.synchronize()
            am.close()
        return losses, {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}

    l0, p0 = run("full")
    for kind in ("eager", "graph"):
        l1, p1 = run(kind)
        np.testing.assert_allclose(l1[1:], l0[1:], rtol=1e-5, atol=1e-6, err_msg=kind)
        for n in p0:
            np.testing.assert_allclose(p1[n], p0[n], rtol=2e-4, atol=2e-6, err_msg=f"{kind}: {n}")
    l2, _ = run("reference-slices")
    assert abs(l2[1] - l0[1]) > 1e-4   # (the synthetic graph's values are 1 / in-degree per relation: not all ones)
