"""The reference's own training loop, unchanged (mrgcn/tasks/node_classification.py:35-37, :190-193):

    optimizer = optim.Adam(groups, lr=..., weight_decay=...); criterion = nn.CrossEntropyLoss()
    ...
    optimizer.zero_grad(); batch_loss.backward(); nn.utils.clip_grad_norm_(model.parameters(), 1.0); optimizer.step()

driven over this package's models (a) with torch's own `optim.Adam` / `clip_grad_norm_` (dense gradients) and
(b) with the drop-ins of mrgcn_amd.optim (row-sparse node-table gradient), against the golden vectors the
reference's loop produced (tests/golden/make_goldens.py: losses, gradient norm, parameters after 1 and n epochs)."""
import copy

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _adjacency(c, name):
    g, A = util.load_graph(util.graph_of_case(name))
    return util.coo_tensor(A, str(c["value_mode"]), "cuda")


def _groups(model):
    """optimizer_params-style groups (tasks/utils.py:8-45): one default group; gates would get their own."""
    return [{"params": [p for _, p in model.named_parameters() if p.requires_grad]}]


def _reference_loop(model, X, A, idx, tgt, Adam, clip, steps, on_step=None):
    optimizer = Adam(_groups(model), lr=0.01, weight_decay=0.0)
    criterion = torch.nn.CrossEntropyLoss()
    out = []
    for step in range(1, steps + 1):
        Y_hat = model(X, A)
        batch_loss = criterion(Y_hat[idx], tgt)           # categorical_crossentropy, :439-444
        optimizer.zero_grad()
        batch_loss.backward()
        norm = clip(model.parameters(), 1.0)
        optimizer.step()
        out.append((float(batch_loss), float(norm)))
        if on_step:
            on_step(step, model)
    return optimizer, out


@pytest.mark.parametrize("flavour", ["torch", "mrgcn_amd"])
@pytest.mark.parametrize("name", util.rgcn_cases())
def test_reference_loop_vs_reference_goldens(name, flavour):
    from mrgcn_amd import optim as O
    c = util.load_case(name)
    model, dims = util.build_rgcn_from_case(c, "cuda")
    util.load_state_from_case(model, c)
    model = model.cuda()
    A = _adjacency(c, name)
    X = None if bool(c["meta.featureless"]) else torch.from_numpy(c["X"]).cuda()
    idx = torch.from_numpy(c["labels_idx"]).cuda()
    tgt = torch.from_numpy(c["labels_y"]).cuda()
    n_adam = int(c["meta.n_adam"])
    Adam, clip = ((torch.optim.Adam, torch.nn.utils.clip_grad_norm_) if flavour == "torch"
                  else (O.RowSparseAdam, O.clip_grad_norm_))
    node_major = [m.weight_I for m in model.layers.values() if m.weight_I_node_major]

    def check(step, model):
        for w in node_major:  # torch's loop needs the dense gradient; the drop-ins may leave it in row-sparse form
            assert w.grad is not None or flavour == "mrgcn_amd"
        if step in (1, n_adam):
            sd = model.state_dict()
            for k in c.files:
                if k.startswith(f"adam{step}."):
                    diff = np.abs(sd[k[len(f"adam{step}."):]].cpu().numpy() - c[k])
                    assert (diff > 2e-5).mean() < 2e-3, (k, float((diff > 2e-5).mean()))
                    assert diff.max() <= 0.021 * step, k

    opt, log = _reference_loop(model, X, A, idx, tgt, Adam, clip, n_adam, check)
    for step, (loss, norm) in enumerate(log, 1):
        np.testing.assert_allclose(loss, float(c[f"loss_step{step}"]), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(log[0][1], float(c["grad_norm"]), rtol=1e-4)
    # the optimizer checkpoint (run.py:232-235) in the reference's layout: moments shaped like state_dict()'s tensors
    osd = O.reference_state_dict(opt)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    sd = model.state_dict()
    for k, st in osd["state"].items():
        assert tuple(st["exp_avg"].shape) == tuple(sd[names[k]].shape), names[k]
        assert int(st["step"]) == n_adam


def _problem(N=5000, R=3, labelled=6, seed=3):
    rng = np.random.default_rng(seed)
    rows = np.concatenate([rng.integers(0, N, N), np.arange(N)])
    cols = np.concatenate([rng.integers(0, (R - 1) * N, N), (R - 1) * N + np.arange(N)])
    key = np.unique(rows.astype(np.int64) * (R * N) + cols)
    rows, cols = key // (R * N), key % (R * N)
    vals = rng.uniform(0.2, 1.0, len(rows)).astype(np.float32)
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
    idx = torch.from_numpy(rng.choice(N, labelled, replace=False).astype(np.int64)).cuda()
    tgt = torch.from_numpy(rng.integers(0, 4, labelled).astype(np.int64)).cuda()
    X = torch.randn((N, 6), device="cuda", generator=torch.Generator("cuda").manual_seed(seed))
    return A, X, idx, tgt, N, R


def _model(N, R, seed=0):
    from mrgcn_amd.models.rgcn import RGCN
    torch.manual_seed(seed)
    return RGCN([(6, 10, "mrgcn", torch.nn.ReLU()), (10, 4, "mrgcn", None)], R, N, 5, 0.0, False, True, False).cuda()


def test_fast_reference_loop_equals_the_dense_one_and_checkpoints_interchange():
    """mrgcn_amd.optim (row-sparse) against torch.optim.Adam + torch's clip (dense) on a graph where most of the node
    table never gets gradient: same losses, norms and parameters; then both optimizers' checkpoints — taken in the
    reference's layout — are loaded into the OTHER kind of optimizer over a deep copy of the model, and three more
    epochs agree again (moments of the node table carried across in (B*N, out) form)."""
    from mrgcn_amd import optim as O
    A, X, idx, tgt, N, R = _problem()
    ma, mb = _model(N, R), _model(N, R)
    oa, la = _reference_loop(ma, X, A, idx, tgt, torch.optim.Adam, torch.nn.utils.clip_grad_norm_, 3)
    ob, lb = _reference_loop(mb, X, A, idx, tgt, O.RowSparseAdam, O.clip_grad_norm_, 3)
    np.testing.assert_allclose(np.array(lb), np.array(la), rtol=1e-5, atol=1e-7)
    for (k, va), vb in zip(ma.state_dict().items(), mb.state_dict().values()):
        torch.testing.assert_close(vb, va, rtol=1e-6, atol=1e-7, msg=k)
    ent = mb.layers["layer_0"].weight_I._mrgcn_rows
    assert 0.0 < float(ent["ever"].float().mean()) < 0.5
    sa, sb = O.reference_state_dict(oa), O.reference_state_dict(ob)
    wi = [n for n, _ in ma.named_parameters()].index("layers.layer_0.weight_I")
    assert tuple(sa["state"][wi]["exp_avg"].shape) == tuple(ma.state_dict()["layers.layer_0.weight_I"].shape)
    torch.testing.assert_close(sb["state"][wi]["exp_avg"], sa["state"][wi]["exp_avg"], rtol=1e-5, atol=1e-8)
    # cross-load: the dense run continues on the fast path and the other way round (deep copies: the tags that mark
    # the node-major table must survive copy.deepcopy)
    mc, md = copy.deepcopy(ma), copy.deepcopy(mb)
    oc, od = O.RowSparseAdam(_groups(mc), lr=0.01), torch.optim.Adam(_groups(md), lr=0.01)
    O.load_reference_state_dict(oc, sa)
    O.load_reference_state_dict(od, sb)
    crit = torch.nn.CrossEntropyLoss()
    for _ in range(3):
        for m, o, clip in ((mc, oc, O.clip_grad_norm_), (md, od, torch.nn.utils.clip_grad_norm_)):
            loss = crit(m(X, A)[idx], tgt)
            o.zero_grad()
            loss.backward()
            clip(m.parameters(), 1.0)
            o.step()
    assert mc.layers["layer_0"].weight_I.grad is None and md.layers["layer_0"].weight_I.grad is not None
    for (k, vc), vd in zip(mc.state_dict().items(), md.state_dict().values()):
        torch.testing.assert_close(vc, vd, rtol=1e-4, atol=1e-6, msg=k)   # (two runs that met at 1e-6 three epochs ago)
    assert int(O.reference_state_dict(oc)["state"][wi]["step"]) == 6


def test_fast_loop_with_a_weight_regulariser_and_weight_decay_falls_back_to_dense_gradients():
    """An L2 term on the parameters (node_classification.py:180-188) puts a dense gradient on the node table next to
    the layer's row-sparse one: the drop-ins merge the two and take one dense step — equal to torch's loop.
    A group with weight_decay != 0 never goes row-sparse (a decayed parameter moves without gradient)."""
    from mrgcn_amd import optim as O
    A, X, idx, tgt, N, R = _problem(N=3000)
    res = []
    for Adam, clip, wd in ((torch.optim.Adam, torch.nn.utils.clip_grad_norm_, 0.0), (O.RowSparseAdam, O.clip_grad_norm_, 0.0),
                           (torch.optim.Adam, torch.nn.utils.clip_grad_norm_, 0.01), (O.RowSparseAdam, O.clip_grad_norm_, 0.01)):
        m = _model(N, R)
        opt = Adam(_groups(m), lr=0.01, weight_decay=wd)
        crit = torch.nn.CrossEntropyLoss()
        for _ in range(3):
            loss = crit(m(X, A)[idx], tgt)
            l2 = torch.zeros((), device="cuda")
            for name, p in m.named_parameters():
                if "weight" in name:
                    l2 = l2 + torch.sum(p ** 2)
            loss = loss + 1e-3 * l2
            opt.zero_grad()
            loss.backward()
            clip(m.parameters(), 1.0)
            opt.step()
        res.append({k: v.clone() for k, v in m.state_dict().items()})
    for a, b in ((0, 1), (2, 3)):
        for k in res[a]:
            torch.testing.assert_close(res[b][k], res[a][k], rtol=1e-5, atol=1e-7, msg=k)


def test_plain_adam_keeps_dense_gradients_so_that_torchs_clip_sees_the_node_table():
    """`mrgcn_amd.optim.Adam` constructed directly (row_sparse=False by default) next to TORCH's clip_grad_norm_: the
    node table's gradient stays in `.grad`, its norm is part of the total and it is scaled like every other gradient —
    the loop equals torch's own.  (With the row-sparse form torch's clip would skip the node table: that form is only
    announced by RowSparseAdam, which install_as_mrgcn(patch_optimizer=True) binds together with this package's clip.)"""
    from mrgcn_amd import optim as O
    A, X, idx, tgt, N, R = _problem(N=3000)
    ma, mb = _model(N, R), _model(N, R)
    oa, la = _reference_loop(ma, X, A, idx, tgt, torch.optim.Adam, torch.nn.utils.clip_grad_norm_, 3)
    ob, lb = _reference_loop(mb, X, A, idx, tgt, O.Adam, torch.nn.utils.clip_grad_norm_, 3)
    assert mb.layers["layer_0"].weight_I.grad is not None
    np.testing.assert_allclose(np.array(lb), np.array(la), rtol=1e-5, atol=1e-7)
    for (k, va), vb in zip(ma.state_dict().items(), mb.state_dict().values()):
        torch.testing.assert_close(vb, va, rtol=1e-5, atol=1e-7, msg=k)


def test_clip_with_a_gradient_on_another_device_goes_through_torch():
    """a parameter whose gradient lives on the CPU (the reference spreads modules over model.devices) next to a
    row-sparse node table: the fast clip must not read it as a device pointer — the entries are densified and torch's
    clip runs"""
    from mrgcn_amd import optim as O
    A, X, idx, tgt, N, R = _problem(N=2000)
    m = _model(N, R)
    extra = torch.nn.Parameter(torch.ones(3))          # CPU parameter with a CPU gradient
    opt = O.RowSparseAdam(_groups(m), lr=0.01)
    loss = torch.nn.CrossEntropyLoss()(m(X, A)[idx], tgt) + (extra.sum() * 0.5).cuda()
    opt.zero_grad()
    loss.backward()
    assert m.layers["layer_0"].weight_I.grad is None and extra.grad is not None
    params = list(m.parameters()) + [extra]
    # torch's own clip refuses tensors on several devices unless foreach=False; that is the call the fallback makes
    norm = O.clip_grad_norm_(params, 1e9, foreach=False)   # (max_norm huge: gradients stay as they are)
    assert m.layers["layer_0"].weight_I.grad is not None   # densified for torch's clip
    want = torch.sqrt(sum((p.grad.double().cpu() ** 2).sum() for p in params if p.grad is not None))
    assert abs(float(want) - float(norm)) < 1e-4 * float(norm) + 1e-6
    opt.step()


@pytest.mark.parametrize("row_sparse", [False, True])
def test_optimizer_checkpoints_interchange_with_the_reference_under_its_unpatched_lines(row_sparse):
    """The reference's checkpoint lines as they are (node_classification.py:35-37, :73-80; run.py:230-236) inside a
    module namespace as `install_as_mrgcn()` leaves the task modules (default: torch's Adam with the layout translated;
    `patch_optimizer=True`: RowSparseAdam + this package's clip), against tests/golden/optim_checkpoint.npz — the
    REFERENCE's model / optimizer state after two and three of its own epochs:
      (a) two epochs here from the reference's initial parameters, then `optimizer.state_dict()`: every entry has the
          reference's shape (`weight_I` moments `(B*N, out)`) and the reference's values — a checkpoint written here
          loads in the reference;
      (b) a fresh model + optimizer, `load_state_dict` of the reference's checkpoint after two epochs, one epoch of the
          reference's loop: the reference's parameters, moments and loss of its third epoch."""
    import os
    import types

    import mrgcn_amd
    from mrgcn_amd.models.rgcn import RGCN
    c = np.load(os.path.join(util.GOLDEN, "optim_checkpoint.npz"))
    g, A_csr = util.load_graph("graph_small")
    A = util.coo_tensor(A_csr, "norm_f32", "cuda")
    N, R, B = int(c["meta.num_nodes"]), int(c["meta.R"]), int(c["meta.num_bases"])
    task = types.ModuleType("fake_task_module")
    task.optim, task.nn = torch.optim, torch.nn               # `import torch.optim as optim`, `import torch.nn as nn`
    mrgcn_amd.patch_task_optimizer(task, row_sparse=row_sparse)
    optim, nn = task.optim, task.nn
    X = torch.from_numpy(c["X"]).cuda()
    idx, tgt = torch.from_numpy(c["labels_idx"]).cuda(), torch.from_numpy(c["labels_y"]).cuda()
    names = [str(n) for n in c["param_names"]]

    def build():
        model = RGCN([(6, 8, "mrgcn", torch.nn.ReLU()), (8, 4, "mrgcn", None)], R, N, B, 0.0, False, True, False).cuda()
        assert [n for n, _ in model.named_parameters()] == names
        optimizer = optim.Adam([{"params": list(model.parameters())}], lr=0.01, weight_decay=0.0)   # :35-37
        return model, optimizer, nn.CrossEntropyLoss()

    def epoch(model, optimizer, criterion):
        loss = criterion(model(X, A)[idx], tgt)
        optimizer.zero_grad()
        loss.backward()                                        # :190-193
        nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        return float(loss)

    def check_against(model, optimizer, k, tight):
        sd, osd = model.state_dict(), optimizer.state_dict()   # run.py:232-233
        for i, n in enumerate(names):
            diff = np.abs(sd[n].cpu().numpy() - c[f"state{k}.{n}"])
            assert diff.max() <= (2e-5 if tight else 0.021 * k), (n, float(diff.max()))
            assert (diff > 2e-5).mean() < 2e-2, n
            st = osd["state"][i]
            assert tuple(st["exp_avg"].shape) == c[f"optim{k}.{i}.exp_avg"].shape == tuple(sd[n].shape), n
            np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), c[f"optim{k}.{i}.exp_avg"], rtol=2e-3,
                                       atol=2e-5 * float(np.abs(c[f"optim{k}.{i}.exp_avg"]).max()), err_msg=n)
            np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), c[f"optim{k}.{i}.exp_avg_sq"], rtol=4e-3,
                                       atol=4e-5 * float(np.abs(c[f"optim{k}.{i}.exp_avg_sq"]).max()), err_msg=n)
            assert float(st["step"]) == float(k)

    # (a) written here, readable there
    model, optimizer, criterion = build()
    model.load_state_dict({n: torch.from_numpy(c["init." + n]) for n in names})
    losses = [epoch(model, optimizer, criterion) for _ in range(2)]
    np.testing.assert_allclose(losses, [float(c["loss_step1"]), float(c["loss_step2"])], rtol=2e-4, atol=2e-5)
    check_against(model, optimizer, 2, tight=False)
    # (b) written there, resumed here
    model, optimizer, criterion = build()
    checkpoint = {"model_state_dict": {n: torch.from_numpy(c["state2." + n]) for n in names},
                  "optimizer_state_dict": {
                      "state": {i: {"step": torch.tensor(float(c[f"optim2.{i}.step"])),
                                    "exp_avg": torch.from_numpy(c[f"optim2.{i}.exp_avg"]),
                                    "exp_avg_sq": torch.from_numpy(c[f"optim2.{i}.exp_avg_sq"])} for i in range(len(names))},
                      "param_groups": torch.optim.Adam([torch.nn.Parameter(torch.zeros(1)) for _ in names],
                                                       lr=0.01).state_dict()["param_groups"]}}
    model.load_state_dict(checkpoint["model_state_dict"])              # :78
    optimizer.load_state_dict(checkpoint["optimizer_state_dict"])      # :79
    loss3 = epoch(model, optimizer, criterion)
    np.testing.assert_allclose(loss3, float(c["loss_step3"]), rtol=2e-4, atol=2e-5)
    check_against(model, optimizer, 3, tight=True)
    from mrgcn_amd import optim as fast
    assert type(optimizer) is (fast.RowSparseAdam if row_sparse else fast.ReferenceLayoutAdam)
