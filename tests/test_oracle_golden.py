"""Pins the CPU oracle (oracle/rgcn_oracle.py) against the golden vectors captured
from the reference itself (tests/golden/make_goldens.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import rgcn_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
RGCN_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "rgcn_*.npz")))


def load_graph(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    A = sp.csr_matrix((g["csr_data"], g["csr_indices"], g["csr_indptr"]), shape=tuple(g["shape"]))
    return g, A


@pytest.mark.parametrize("gname", ["graph_small", "graph_smoke"])
def test_adjacency_layout_bit_exact(gname):
    """a-1: block order, column = r*N + j, values 1/deg — index sets bit-exact."""
    g, A_ref = load_graph(gname)
    A = O.build_stacked_adjacency(g["triples"], int(g["num_nodes"]), int(g["num_pred"]))
    assert A.shape == A_ref.shape and A.nnz == A_ref.nnz
    np.testing.assert_array_equal(A.indptr, A_ref.indptr)
    # hstack output is not guaranteed column-sorted within a row: compare canonical forms
    a, b = A.copy(), A_ref.copy()
    a.sort_indices(); b.sort_indices()
    np.testing.assert_array_equal(a.indices, b.indices)
    np.testing.assert_array_equal(a.data, b.data)  # float32 bit-exact


@pytest.mark.parametrize("gname", ["graph_small", "graph_smoke"])
def test_coo_int8_bit_exact(gname):
    """a-2: COO indices (order included) and the int8 truncation."""
    g, A_ref = load_graph(gname)
    idx, val = O.csr_to_coo(A_ref, "ref_int8")
    np.testing.assert_array_equal(idx, g["coo_indices"])
    np.testing.assert_array_equal(val, g["coo_values_i8"])
    assert val.dtype == np.int8
    # only exact ones survive the truncation
    assert set(np.unique(val)) <= {0, 1}


def _case(name):
    c = np.load(os.path.join(GOLDEN, name + ".npz"))
    gname = "graph_smoke" if "_smoke_" in name else "graph_small"
    g, A_csr = load_graph(gname)
    idx, val = O.csr_to_coo(A_csr, str(c["value_mode"]))
    A = O.coo_to_csr(idx, val, A_csr.shape)
    return c, A


@pytest.mark.parametrize("name", RGCN_CASES)
def test_rgcn_forward_backward_adam(name):
    c, A = _case(name)
    N, R, B = int(c["meta.num_nodes"]), int(c["meta.R"]), int(c["meta.num_bases"])
    bias, fl = bool(c["meta.bias"]), bool(c["meta.featureless"])
    lp = bool(c["meta.link_prediction"])
    dims = [tuple(d) for d in c["dims"]]
    state = {k[len("init."):]: c[k] for k in c.files if k.startswith("init.")}
    X = None if fl else c["X"]
    n_adam = int(c["meta.n_adam"])
    recs = O.train_steps(dims, R, N, B, bias, fl, state, X, A, c["labels_idx"], c["labels_y"],
                         n_adam, relu_last=lp)
    r0 = recs[0]
    # per-layer activations
    for li, (_, pre, _, act) in enumerate(r0["tape"]):
        np.testing.assert_allclose(pre, c[f"act.pre_{li}"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(r0["logits"], c["logits"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(r0["loss"], c["loss"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r0["grad_norm"], c["grad_norm"], rtol=1e-5)
    for k in c.files:
        if k.startswith("grad.") and k != "grad.X":
            key = k[len("grad."):]
            if key == "relations":
                continue
            np.testing.assert_allclose(r0["grads"][key], c[k], rtol=1e-4, atol=1e-6, err_msg=k)
    if not fl:
        np.testing.assert_allclose(r0["dX"], c["grad.X"], rtol=1e-4, atol=1e-6)
    for step in (1, n_adam):
        st = recs[step - 1]["state"]
        for k in c.files:
            if k.startswith(f"adam{step}."):
                key = k[len(f"adam{step}."):]
                if key == "relations":
                    continue
                # Adam's first steps are ~lr*sign(g): elements whose fp32 gradient is at
                # rounding-noise level may flip sign; compare with an lr-sized tolerance
                # on <0.1 % of elements and tightly elsewhere
                diff = np.abs(st[key] - c[k])
                assert (diff > 1e-5).mean() < 1e-3, k
                assert diff.max() <= 0.021 * step, k
    for step in range(1, n_adam + 1):
        np.testing.assert_allclose(recs[step - 1]["loss"], c[f"loss_step{step}"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("name", RGCN_CASES)
def test_aten_literal_port_matches_reference(name):
    """The timed CPU baseline (oracle/aten_literal.py) executes the reference's ATen op
    sequence: on the reference's own inputs it must reproduce logits, loss, gradients and
    the post-Adam parameters to float32 round-off (same ops, same order)."""
    import torch
    from oracle import aten_literal as AL
    c = np.load(os.path.join(GOLDEN, name + ".npz"))
    gname = "graph_smoke" if "_smoke_" in name else "graph_small"
    g, A_csr = load_graph(gname)
    idx, val = O.csr_to_coo(A_csr, str(c["value_mode"]))
    N, R, B = int(c["meta.num_nodes"]), int(c["meta.R"]), int(c["meta.num_bases"])
    fl, lp = bool(c["meta.featureless"]), bool(c["meta.link_prediction"])
    dims = [tuple(d) for d in c["dims"]]
    A = AL.coo_tensor(idx[0], idx[1], val, (N, R * N))
    p = {k[len("init."):]: torch.from_numpy(np.array(c[k])).requires_grad_(True)
         for k in c.files if k.startswith("init.") and k != "init.relations"}
    X = None if fl else torch.from_numpy(c["X"])
    ep = AL.Epoch(p, len(dims), R, N, B, fl, relu_last=lp)
    it, tt = torch.from_numpy(c["labels_idx"]), torch.from_numpy(c["labels_y"])
    n_adam = int(c["meta.n_adam"])
    for step in range(1, n_adam + 1):
        Y_hat, loss, norm = ep.step(X, A, it, tt)
        if step == 1:
            np.testing.assert_allclose(Y_hat.detach().numpy(), c["logits"], rtol=0, atol=1e-6)
            np.testing.assert_allclose(loss.item(), c["loss"], rtol=1e-6)
            np.testing.assert_allclose(float(norm), c["grad_norm"], rtol=1e-5)
        if step in (1, n_adam):
            for k, v in p.items():
                diff = np.abs(v.detach().numpy() - c[f"adam{step}.{k}"])
                assert (diff > 1e-6).mean() < 1e-3 and diff.max() <= 0.021 * step, k


# ---- link-prediction decoder ------------------------------------------------
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_lp_oracle_against_reference_goldens(tag):
    from oracle import lp_oracle as lo
    g = np.load(os.path.join(GOLDEN, "lp_decoder.npz"))
    E, Rel, facts = g[f"{tag}.E"], g[f"{tag}.Rel"], g[f"{tag}.facts"]
    np.testing.assert_allclose(lo.score_distmult(facts, E, Rel), g[f"{tag}.scores"], rtol=1e-5, atol=1e-5)
    assert abs(lo.bce_with_logits(g[f"{tag}.scores"], g[f"{tag}.y"]) - float(g[f"{tag}.loss"])) < 1e-6
    dE, dR = lo.distmult_bce_grads(facts, E, Rel, g[f"{tag}.y"])
    np.testing.assert_allclose(dE, g[f"{tag}.dE"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(dR, g[f"{tag}.dRel"], rtol=1e-4, atol=1e-6)
    # ranks are integers: exact
    assert np.array_equal(lo.compute_ranks(facts, E, Rel, False), g[f"{tag}.ranks_raw"])
    assert np.array_equal(lo.compute_ranks(facts, E, Rel, True), g[f"{tag}.ranks_flt"])


def test_lp_filter_lists_host_matches_oracle():
    from oracle import lp_oracle as lo
    from mrgcn_amd.tasks.link_prediction import filter_lists
    rng = np.random.default_rng(5)
    for n, N, P in [(0, 5, 2), (1, 5, 2), (400, 12, 3), (1000, 200, 7)]:
        facts = np.stack([rng.integers(0, N, n), rng.integers(0, P, n), rng.integers(0, N, n)], 1).astype(np.int64)
        for a, b in zip(filter_lists(facts), lo.filter_lists(facts)):
            assert np.array_equal(a, b)


def test_lp_negative_sampling_shape_and_labels():
    from mrgcn_amd.tasks.link_prediction import sample_negatives
    rng = np.random.RandomState(3)
    facts = np.stack([rng.randint(0, 50, 103), rng.randint(0, 4, 103), rng.randint(0, 50, 103)], 1)
    neg, Y = sample_negatives(facts, np.random.RandomState(0))
    assert neg.shape == (20, 3) and Y.shape == (123,) and Y[:103].all() and not Y[103:].any()
    nodes = np.union1d(facts[:, 0], facts[:, 2])
    assert np.isin(neg[:, 0], nodes).all() and np.isin(neg[:, 2], nodes).all()
    # relation untouched, exactly one side re-drawn per corrupted triple
    pos = {tuple(f[1:]) for f in facts.tolist()} | {tuple(f[:2]) for f in facts.tolist()}
    assert all(tuple(t[1:]) in pos or tuple(t[:2]) in pos for t in neg.tolist())


@pytest.mark.parametrize("name", RGCN_CASES)
def test_receptive_field_forward_matches_reference_logits(name):
    """`rgcn_forward_at_rows` (the full-size AM checker of tests/test_gpu_fullsize.py) against the
    reference's own logits on every golden case, at a sample of rows and at all of them."""
    c, A = _case(name)
    N, R, B = int(c["meta.num_nodes"]), int(c["meta.R"]), int(c["meta.num_bases"])
    bias, fl = bool(c["meta.bias"]), bool(c["meta.featureless"])
    lp = bool(c["meta.link_prediction"])
    dims = [tuple(d) for d in c["dims"]]
    state = {k[len("init."):]: c[k] for k in c.files if k.startswith("init.")}
    cfgs = O.rgcn_cfgs(dims, R, N, B, bias, fl)
    params = O.split_params(state, len(cfgs))
    X = None if fl else c["X"]
    rng = np.random.default_rng(3)
    for rows in (np.sort(rng.choice(N, min(N, 17), replace=False)), np.arange(N)):
        got = O.rgcn_forward_at_rows(cfgs, params, X, A, rows, relu_last=lp, chunk=7)
        np.testing.assert_allclose(got, c["logits"][rows], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("name", [n for n in RGCN_CASES if "_b0" not in n and "_lp_" not in n])
def test_comp_gradient_at_rows_matches_reference_gradient(name):
    """`input_term_comp_grad_at_rows` (the FB15k-237 / AM full-shape gradient checker) against the gradient the
    reference's autograd produced for `weight_I_comp` on every golden case with bases: the cross-entropy gradient at
    the labelled rows is pushed back to layer 0's pre-activation output with the oracle's own backward, then handed
    to the helper with only the rows that carry any."""
    c, A = _case(name)
    A = A.astype(np.float64)
    N, R, B = int(c["meta.num_nodes"]), int(c["meta.R"]), int(c["meta.num_bases"])
    bias, fl = bool(c["meta.bias"]), bool(c["meta.featureless"])
    dims = [tuple(d) for d in c["dims"]]
    state = {k[len("init."):]: c[k] for k in c.files if k.startswith("init.")}
    cfgs = O.rgcn_cfgs(dims, R, N, B, bias, fl)
    params = O.split_params(state, len(cfgs))
    X = None if fl else c["X"].astype(np.float64)
    logits, tape = O.rgcn_forward(cfgs, params, X, A)
    _, dH = O.cross_entropy(logits, c["labels_idx"], c["labels_y"])
    for li in range(len(cfgs) - 1, 0, -1):
        H_in, pre, cache, act = tape[li]
        _, dH = O.layer_backward(cfgs[li], params[li], H_in, A, dH * (pre > 0) if act else dH, cache)
    dpre0 = dH * (tape[0][1] > 0) if tape[0][3] else dH
    rows = np.flatnonzero(np.abs(dpre0).sum(1) > 0)
    got = O.input_term_comp_grad_at_rows(cfgs[0], params[0], A, rows, dpre0[rows])
    np.testing.assert_allclose(got, c["grad.layers.layer_0.weight_I_comp"], rtol=1e-4, atol=1e-7)


def _ref_blocks(arr, cfg, rows):
    """Rows of a reference-shaped weight_I array as the at-rows oracle hands them out: node-major blocks `[n, B, out]`
    of the nodes `rows` (bases) or the literal rows `r*N + j` (no bases)."""
    if cfg.B > 0:
        return np.transpose(arr.reshape(cfg.B, cfg.N, cfg.outdim)[:, rows, :], (1, 0, 2))
    return arr[rows]


@pytest.mark.parametrize("name", RGCN_CASES)
def test_receptive_field_train_step_matches_reference_gradients_and_adam(name):
    """`rgcn_train_step_at_rows` (the full-size gradient / clip / Adam checker of tests/test_gpu_step_oracle.py) against
    what the reference's own autograd, `clip_grad_norm_` and `torch.optim.Adam` produced on every golden case: loss,
    every gradient (`weight_I` at ALL node blocks), the clip norm, the parameters after the first step — and the
    second step from the full oracle's state after the first (moments handed in) against the full oracle."""
    c, A = _case(name)
    N, R, B = int(c["meta.num_nodes"]), int(c["meta.R"]), int(c["meta.num_bases"])
    bias, fl = bool(c["meta.bias"]), bool(c["meta.featureless"])
    dims = [tuple(d) for d in c["dims"]]
    state = {k[len("init."):]: c[k] for k in c.files if k.startswith("init.")}
    cfgs = O.rgcn_cfgs(dims, R, N, B, bias, fl)
    params = O.split_params(state, len(cfgs))
    X = None if fl else c["X"]
    idx, y = c["labels_idx"], c["labels_y"]
    nodes = np.arange(N)
    lp = bool(c["meta.link_prediction"])   # (the golden's loss is the cross-entropy over the ReLU'd embeddings)
    got = O.rgcn_train_step_at_rows(cfgs, params, X, A, idx, y, sample_nodes=nodes, chunk=11, relu_last=lp)
    np.testing.assert_allclose(got["loss"], c["loss"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(got["logits"], c["logits"][idx], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(got["grad_norm"], c["grad_norm"], rtol=1e-5)
    for li, cfg in enumerate(cfgs):
        pre = f"layers.layer_{li}."
        for k, v in got["grads"][li].items():
            np.testing.assert_allclose(v, c["grad." + pre + k], rtol=1e-4, atol=1e-6, err_msg=pre + k)
        w = got["wI"][li]
        if w is None:
            continue
        ref = c["grad." + pre + "weight_I"]
        dense = np.zeros_like(ref, dtype=np.float64)
        if cfg.B > 0:
            dense.reshape(cfg.B, cfg.N, cfg.outdim)[:, w["rows"], :] = np.transpose(w["grad"], (1, 0, 2))
            # the nodes the oracle calls live are exactly those with any gradient column
            dead = np.setdiff1d(nodes, w["live_nodes"])
            assert not np.abs(_ref_blocks(ref, cfg, dead)).any()
        else:
            dense[w["rows"]] = w["grad"]
        np.testing.assert_allclose(dense, ref, rtol=1e-4, atol=1e-6, err_msg=pre + "weight_I")
    # parameters after the reference's first step; an element may differ by a whole Adam step (lr * sign) only where
    # its float32 gradient is rounding noise
    for li, cfg in enumerate(cfgs):
        pre = f"layers.layer_{li}."
        for k, (pn, mn, vn) in got["new"][li].items():
            ref = c["adam1." + pre + k]
            gk = c["grad." + pre + k]
            if k == "weight_I":
                ref, gk = _ref_blocks(ref, cfg, got["wI"][li]["rows"]), _ref_blocks(gk, cfg, got["wI"][li]["rows"])
            diff = np.abs(pn - ref)
            assert diff.max() <= 0.0201, pre + k
            assert not (diff[np.abs(gk) * got["coef"] > 1e-6] > 2e-5).any(), pre + k
    # the second step, from the full oracle's state and moments after the first
    full = O.train_steps(dims, R, N, B, bias, fl, state, X, A, idx, y, 2, relu_last=lp)
    st1 = {k: v for k, v in full[0]["state"].items()}
    params1 = O.split_params(st1, len(cfgs))
    coef1 = min(1.0, 1.0 / (full[0]["grad_norm"] + 1e-6))
    moments = []
    for li, cfg in enumerate(cfgs):
        mom = {}
        for k in list(got["grads"][li]) + (["weight_I"] if got["wI"][li] is not None else []):
            g1 = full[0]["grads"][f"layers.layer_{li}.{k}"] * coef1
            m, v = 0.1 * g1, 0.001 * g1 * g1
            if k == "weight_I":
                rows = nodes if cfg.B > 0 else np.arange(R * N)
                m, v = _ref_blocks(m, cfg, rows), _ref_blocks(v, cfg, rows)
            mom[k] = (m, v)
        moments.append(mom)
    samp = nodes
    got2 = O.rgcn_train_step_at_rows(cfgs, params1, X, A, idx, y, sample_nodes=samp, moments=moments, t=2, chunk=64,
                                     relu_last=lp)
    np.testing.assert_allclose(got2["loss"], full[1]["loss"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got2["grad_norm"], full[1]["grad_norm"], rtol=1e-9)
    for li, cfg in enumerate(cfgs):
        for k, (pn, mn, vn) in got2["new"][li].items():
            ref = full[1]["state"][f"layers.layer_{li}.{k}"]
            if k == "weight_I":
                ref = _ref_blocks(ref, cfg, got2["wI"][li]["rows"])
            np.testing.assert_allclose(pn, ref, rtol=1e-9, atol=1e-12, err_msg=k)


def test_receptive_field_step_with_the_decoder_loss_equals_the_full_oracles():
    """`rgcn_train_step_at_rows(loss_fn=...)` with the DistMult + BCE decoder on top (the FB15k-237 full-shape step
    checker): against the composition of the two pinned full-size oracles — `rgcn_forward` / `rgcn_backward` (pinned to
    the reference's autograd above) and `lp_oracle.distmult_bce_grads` (pinned to lp_decoder.npz) — on the golden graph
    with a one-layer featureless encoder (link_prediction.py:266-275 over rgcn.py:69-89)."""
    from oracle import lp_oracle as lo
    c, A = _case("rgcn_small_lp_b2")
    N, R, B = int(c["meta.num_nodes"]), int(c["meta.R"]), int(c["meta.num_bases"])
    dims = [tuple(d) for d in c["dims"]]
    state = {k[len("init."):]: c[k] for k in c.files if k.startswith("init.")}
    Rel = state.pop("relations").astype(np.float64)
    cfgs = O.rgcn_cfgs(dims, R, N, B, False, True)
    params = O.split_params(state, 1)
    rng = np.random.default_rng(0)
    P = (R - 1) // 2
    facts = np.stack([rng.integers(0, N - 7, 60), rng.integers(0, P, 60), rng.integers(0, N - 7, 60)], 1)
    y = (rng.random(60) < 0.8).astype(np.float64)
    # full-size composition
    E, tape = O.rgcn_forward(cfgs, params, None, A.astype(np.float64), relu_last=True)
    dE, dR = lo.distmult_bce_grads(facts, E, Rel, y)
    grads, _ = O.rgcn_backward(cfgs, params, A.astype(np.float64), tape, dE)
    x = (E[facts[:, 0]] * Rel[facts[:, 1]] * E[facts[:, 2]]).sum(-1)
    want_loss = lo.bce_with_logits(x, y)
    rows = np.unique(np.concatenate([facts[:, 0], facts[:, 2]]))

    def loss_fn(top_rows, H):
        local = np.stack([np.searchsorted(top_rows, facts[:, 0]), facts[:, 1], np.searchsorted(top_rows, facts[:, 2])], 1)
        dEl, dRl = lo.distmult_bce_grads(local, H, Rel, y)
        xs = (H[local[:, 0]] * Rel[local[:, 1]] * H[local[:, 2]]).sum(-1)
        return lo.bce_with_logits(xs, y), dEl, {"relations": dRl}

    got = O.rgcn_train_step_at_rows(cfgs, params, None, A, rows, None, sample_nodes=np.arange(N), relu_last=True,
                                    loss_fn=loss_fn, extra_params={"relations": Rel}, chunk=13)
    np.testing.assert_allclose(got["loss"], want_loss, rtol=1e-12)
    np.testing.assert_allclose(got["extra_grads"]["relations"], dR, rtol=1e-10, atol=1e-15)
    np.testing.assert_allclose(got["grads"][0]["weight_I_comp"], grads[0]["weight_I_comp"], rtol=1e-9, atol=1e-14)
    dense = np.transpose(got["wI"][0]["grad"], (1, 0, 2)).reshape(B * N, -1)
    np.testing.assert_allclose(dense, grads[0]["weight_I"], rtol=1e-9, atol=1e-14)
    flat = [dR] + list(grads[0].values())
    np.testing.assert_allclose(got["grad_norm"], O.clip_grad_norm(flat)[0], rtol=1e-10)
