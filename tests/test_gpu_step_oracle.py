"""Gradients, the clip norm and the Adam step AT THE BENCHMARKED SIZES against the float64 oracle.

`oracle.rgcn_oracle.rgcn_train_step_at_rows` evaluates one epoch of node_classification.py:166-193 over
graph.py:62-102 / rgcn.py:69-89 on the receptive field of the labelled rows (pinned to the reference's own
gradients and post-Adam parameters by tests/test_oracle_golden.py).  Here the MI355X paths that only run at scale —
rows split over several blocks, dozens of node bands, the mix backward on a gradient support, the fused row Adam over
hundreds of thousands of live nodes, the hipGraph replay — are compared with it from identical initial parameters:

  * d weight_I on >= 500 sampled node blocks (labelled nodes, the longest rows of the support, the largest hubs,
    random live nodes, nodes without any gradient), every other gradient whole, the clip norm;
  * the parameters and both Adam moments of those blocks (and of every small parameter) after ONE step, on the default
    path (gradient support, row-sparse fused Adam), on the replayed hipGraph (one step further, from the GPU's own
    state), on the dense path and on the path without any gradient-sparsity shortcut.

Tolerances: gradients rtol 1e-3 plus an absolute term of 2e-5 of the block's (tensor's) largest gradient; a parameter
after the step must lie inside the interval Adam maps that gradient uncertainty to (an element may move by a whole
lr-sized step only where its gradient is that close to zero), so one mishandled node block fails."""
import numpy as np
import pytest
import os

import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu

LR, B1, B2, EPS = 0.01, 0.9, 0.999, 1e-8


# ---- comparison helpers -------------------------------------------------------------------------------------------
def _gtol(ref, per_block):
    """Absolute gradient tolerance: 2e-5 of the largest element of the block (axis 0 = blocks) or of the tensor,
    plus 1e-6 of the tensor's largest (sums over hub columns cancel)."""
    a = np.abs(ref)
    top = float(a.max()) if a.size else 0.0
    if per_block and ref.ndim > 1:
        blk = a.reshape(a.shape[0], -1).max(1).reshape((-1,) + (1,) * (ref.ndim - 1))
        return 2e-5 * blk + 1e-6 * top + 1e-30
    return 2e-5 * top + 1e-30


def _close_grad(got, ref, name, per_block=False, rtol=1e-3, uncertain=None, max_uncertain=5e-4):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    bad = np.abs(got - ref) > rtol * np.abs(ref) + _gtol(ref, per_block)
    if uncertain is not None:   # elements that hang on a unit at the ReLU kink (`_kink_mask`)
        assert uncertain.mean() <= max_uncertain, (name, "too many elements declared uncertain", int(uncertain.sum()))
        bad &= ~uncertain
    assert not bad.any(), (name, int(bad.sum()), float(np.abs(got - ref).max()), np.argwhere(bad)[:5].tolist())


def _adam_update(g, m0, v0, t, coef):
    gc = g * coef
    m = B1 * m0 + (1 - B1) * gc
    v = B2 * v0 + (1 - B2) * gc * gc
    return LR * (m / (1 - B1 ** t)) / (np.sqrt(v) / np.sqrt(1 - B2 ** t) + EPS), m, v


KINK = 3e-7   # a ReLU unit whose float64 pre-activation is within KINK of zero, relative to the sum of the magnitudes
              # of its terms (five float32 roundings of that sum; the flips seen sat at 0.1 - 1.2 roundings), may carry
              # the other mask in a float32 forward


def _kink_mask(shape, kinks, sample=None):
    """bool [N, B, F] (node-major weight_I blocks): the elements whose gradient holds a term that a unit on the ReLU
    kink switches on or off — feature f of every source node of such a unit's row (oracle: `_kinks`).  `sample`
    (sorted node ids): the mask over those nodes' blocks only, [len(sample), B, F]."""
    mask = np.zeros(shape, dtype=bool)
    if kinks is None:
        return mask
    for pre, mag, col, nodes in zip(kinks["pre"], kinks["mag"], kinks["col"], kinks["nodes"]):
        if abs(pre) <= KINK * mag:
            if sample is not None:   # positions of the unit's source nodes among the sampled blocks
                nodes = np.asarray(nodes)
                at = np.searchsorted(sample, nodes)
                ok = at < len(sample)
                ok[ok] &= sample[at[ok]] == nodes[ok]
                nodes = at[ok]
            mask[nodes, :, col] = True
    return mask


def _check_adam(name, p_before, g, coef, got_p, got_m, got_v, m0=0.0, v0=0.0, t=1, per_block=False, uncertain=None,
                max_uncertain=5e-4):
    """`got_*` (the GPU's parameter and moments after the step) against Adam applied to the oracle's gradient `g`
    (unclipped; `coef` the oracle's clip coefficient) with the gradient tolerance mapped through the update.
    `uncertain` (bool, like the parameter): elements left out — their gradient depends on which side of the ReLU kink a
    float32 forward puts a unit that float64 has within rounding of zero (`_kink_mask`); at most `max_uncertain` of the
    tensor (5e-4 of a whole parameter; a sample of node blocks that seeks out hubs and long rows: 2e-2)."""
    if uncertain is not None:
        assert uncertain.mean() <= max_uncertain, (name, "too many elements declared uncertain", int(uncertain.sum()))
    certain = True if uncertain is None else ~uncertain
    g = np.asarray(g, np.float64)
    p0 = np.asarray(p_before, np.float64)
    d = 1e-3 * np.abs(g) + _gtol(g, per_block)
    ups = [_adam_update(g + s * d, m0, v0, t, coef) for s in (-1.0, 0.0, 1.0)]
    u = np.stack([x[0] for x in ups])
    lo, hi = p0 - u.max(0), p0 - u.min(0)
    tol = 1e-7 + 3e-7 * np.abs(p0)
    gp = np.asarray(got_p, np.float64)
    bad = ((gp < lo - tol) | (gp > hi + tol)) & certain
    over = np.where(certain, np.maximum(lo - tol - gp, gp - hi - tol) / (hi - lo + 2 * tol), -1.0)  # (in interval widths)
    if os.environ.get("MRGCN_TEST_MARGINS"):   # (how close the closest element comes: <= 0 inside, in interval widths)
        print("margin", name, float(over.max()), np.unravel_index(int(over.argmax()), over.shape))
    assert not bad.any(), (name, "parameter", int(bad.sum()), np.argwhere(bad)[:5].tolist(), float(over.max()))
    _, m_ref, v_ref = ups[1]
    dm = (1 - B1) * coef * d
    bad = (np.abs(np.asarray(got_m, np.float64) - m_ref) > dm + 1e-6 * np.abs(m_ref) + 1e-30) & certain
    assert not bad.any(), (name, "exp_avg", int(bad.sum()), np.argwhere(bad)[:5].tolist())
    # exp_avg_sq is quadratic in the gradient: compared through its square root
    sv_ref, sv_got = np.sqrt(v_ref), np.sqrt(np.maximum(np.asarray(got_v, np.float64), 0.0))
    dv = np.sqrt(1 - B2) * coef * d
    bad = (np.abs(sv_got - sv_ref) > 1.5 * dv + 2e-6 * sv_ref + 1e-30) & certain
    assert not bad.any(), (name, "exp_avg_sq", int(bad.sum()), np.argwhere(bad)[:5].tolist())


def _np(t):
    return t.detach().cpu().numpy()


# ---- node classification (AM, synth10m) ------------------------------------------------------------------------------
class _NcCase:
    def __init__(self, name, scale=1.0, seed=0, labelled=None):
        from mrgcn_amd import synth
        from mrgcn_amd.models.rgcn import RGCN
        from mrgcn_amd.plan import GraphPlan
        from oracle import rgcn_oracle as O
        self.O = O
        g = synth.make_graph(name, seed=0, scale=scale)
        self.g = g
        N, R = g.num_nodes, g.num_relations
        self.N, self.R = N, R
        self.dims = synth.layer_dims(name)
        self.B = synth.SHAPES[name]["bases"]
        A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals),
                                    (N, R * N)).cuda()
        torch.manual_seed(seed)
        d = self.dims
        self.model = RGCN([(d[0][0], d[0][1], "mrgcn", torch.nn.ReLU()), (d[1][0], d[1][1], "mrgcn", None)],
                          R, N, self.B, 0.0, False, True, False).cuda()
        with torch.no_grad():   # (zeros at init: make the bias terms and their gradients' paths visible)
            for layer in self.model.layers.values():
                layer.b.normal_(0.0, 0.05)
        self.plan = GraphPlan(A, N, R, operand_row_bytes=self.model.operand_row_bytes())
        del A
        self.A = self.plan.as_adjacency_handle()
        self.A_csr = sp.csr_matrix((g.vals.astype(np.float64), (g.rows, g.cols)), shape=(N, R * N))
        self.X = torch.randn((N, d[0][0]), device="cuda", generator=torch.Generator("cuda").manual_seed(seed + 1))
        self.X_host = self.X.cpu().numpy()
        idx, y = synth.make_labels(name, N, seed=0, scale=scale)
        if labelled:
            idx, y = idx[:labelled], y[:labelled]
        self.idx_np, self.y_np = idx, y
        self.idx, self.y = torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()
        self.init = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        self.cfgs = O.rgcn_cfgs(self.dims, R, N, self.B, True, False)
        self.sample = self._sample_nodes()
        self.sel = torch.from_numpy(self.sample).cuda()
        self.wI = self.model.layers["layer_0"].weight_I
        self.small = [(n, p) for n, p in self.model.named_parameters() if p is not self.wI]
        self.ora1 = self.oracle(self.init, None, 1)

    def _sample_nodes(self):
        """>= 500 node blocks of weight_I: labelled nodes, the longest rows among the layer-0 support's rows (split rows),
        the largest hubs, random live nodes, and nodes outside the receptive field (no gradient at all)."""
        N = self.N
        rng = np.random.default_rng(11)
        A = self.A_csr
        s1 = np.unique(A[self.idx_np].indices % N)            # the rows with gradient below the top layer
        live = np.unique(A[s1].indices % N)                   # the nodes whose weight_I block gets any
        rowlen = np.diff(A.indptr)
        longest = s1[np.argsort(rowlen[s1])[-40:]]
        hubs = np.intersect1d(np.argsort(rowlen)[-60:], live)
        dead = np.setdiff1d(np.arange(N), live)
        picks = [self.idx_np[:100], longest, hubs, rng.choice(live, 340, replace=False)]
        if len(dead):
            picks.append(rng.choice(dead, min(60, len(dead)), replace=False))
        self.n_live = len(live)
        self.dead = dead
        out = np.unique(np.concatenate(picks))
        assert len(out) >= 500, len(out)
        self.longest_row = int(rowlen[longest].max())
        return out

    def state_np(self, sd=None):
        sd = self.model.state_dict() if sd is None else sd
        return {k: _np(v) for k, v in sd.items()}

    def oracle(self, sd, moments, t):
        O = self.O
        st = self.state_np(sd)
        return O.rgcn_train_step_at_rows(self.cfgs, O.split_params(st, len(self.cfgs)), self.X_host, self.A_csr,
                                         self.idx_np, self.y_np, sample_nodes=self.sample, moments=moments, t=t, lr=LR)

    def reset(self):
        self.model.load_state_dict(self.init)
        self.model.zero_grad(set_to_none=True)

    def ora_grad(self, ora, name):
        li, key = int(name.split(".")[1].split("_")[1]), name.split(".")[2]
        return ora["grads"][li][key]

    def kinks(self, ora):
        """bool [len(sample), B, F]: the sampled weight_I elements that hang on a hidden unit at the ReLU kink."""
        kk = ora["levels"][0]["kinks"]
        m = _kink_mask((len(self.sample), self.B, self.dims[0][1]), kk, sample=self.sample)
        if os.environ.get("MRGCN_TEST_MARGINS"):
            print("kinks", int((np.abs(kk["pre"]) <= KINK * kk["mag"]).sum()), "units within the band,",
                  int(m.sum()), "of", m.size, "sampled elements left out")
        return m

    def check_small_grads(self, ora, where):
        for n, p in self.small:
            assert p.grad is not None, (where, n)
            _close_grad(_np(p.grad), self.ora_grad(ora, n), f"{where}: grad {n}")

    def check_after_step(self, ora, opt, before, where, moments_before=None, t=1):
        """Parameters and Adam moments after the step against the oracle record `ora` (computed from `before`)."""
        coef = ora["coef"]
        for n, p in self.small:
            st = opt.state[p]
            m0, v0 = moments_before[n] if moments_before else (0.0, 0.0)
            _check_adam(f"{where}: {n}", before[n], self.ora_grad(ora, n), coef, _np(p), _np(st["exp_avg"]),
                        _np(st["exp_avg_sq"]), m0, v0, t)
        st = opt.state[self.wI]
        m0, v0 = moments_before["wI"] if moments_before else (0.0, 0.0)
        _check_adam(f"{where}: weight_I blocks", before["wI"], ora["wI"][0]["grad"], coef, _np(self.wI[self.sel]),
                    _np(st["exp_avg"][self.sel]), _np(st["exp_avg_sq"][self.sel]), m0, v0, t, per_block=True,
                    uncertain=self.kinks(ora), max_uncertain=2e-2)
        # a node outside the receptive field never moves and never gets moments
        if len(self.dead):
            d = torch.from_numpy(self.dead[:: max(len(self.dead) // 5000, 1)]).cuda()
            N, Bn, out = self.wI.shape
            ref0 = self.init["layers.layer_0.weight_I"].view(Bn, N, out).permute(1, 0, 2)
            assert torch.equal(self.wI.detach()[d], ref0[d]), where
            assert not bool(st["exp_avg"][d].any()) and not bool(st["exp_avg_sq"][d].any()), where

    def snapshot(self, opt=None):
        snap = {n: _np(p).astype(np.float64) for n, p in self.small}
        snap["wI"] = _np(self.wI[self.sel]).astype(np.float64)
        if opt is None:
            return snap
        mom = {n: (_np(opt.state[p]["exp_avg"]).astype(np.float64), _np(opt.state[p]["exp_avg_sq"]).astype(np.float64))
               for n, p in self.small}
        st = opt.state[self.wI]
        mom["wI"] = (_np(st["exp_avg"][self.sel]).astype(np.float64), _np(st["exp_avg_sq"][self.sel]).astype(np.float64))
        return snap, mom

    def oracle_moments(self, mom):
        out = [dict() for _ in self.cfgs]
        for n, (m, v) in mom.items():
            if n == "wI":
                out[0]["weight_I"] = (m, v)
            else:
                out[int(n.split(".")[1].split("_")[1])][n.split(".")[2]] = (m, v)
        return out


def _row_sparse_backward(case):
    """Forward, loss and backward as train_step runs them (row-sparse weight_I gradient), without the optimizer step.
    Returns (loss, the row-sparse entry of weight_I)."""
    from mrgcn_amd import functional as Fn
    from mrgcn_amd.train import categorical_crossentropy
    m = case.model
    Fn.clear_row_grads(list(m.parameters()))
    logits = m(case.X, case.A)
    loss = categorical_crossentropy(logits, case.idx, case.y, sole_consumer=True)
    m.zero_grad(set_to_none=True)
    prev = Fn.row_sparse_weight_grad(True)
    try:
        loss.backward()
    finally:
        Fn.row_sparse_weight_grad(prev)
    ent = Fn.pop_row_grad(case.wI)
    # (detached: a live autograd graph keeps its AccumulateGrad nodes on THIS stream, which breaks a later capture)
    return logits.detach(), loss.detach(), ent


def _run_nc(case):
    from mrgcn_amd import functional as Fn
    from mrgcn_amd.train import ClipAdam, GraphedTrainStep, train_step
    ora1 = case.ora1
    fwd = lambda: case.model(case.X, case.A)   # noqa: E731

    # (1) the default backward: on the gradient support of the label set, weight_I's gradient row-sparse
    case.reset()
    logits, loss, ent = _row_sparse_backward(case)
    np.testing.assert_allclose(_np(logits)[case.idx_np], ora1["logits"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(float(loss), ora1["loss"], rtol=2e-5, atol=1e-6)
    assert ent is not None and case.wI.grad is None, "the row-sparse path was not taken"
    assert ent["fused"] is not None and ent["fused"].get("sup") is not None, "not on the gradient support / fused Adam"
    assert int(ent["cur"].sum()) == case.n_live == len(ora1["wI"][0]["live_nodes"])
    g = Fn.dense_from_rows(case.wI, ent)
    _close_grad(_np(g[case.sel]), ora1["wI"][0]["grad"], "support path: d weight_I blocks", per_block=True,
                uncertain=case.kinks(ora1), max_uncertain=2e-2)
    if len(case.dead):
        d = torch.from_numpy(case.dead[:: max(len(case.dead) // 5000, 1)]).cuda()
        assert not bool(g[d].any())
    del g
    case.check_small_grads(ora1, "support path")
    np.testing.assert_allclose(float(torch.sqrt(ent["sumsq"] + sum((p.grad.double() ** 2).sum() for _, p in case.small))),
                               ora1["grad_norm"], rtol=2e-5)

    # (2) one eager step of the default train_step (fused row Adam), then one more replayed from a hipGraph
    import mrgcn_amd
    case.reset()
    before = case.snapshot()
    opt = ClipAdam(case.model.parameters(), lr=LR, max_norm=1.0, capturable=True)
    mrgcn_amd.reset_stats()
    step = GraphedTrainStep(case.model, fwd, case.idx, case.y, opt, warmup=1)    # one eager epoch, then the capture
    st = mrgcn_amd.stats()   # the eager epoch + the captured one: both layers on their gradient supports, every time
    assert st.get("backward.support") == 4 and st.get("weight_I.fused_rows") == 2 and st.get("adam.list") == 2, st
    assert not any(k in st for k in ("backward.marking", "backward.general", "weight_I.dense", "adam.rows")), st
    np.testing.assert_allclose(opt.last_grad_norm(), ora1["grad_norm"], rtol=2e-5)
    case.check_after_step(ora1, opt, before, "default path, eager step 1")
    before2, mom2 = case.snapshot(opt)
    sd2 = {k: v.detach().clone() for k, v in case.model.state_dict().items()}
    loss2 = float(step())                                                          # the replay: step 2
    ora2 = case.oracle(sd2, case.oracle_moments(mom2), 2)
    np.testing.assert_allclose(loss2, ora2["loss"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(opt.last_grad_norm(), ora2["grad_norm"], rtol=2e-5)
    case.check_after_step(ora2, opt, before2, "default path, replayed step 2", moments_before=mom2, t=2)
    assert all(int(s["step"]) == 2 for s in opt.state_dict()["state"].values())
    del step, opt

    # (3) the dense gradient in .grad and the plain dense Adam kernel, still on the gradient support
    case.reset()
    opt = ClipAdam(case.model.parameters(), lr=LR, max_norm=1.0)
    loss = train_step(case.model, fwd, case.idx, case.y, opt, row_sparse=False)
    np.testing.assert_allclose(float(loss), ora1["loss"], rtol=2e-5, atol=1e-6)
    assert case.wI.grad is not None
    _close_grad(_np(case.wI.grad[case.sel]), ora1["wI"][0]["grad"], "dense path: d weight_I blocks", per_block=True,
                uncertain=case.kinks(ora1), max_uncertain=2e-2)
    case.check_small_grads(ora1, "dense path")
    np.testing.assert_allclose(opt.last_grad_norm(), ora1["grad_norm"], rtol=2e-5)
    case.check_after_step(ora1, opt, before, "dense path, step 1")
    del opt

    # (4) no gradient-sparsity shortcut at all: general transposed product, plan-level mix backward
    case.reset()
    opt = ClipAdam(case.model.parameters(), lr=LR, max_norm=1.0)
    prev, Fn._LIVE_COLS = Fn._LIVE_COLS, False
    try:
        loss = train_step(case.model, fwd, case.idx, case.y, opt, row_sparse=False)
    finally:
        Fn._LIVE_COLS = prev
    _close_grad(_np(case.wI.grad[case.sel]), ora1["wI"][0]["grad"], "plain path: d weight_I blocks", per_block=True,
                uncertain=case.kinks(ora1), max_uncertain=2e-2)
    case.check_small_grads(ora1, "plain path")
    np.testing.assert_allclose(opt.last_grad_norm(), ora1["grad_norm"], rtol=2e-5)
    case.check_after_step(ora1, opt, before, "plain path, step 1")


@pytest.fixture(scope="module")
def am_case():
    case = _NcCase("am")
    yield case
    del case
    torch.cuda.empty_cache()


def test_am_gradients_clip_and_adam_against_the_float64_oracle_at_full_size(am_case):
    """BASELINE config 3 at the benchmarked size (N = 1 666 764, R = 267, 40 bases, 155 -> 10 -> 11, 1 000 labels)."""
    case = am_case
    assert (case.N, case.R) == (1666764, 267) and case.longest_row > 512   # (rows split over several blocks take part)
    _run_nc(case)


BF16_TOL = 2e-2   # the bf16 run's stated tolerance, relative to the largest element of the compared tensor


def _close_bf16(got, ref, name, per_block=False):
    """A gradient of the bf16 run against the fp32 oracle's.  ReLU's derivative is discontinuous: a hidden unit whose
    pre-activation lies within the bf16 rounding of zero switches its whole gradient term on or off, so single
    elements (node blocks) may differ by a whole term however small the rounding.  Hence: relative L2 error of the
    tensor <= 5e-2, cosine >= 0.995, and at most 10 % of the elements further than 2e-2 of their block's (tensor's)
    largest element from the oracle."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    a = np.abs(ref)
    if per_block and ref.ndim > 1:
        top = a.reshape(a.shape[0], -1).max(1).reshape((-1,) + (1,) * (ref.ndim - 1))
        tol = BF16_TOL * top + 1e-3 * float(a.max())
    else:
        tol = BF16_TOL * float(a.max())
    bad = np.abs(got - ref) > tol + 1e-30
    nr = float(np.linalg.norm(ref))
    rel = float(np.linalg.norm(got - ref)) / max(nr, 1e-300)
    cos = float(got.ravel() @ ref.ravel()) / max(float(np.linalg.norm(got)) * nr, 1e-300)
    assert rel <= 5e-2 and cos >= 0.995 and (bad.mean() <= 0.10 or got.size < 1000), (name, rel, cos, float(bad.mean()))


def _close_bf16_values(got, ref, name):
    """forward values (logits): every element within 2e-2 of the largest"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = float(np.abs(got - ref).max())
    assert err <= BF16_TOL * float(np.abs(ref).max()), (name, err)


def test_am_bf16_pipeline_against_the_float64_oracle_at_full_size(am_case):
    """BASELINE config 3 names bf16 (the reference has none: graph.py:93-95 is fp32).  The bf16 PIPELINE at the
    benchmarked size — X read as bf16 rows by the layer-0 transform on v_mfma_f32_16x16x32_bf16 (weights rounded as they
    are staged), the feature term's rows and the compact operands in bf16, layer-0 dW gathering the bf16 rows; fp32
    parameters, accumulation, outputs and optimizer — against the float64 oracle of the fp32 model: logits of the
    labelled rows, the loss, d weight_I on the sampled node blocks, every other gradient, the clip norm, and the
    parameters after one step of the default path (they move by at most lr, so only their direction can differ: checked
    through the first moment).  Logits: 2e-2 of the largest.  Gradients: `_close_bf16` (L2 5e-2, cosine, 10 % outliers at
    2e-2 — hidden units at the ReLU's kink switch whole terms); the basis coefficients' gradients (sums over every
    column of a relation, cancelling to a small remainder) by cosine > 0.99 and norm within 5 %."""
    import mrgcn_amd
    from mrgcn_amd import functional as Fn
    from mrgcn_amd.train import ClipAdam, train_step
    case = am_case
    ora1 = case.ora1
    case.reset()
    case.model.set_operand_dtype("bf16")
    try:
        mrgcn_amd.reset_stats()
        logits, loss, ent = _row_sparse_backward(case)
        st = mrgcn_amd.stats()
        assert st.get("bf16.xform_xbf16") == 1 and st.get("bf16.dw_xbf16") == 1, st   # the pipeline really ran
        assert st.get("backward.support") == 2, st
        _close_bf16_values(_np(logits)[case.idx_np], ora1["logits"], "bf16 logits")
        err = np.abs(_np(logits)[case.idx_np] - ora1["logits"]).max()
        assert err > 1e-6, "bit-equal to the fp32 result: the bf16 path did not run"
        assert abs(float(loss) - ora1["loss"]) <= BF16_TOL * abs(ora1["loss"])
        g = Fn.dense_from_rows(case.wI, ent)
        _close_bf16(_np(g[case.sel]), ora1["wI"][0]["grad"], "bf16: d weight_I blocks", per_block=True)
        del g
        for n, p in case.small:
            if n.endswith("_comp"):   # sums over every column of a relation that cancel to a small remainder: by
                a, b = _np(p.grad).astype(np.float64).ravel(), np.asarray(case.ora_grad(ora1, n), np.float64).ravel()
                cos = float(a @ b) / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300)   # direction and size
                assert cos > 0.99 and abs(np.linalg.norm(a) / np.linalg.norm(b) - 1) < 5e-2, (n, cos)
            else:
                _close_bf16(_np(p.grad), case.ora_grad(ora1, n), f"bf16: grad {n}")
        norm = float(torch.sqrt(ent["sumsq"] + sum((p.grad.double() ** 2).sum() for _, p in case.small)))
        assert abs(norm - ora1["grad_norm"]) <= BF16_TOL * ora1["grad_norm"]
        # one step of the default path: first moments = (1 - beta1) * clipped gradient
        case.reset()
        opt = ClipAdam(case.model.parameters(), lr=LR, max_norm=1.0)
        train_step(case.model, lambda: case.model(case.X, case.A), case.idx, case.y, opt)
        coef = ora1["coef"]
        m_ref = (1 - B1) * coef * ora1["wI"][0]["grad"]
        _close_bf16(_np(opt.state[case.wI]["exp_avg"][case.sel]), m_ref, "bf16: exp_avg of weight_I blocks", per_block=True)
        for n, p in case.small:
            if not n.endswith("_comp"):
                _close_bf16(_np(opt.state[p]["exp_avg"]), (1 - B1) * coef * case.ora_grad(ora1, n), f"bf16: exp_avg {n}")
        del opt
    finally:
        case.model.set_operand_dtype("f32")
        case.reset()


def test_the_step_check_catches_one_bad_element_of_one_node_block():
    """The comparison itself: after a correct default step at AM/20, ONE element of ONE sampled live node block is
    moved by 2e-5 (parameter) / has its first moment scaled by 1.01 — either must fail the check."""
    from mrgcn_amd.train import ClipAdam, train_step
    case = _NcCase("am", scale=0.05, seed=1)
    case.reset()
    before = case.snapshot()
    opt = ClipAdam(case.model.parameters(), lr=LR, max_norm=1.0)
    train_step(case.model, lambda: case.model(case.X, case.A), case.idx, case.y, opt)
    case.check_after_step(case.ora1, opt, before, "unmodified")
    live = np.flatnonzero(np.abs(case.ora1["wI"][0]["grad"]).reshape(len(case.sample), -1).max(1) > 0)
    node = int(case.sample[live[len(live) // 2]])
    k = int(np.abs(case.ora1["wI"][0]["grad"][live[len(live) // 2]]).argmax())
    with torch.no_grad():
        case.wI.view(case.N, -1)[node, k] += 2e-5
    with pytest.raises(AssertionError, match="parameter"):
        case.check_after_step(case.ora1, opt, before, "one parameter element moved")
    with torch.no_grad():
        case.wI.view(case.N, -1)[node, k] -= 2e-5
        opt.state[case.wI]["exp_avg"].view(case.N, -1)[node, k] *= 1.01
    with pytest.raises(AssertionError, match="exp_avg"):
        case.check_after_step(case.ora1, opt, before, "one moment element scaled")


def test_am_quarter_gradients_clip_and_adam_against_the_float64_oracle():
    """The same at AM/4 with another seed (a cheaper second sample of the same kernels)."""
    _run_nc(_NcCase("am", scale=0.25, seed=3))


def test_synth10m_gradients_clip_and_adam_against_the_float64_oracle_at_full_size():
    """BASELINE config 5 (N = 10 M, R = 101, 10 bases, 155 -> 16 -> 11): the label set is cut to 2 000 of the 10 000 so
    that the host side stays within a minute or two; the receptive field still spans 4.8 M nodes."""
    case = _NcCase("synth10m", labelled=2000)
    assert (case.N, case.R) == (10_000_000, 101)
    _run_nc(case)


# ---- link prediction (FB15k-237 full shape) --------------------------------------------------------------------------
def test_fb15k_lp_step_against_the_float64_oracle_at_full_shape():
    """BASELINE config 4 at the full shape (N = 14 541, R = 475, one featureless layer -> 200 with ReLU, 2 bases;
    272 115 training facts + 20 % corrupted ones): one epoch of link_prediction.py:244-326 as bench.py runs it —
    negatives from the device sampler, encoder, DistMult scores, BCE, backward, clip, Adam — against the float64
    oracle on the very triples the step scored: loss, every gradient (weight_I at ALL node blocks), the clip norm and
    the parameters / moments after the step; then one replayed step from the GPU's own state after three."""
    from mrgcn_amd import synth
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.tasks import link_prediction as lp
    from mrgcn_amd.train import ClipAdam
    from oracle import lp_oracle as lo
    from oracle import rgcn_oracle as O
    g = synth.make_graph("fb15k", seed=0, scale=1.0)
    N, R, H, Bn = g.num_nodes, g.num_relations, 200, 2
    assert (N, R) == (14541, 475)
    rng = np.random.RandomState(0)
    perm = rng.permutation(len(g.triples))
    train = g.triples[perm[: int(272115 / 310116 * len(g.triples))]]
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals), (N, R * N)).cuda()
    A_csr = sp.csr_matrix((g.vals.astype(np.float64), (g.rows, g.cols)), shape=(N, R * N))
    torch.manual_seed(0)
    model = RGCN([(0, H, "mrgcn", torch.nn.ReLU())], R, N, Bn, 0.0, True, False, True).cuda()
    wI = model.layers["layer_0"].weight_I
    opt = ClipAdam(model.parameters(), lr=LR, max_norm=1.0)
    sampler = lp.DeviceNegativeSampler(torch.from_numpy(train).cuda(), None)
    static = lp.SortedTriples(sampler.facts, N, R)
    cfgs = O.rgcn_cfgs([(0, H)], R, N, Bn, False, True)
    nodes = np.arange(N)
    used = {}

    def step():
        t, Y = sampler()
        used["t"], used["y"] = t, Y
        emb = model(None, A)
        loss = lp.binary_crossentropy(lp.score_distmult_bc(t, emb, model.relations, static=static), Y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss.detach()

    def oracle(sd, triples, y, moments, mom_rel, t):
        st = {k: _np(v) for k, v in sd.items()}
        Rel = st.pop("relations").astype(np.float64)
        rows = np.unique(np.concatenate([triples[:, 0], triples[:, 2]]))

        def loss_fn(top_rows, Hm):
            local = np.stack([np.searchsorted(top_rows, triples[:, 0]), triples[:, 1],
                              np.searchsorted(top_rows, triples[:, 2])], 1)
            dE, dR = lo.distmult_bce_grads(local, Hm, Rel, y)
            xs = (Hm[local[:, 0]] * Rel[local[:, 1]] * Hm[local[:, 2]]).sum(-1)
            return lo.bce_with_logits(xs, y), dE, {"relations": dR}

        return O.rgcn_train_step_at_rows(cfgs, O.split_params(st, 1), None, A_csr, rows, None, sample_nodes=nodes,
                                         moments=moments, t=t, lr=LR, relu_last=True, loss_fn=loss_fn,
                                         extra_params={"relations": Rel}, moments_extra=mom_rel)

    def snapshot():
        return dict(wI=_np(wI).astype(np.float64), comp=_np(model.layers["layer_0"].weight_I_comp).astype(np.float64),
                    rel=_np(model.relations).astype(np.float64))

    def check(ora, before, where, mom=None, t=1):
        coef = ora["coef"]
        comp, rel = model.layers["layer_0"].weight_I_comp, model.relations
        for name, p, gref, key in (("weight_I_comp", comp, ora["grads"][0]["weight_I_comp"], "comp"),
                                   ("relations", rel, ora["extra_grads"]["relations"], "rel"),
                                   ("weight_I", wI, ora["wI"][0]["grad"], "wI")):
            s = opt.state[p]
            m0, v0 = mom[key] if mom else (0.0, 0.0)
            unc = None
            if name == "weight_I":   # every node block is compared: the few that hang on a unit at the ReLU kink are not
                kk = ora["levels"][0]["kinks"]
                unc = _kink_mask(tuple(p.shape), kk)
                if os.environ.get("MRGCN_TEST_MARGINS"):
                    o = np.argsort(np.abs(kk["pre"]) / kk["mag"])[:3]
                    print("kinks", where, [(int(kk["row"][i]), int(kk["col"][i]), float(kk["pre"][i]), float(kk["mag"][i]),
                                            len(kk["nodes"][i])) for i in o], int(unc.sum()))
            _check_adam(f"{where}: {name}", before[key], gref, coef, _np(p), _np(s["exp_avg"]), _np(s["exp_avg_sq"]),
                        m0, v0, t, per_block=(name == "weight_I"), uncertain=unc)

    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    before = snapshot()
    loss = float(step())
    tr, yy = _np(used["t"]).copy(), _np(used["y"]).astype(np.float64)
    assert len(tr) == len(train) + len(train) // 5 and np.array_equal(tr[: len(train)], train)
    ora = oracle(sd0, tr, yy, None, None, 1)
    np.testing.assert_allclose(loss, ora["loss"], rtol=2e-5, atol=1e-6)
    _close_grad(_np(wI.grad), ora["wI"][0]["grad"], "d weight_I (all node blocks)", per_block=True)
    _close_grad(_np(model.layers["layer_0"].weight_I_comp.grad), ora["grads"][0]["weight_I_comp"], "d weight_I_comp")
    _close_grad(_np(model.relations.grad), ora["extra_grads"]["relations"], "d relations")
    np.testing.assert_allclose(opt.last_grad_norm(), ora["grad_norm"], rtol=2e-5)
    check(ora, before, "eager step 1")

    # the captured epoch: two more eager steps, then a replay (step 4) from the GPU's state after three
    from mrgcn_amd.train import GraphedStep
    model.load_state_dict(sd0)
    opt = ClipAdam(model.parameters(), lr=LR, max_norm=1.0, capturable=True)
    graphed = GraphedStep(step, warmup=3)
    sd3 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    before = snapshot()

    def moments_of(p):
        s = opt.state[p]
        return _np(s["exp_avg"]).astype(np.float64), _np(s["exp_avg_sq"]).astype(np.float64)
    mom = dict(wI=moments_of(wI), comp=moments_of(model.layers["layer_0"].weight_I_comp), rel=moments_of(model.relations))
    loss = float(graphed())
    tr, yy = _np(sampler.buf).copy(), _np(sampler.labels).astype(np.float64)
    ora = oracle(sd3, tr, yy, [dict(weight_I=mom["wI"], weight_I_comp=mom["comp"])], {"relations": mom["rel"]}, 4)
    np.testing.assert_allclose(loss, ora["loss"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(opt.last_grad_norm(), ora["grad_norm"], rtol=2e-5)
    check(ora, before, "replayed step 4", mom=mom, t=4)
