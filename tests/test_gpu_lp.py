"""GPU parity of the DistMult link-prediction decoder (csrc/distmult.hip through the C ABI) against
the reference's goldens and the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import lp_oracle as lo

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _dev(*arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs]


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_scores_loss_grads_against_reference(tag):
    from mrgcn_amd.tasks import link_prediction as lp
    g = np.load(os.path.join(GOLDEN, "lp_decoder.npz"))
    E, Rel, facts, y = _dev(g[f"{tag}.E"], g[f"{tag}.Rel"], g[f"{tag}.facts"], g[f"{tag}.y"])
    E.requires_grad_(True)
    Rel.requires_grad_(True)
    sc = lp.score_distmult_bc((facts[:, 0], facts[:, 1], facts[:, 2]), E, Rel)
    np.testing.assert_allclose(sc.detach().cpu().numpy(), g[f"{tag}.scores"], rtol=1e-4, atol=1e-4)
    loss = lp.binary_crossentropy(sc, y, torch.nn.BCEWithLogitsLoss())
    assert abs(loss.item() - float(g[f"{tag}.loss"])) < 1e-5
    loss.backward()
    np.testing.assert_allclose(E.grad.cpu().numpy(), g[f"{tag}.dE"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(Rel.grad.cpu().numpy(), g[f"{tag}.dRel"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("filtered", [False, True])
def test_ranks_against_reference_bit_exact(tag, filtered):
    from mrgcn_amd.tasks import link_prediction as lp
    g = np.load(os.path.join(GOLDEN, "lp_decoder.npz"))
    E, Rel = _dev(g[f"{tag}.E"], g[f"{tag}.Rel"])
    ranks = lp.compute_ranks_fast(g[f"{tag}.facts"], E, Rel, 50, filtered).cpu().numpy()
    assert np.array_equal(ranks, g[f"{tag}.ranks_flt" if filtered else f"{tag}.ranks_raw"])


@pytest.mark.parametrize("N,P,H,nf", [(1000, 9, 200, 333), (257, 3, 7, 700), (4099, 20, 64, 1)])
def test_ranks_against_oracle_bit_exact(N, P, H, nf):
    """Ragged sizes (node / fact / h tiles all partial), ReLU-style exact zeros (ties), more
    facts than nodes (the reference's slicing quirk), padded leading dimension."""
    from mrgcn_amd.tasks import link_prediction as lp
    rng = np.random.default_rng(N + nf)
    E = np.maximum(rng.standard_normal((N, H)), 0).astype(np.float32)
    E[rng.choice(N, N // 8, replace=False)] = 0
    Rel = rng.standard_normal((2 * P + 1, H)).astype(np.float32)
    facts = np.stack([rng.integers(0, N, nf), rng.integers(0, P, nf), rng.integers(0, N, nf)], 1).astype(np.int64)
    facts[nf // 2:, 0] = facts[: nf - nf // 2, 0]  # shared (s, p) pairs for the filter
    facts[nf // 2:, 1] = facts[: nf - nf // 2, 1]
    Epad = torch.zeros((N, H + 3), device="cuda")
    Epad[:, :H] = torch.from_numpy(E).cuda()
    Ed, (Rd,) = Epad[:, :H], _dev(Rel)
    for filtered in (False, True):
        got = lp.compute_ranks_fast(facts, Ed, Rd, filtered=filtered).cpu().numpy()
        assert np.array_equal(got, lo.compute_ranks(facts, E, Rel, filtered))
    mrr, hits = lp.mrr_hits(torch.from_numpy(got))
    omrr, ohits = lo.mrr_hits(got)
    assert abs(mrr - omrr) < 1e-6 and np.allclose(hits, ohits)


def test_fb15k_sized_ranks_properties():
    """BASELINE config 4's decoder size (14 541 nodes, 237 relations, h = 200, one test batch of
    500 facts): raw rank >= filtered rank, ranks within [1, N], and an embedding table built so
    that every fact's answer is the unique best candidate ranks 1."""
    from mrgcn_amd.tasks import link_prediction as lp
    N, P, H, nf = 14541, 237, 200, 500
    gen = torch.Generator(device="cuda").manual_seed(0)
    E = torch.randn((N, H), device="cuda", generator=gen)
    Rel = torch.randn((2 * P + 1, H), device="cuda", generator=gen)
    rng = np.random.default_rng(0)
    facts = np.stack([rng.integers(0, N, nf), rng.integers(0, P, nf), rng.integers(0, N, nf)], 1)
    raw = lp.compute_ranks_fast(facts, E, Rel, filtered=False)
    flt = lp.compute_ranks_fast(facts, E, Rel, filtered=True)
    assert int(raw.min()) >= 1 and int(raw.max()) <= N and bool((flt <= raw).all())
    # one-hot embeddings: score(s, p, o) = Rel[p, s] if s == o else 0
    Eh = torch.eye(64, device="cuda")
    Rh = torch.ones((3, 64), device="cuda")
    f2 = np.stack([np.arange(64), np.zeros(64, int), np.arange(64)], 1)
    assert bool((lp.compute_ranks_fast(f2, Eh, Rh) == 1).all())


def test_lp_epoch_on_encoder_output():
    """Encoder (fused R-GCN) + decoder + BCE + clip + Adam in one step; the loss falls."""
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.tasks import link_prediction as lp
    from mrgcn_amd.train import ClipAdam
    import torch.nn as nn
    N, P = 300, 4
    R = 2 * P + 1
    rng = np.random.default_rng(1)
    facts = np.unique(np.stack([rng.integers(0, N, 900), rng.integers(0, P, 900), rng.integers(0, N, 900)], 1), axis=0)
    rows = np.concatenate([facts[:, 0], facts[:, 2], np.arange(N)])
    cols = np.concatenate([facts[:, 1] * N + facts[:, 2], (facts[:, 1] + P) * N + facts[:, 0], 2 * P * N + np.arange(N)])
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.ones(len(rows)), (N, R * N))
    torch.manual_seed(0)
    model = RGCN([(N, 32, "rgcn", nn.ReLU()), (32, 32, "rgcn", None)], R, N, 2, 0.0, True, False, True).cuda()
    A = A.cuda()
    opt = ClipAdam(list(model.parameters()), lr=0.01)
    losses = []
    rs = np.random.RandomState(0)
    for _ in range(8):
        neg, Y = lp.sample_negatives(facts, rs)
        emb = model(None, A)
        tr = torch.from_numpy(np.concatenate([facts, neg])).cuda()
        sc = lp.score_distmult_bc((tr[:, 0], tr[:, 1], tr[:, 2]), emb, model.relations)
        loss = lp.binary_crossentropy(sc, torch.from_numpy(Y).cuda())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    assert model.relations.grad is not None and float(model.relations.grad.abs().sum()) > 0


def test_sorted_backward_equals_scatter_backward():
    """The run-accumulating backward (three passes over sorted triples) against the float64 oracle
    and against the scatter kernel, on a size that takes the sorted path (>= 4096 triples)."""
    import os
    from mrgcn_amd.tasks import link_prediction as lp
    rng = np.random.default_rng(3)
    N, P, H, n = 400, 7, 70, 6000
    E = rng.standard_normal((N, H)).astype(np.float32)
    Rel = rng.standard_normal((2 * P + 1, H)).astype(np.float32)
    facts = np.stack([rng.integers(0, N, n), rng.integers(0, P, n), rng.integers(0, N, n)], 1).astype(np.int64)
    y = (rng.random(n) < 0.8).astype(np.float32)
    dE_ref, dR_ref = lo.distmult_bce_grads(facts, E, Rel, y)
    outs = []
    for flag in ("1", "0"):
        os.environ["MRGCN_LP_SORTED_BWD"] = flag
        Et, Rt = torch.from_numpy(E).cuda().requires_grad_(True), torch.from_numpy(Rel).cuda().requires_grad_(True)
        ft = torch.from_numpy(facts).cuda()
        loss = lp.binary_crossentropy(lp.score_distmult_bc((ft[:, 0], ft[:, 1], ft[:, 2]), Et, Rt),
                                      torch.from_numpy(y).cuda())
        loss.backward()
        outs.append((Et.grad.cpu().numpy(), Rt.grad.cpu().numpy()))
    os.environ.pop("MRGCN_LP_SORTED_BWD")
    for dE, dR in outs:
        np.testing.assert_allclose(dE, dE_ref, rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(dR, dR_ref, rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("H", [200, 64, 70])
def test_stored_orders_of_fixed_facts_and_vector_rows_against_the_oracle(H):
    """A run scores the same training facts every epoch: their orders are stored once (SortedTriples), the corrupted
    facts behind them take the scatter kernel — no sort per backward.  Rows of whole 16-byte pieces (H = 200, 64) go
    through the pipelined kernels (k_distmult_fwd4 / _bwd_sorted4), H = 70 through the element-wise ones; scores and
    both gradients against the float64 oracle."""
    from mrgcn_amd.tasks import link_prediction as lp
    rng = np.random.default_rng(H)
    N, P, n = 500, 9, 9000
    E = rng.standard_normal((N, H)).astype(np.float32)
    Rel = rng.standard_normal((2 * P + 1, H)).astype(np.float32)
    facts = np.stack([rng.integers(0, N, n), rng.integers(0, P, n), rng.integers(0, N, n)], 1).astype(np.int64)
    facts[:3000, 1] = 2          # long runs of one predicate, and of one subject
    facts[3000:3400, 0] = 7
    sampler = lp.DeviceNegativeSampler(torch.from_numpy(facts).cuda(), torch.Generator(device="cuda").manual_seed(1))
    static = lp.SortedTriples(sampler.facts, N, 2 * P + 1)
    for _ in range(2):           # two draws of negatives on the same buffers
        t, y = sampler()
        tn, yn = t.cpu().numpy(), y.cpu().numpy()
        dE_ref, dR_ref = lo.distmult_bce_grads(tn, E, Rel, yn)
        Et, Rt = torch.from_numpy(E).cuda().requires_grad_(True), torch.from_numpy(Rel).cuda().requires_grad_(True)
        sc = lp.score_distmult_bc(t, Et, Rt, static=static)
        ref_sc = (E[tn[:, 0]].astype(np.float64) * Rel[tn[:, 1]] * E[tn[:, 2]]).sum(1)
        np.testing.assert_allclose(sc.detach().cpu().numpy(), ref_sc, rtol=1e-4, atol=1e-4)
        lp.binary_crossentropy(sc, y).backward()
        np.testing.assert_allclose(Et.grad.cpu().numpy(), dE_ref, rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(Rt.grad.cpu().numpy(), dR_ref, rtol=1e-3, atol=1e-6)
    # a triple tensor that does not start with the stored facts is refused (the general path runs)
    other = torch.from_numpy(facts[::-1].copy()).cuda()
    assert not lp.SortedTriples(torch.from_numpy(facts).cuda(), N, 2 * P + 1).covers(other)


@pytest.mark.parametrize("n,N,R", [(1, 5, 3), (5000, 700, 19), (54423, 14541, 475), (30000, 40000, 7), (9000, 50, 20000)])
def test_counting_sort_orders_are_sorted_permutations(n, N, R):
    import ctypes as C
    from mrgcn_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(n)
    tr = torch.from_numpy(np.stack([rng.integers(0, N, n), rng.integers(0, R, n), rng.integers(0, N, n)], 1)).cuda()
    orders = [torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(3)]
    ws = torch.empty(int(lib.mrgcn_distmult_orders_counting_workspace(N, R)), dtype=torch.uint8, device="cuda")
    for _ in range(2):   # (the workspace is re-zeroed inside: a second call on the same buffers)
        L.check(lib.mrgcn_distmult_orders_counting(tr.data_ptr(), n, N, R, orders[0].data_ptr(), orders[1].data_ptr(),
                                                   orders[2].data_ptr(), ws.data_ptr(), ws.numel(),
                                                   torch.cuda.current_stream().cuda_stream))
    t = tr.cpu().numpy()
    for c, o in enumerate(orders):
        o = o.cpu().numpy()
        np.testing.assert_array_equal(np.sort(o), np.arange(n))
        assert np.all(np.diff(t[o, c]) >= 0)


def test_device_negative_sampler_draws_distinct_facts_with_in_batch_replacements():
    from mrgcn_amd.tasks import link_prediction as lp
    rs = np.random.RandomState(1)
    f = np.unique(np.stack([rs.randint(0, 500, 5003), rs.randint(0, 40, 5003), rs.randint(0, 500, 5003)], 1), axis=0)
    facts = torch.from_numpy(f).cuda()
    n = len(f)
    sampler = lp.DeviceNegativeSampler(facts, torch.Generator(device="cuda").manual_seed(0))
    t1, y1 = sampler()
    neg1 = t1[n:].clone()
    t2, y2 = sampler()
    assert t1 is t2 and y1 is y2 and t1.shape == (n + n // 5, 3)
    assert bool((t2[:n] == facts).all()) and bool(y1[:n].all()) and not bool(y1[n:].any())
    neg2 = t2[n:]
    assert not bool((neg1 == neg2).all())                      # another seed, another draw
    nodes = torch.unique(torch.cat([facts[:, 0], facts[:, 2]]))
    nh = (n // 5) // 2
    for neg in (neg1, neg2):
        assert bool(torch.isin(neg[:, 0], nodes).all()) and bool(torch.isin(neg[:, 2], nodes).all())
        # every corrupted fact is a copy of a DISTINCT fact with one end replaced: (p, o) of the head-corrupted ones
        # and (s, p) of the tail-corrupted ones still belong to facts, and the source facts do not repeat
        key = lambda a, b: a * 100003 + b  # noqa: E731
        po = set(key(f[:, 1], f[:, 2]).tolist())
        sp = set(key(f[:, 0], f[:, 1]).tolist())
        ng = neg.cpu().numpy()
        assert all(k in po for k in key(ng[:nh, 1], ng[:nh, 2]).tolist())
        assert all(k in sp for k in key(ng[nh:, 0], ng[nh:, 1]).tolist())


def test_device_negative_sampling_shapes():
    from mrgcn_amd.tasks import link_prediction as lp
    rs = np.random.RandomState(1)
    facts = torch.from_numpy(np.stack([rs.randint(0, 50, 103), rs.randint(0, 4, 103), rs.randint(0, 50, 103)], 1)).cuda()
    neg, Y = lp.sample_negatives_device(facts, torch.Generator(device="cuda").manual_seed(0))
    assert neg.shape == (20, 3) and Y.shape == (123,) and bool(Y[:103].all()) and not bool(Y[103:].any())
    nodes = torch.unique(torch.cat([facts[:, 0], facts[:, 2]]))
    assert bool(torch.isin(neg[:, 0], nodes).all()) and bool(torch.isin(neg[:, 2], nodes).all())
    assert bool(torch.isin(neg[:, 1], facts[:, 1]).all())


@pytest.mark.parametrize("n,N,R", [(1, 1, 1), (1000, 50, 7), (70000, 14541, 475), (5000, 3, 2)])
@pytest.mark.parametrize("which", [(1, 1, 1), (1, 0, 1), (0, 1, 0)])
def test_triple_orders_equal_a_stable_argsort(n, N, R, which):
    """mrgcn_distmult_orders (the permutations of the sorted decoder backward): stable sorts of the subject /
    predicate / object columns, any subset of the three."""
    import ctypes as C

    from mrgcn_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(n + N)
    t = np.stack([rng.integers(0, N, n), rng.integers(0, R, n), rng.integers(0, N, n)], 1).astype(np.int64)
    td = torch.from_numpy(t).cuda()
    outs = [torch.full((n,), -1, dtype=torch.int64, device="cuda") if w else None for w in which]
    ws = torch.empty(int(lib.mrgcn_distmult_orders_workspace(n)), dtype=torch.uint8, device="cuda")
    L.check(lib.mrgcn_distmult_orders(td.data_ptr(), n, N, R, *[o.data_ptr() if o is not None else 0 for o in outs],
                                      ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
    for c, o in enumerate(outs):
        if o is not None:
            np.testing.assert_array_equal(o.cpu().numpy(), np.argsort(t[:, c], kind="stable"))


@pytest.mark.parametrize("n,k", [(1, 1), (2, 2), (103, 20), (1000, 1000), (272115, 54423), (5, 0)])
def test_random_subset_is_a_keyed_bijection(n, k):
    """mrgcn_random_subset_i64 (the corrupted facts of the device-side negative sampler): k distinct indices below n;
    k = n gives a permutation; another seed gives another one."""
    from mrgcn_amd import _lib as L
    lib = L.load()
    outs = []
    for seed in (12345, 987654321):
        sd = torch.tensor([seed], dtype=torch.int64, device="cuda")
        out = torch.full((max(k, 1),), -1, dtype=torch.int64, device="cuda")
        L.check(lib.mrgcn_random_subset_i64(n, k, sd.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
        o = out[:k].cpu().numpy()
        assert ((o >= 0) & (o < n)).all() and len(np.unique(o)) == k
        outs.append(o)
    if k >= 20:
        assert (outs[0] != outs[1]).mean() > 0.5
        if k < n:   # not simply the first k indices, and spread over the range
            assert outs[0].max() > n // 2 and (outs[0] != np.arange(k)).mean() > 0.75
