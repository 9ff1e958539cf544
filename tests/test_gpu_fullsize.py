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
