"""bf16 dense operand (SURVEY §8b's {bf16 dense} SpMM and §8d's additional bf16 run): the operand
is *stored* in bf16, values / accumulation / outputs stay fp32.  The reference has no bf16, so the
contracts are: (1) the product equals the float64 product of the bf16-rounded operand to fp32
accuracy, (2) the operand builders round the same fp32 value the f32 builders store (bit-exact
against torch's round-to-nearest-even), (3) model logits stay within 2e-2 of the fp32 engine."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests import util
from tests.test_gpu_plan_spmm import _plan_from_coo, _random_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def skewed():
    rng = np.random.default_rng(12)
    N, R, num_rows = 1500, 4, 1500
    rows, cols, vals = _random_graph(rng, num_rows, N, R, 20000, hub_rows=2, hub_len=1300, hub_cols=2)
    plan = _plan_from_coo(rows, cols, vals, num_rows, N, R)
    A = sp.csr_matrix((vals.astype(np.float64), (rows, cols)), shape=(num_rows, R * N))
    return plan, A, util.numpy_plan(rows, cols, vals, num_rows, N, R), rng


@pytest.mark.parametrize("F", [1, 2, 3, 7, 8, 10, 11, 16, 17, 40, 64, 200, 300])
def test_spmm_bf16_operand_all_views(skewed, F):
    from mrgcn_amd import _lib as L
    plan, A, ref, rng = skewed
    RN = A.shape[1]
    Dg = torch.from_numpy(rng.standard_normal((RN, F)).astype(np.float32)).cuda().to(torch.bfloat16)
    D64 = Dg.float().cpu().numpy().astype(np.float64)   # the rounded operand, exactly
    Y_ref = A @ D64
    np.testing.assert_allclose(plan.spmm(L.VIEW_LITERAL, Dg).cpu().numpy(), Y_ref, rtol=1e-4, atol=1e-4)
    for ld in sorted({F, (F + 1) // 2 * 2, (F + 3) // 4 * 4, (F + 7) // 8 * 8, F + 5}):
        M = torch.full((plan.ncols, ld), 1e30, dtype=torch.bfloat16, device="cuda")  # padding must not leak
        M[torch.from_numpy(ref["mpos"]).long().cuda(), :F] = Dg[torch.from_numpy(ref["ulcol"]).long().cuda()]
        b = torch.randn(F, device="cuda")
        Yc = plan.spmm(L.VIEW_COMPACT, M, F=F, bias=b, relu=True).cpu().numpy()
        np.testing.assert_allclose(Yc, np.maximum(Y_ref + b.cpu().numpy(), 0), rtol=1e-4, atol=1e-4)
    dY = torch.randn((A.shape[0], F), device="cuda").to(torch.bfloat16)
    dM_ref = (A.T @ dY.float().cpu().numpy().astype(np.float64))[ref["ulcol"]]
    np.testing.assert_allclose(plan.spmm(L.VIEW_TRANSPOSED, dY).cpu().numpy(), dM_ref, rtol=1e-4, atol=1e-4)
    assert torch.equal(plan.spmm(L.VIEW_LITERAL, Dg), plan.spmm(L.VIEW_LITERAL, Dg))  # no atomics


@pytest.mark.parametrize("B,K,F", [(0, 0, 10), (3, 0, 10), (3, 9, 11), (0, 20, 16), (40, 155, 10), (5, 33, 64)])
def test_operand_builders_round_the_f32_operand(skewed, B, K, F):
    """M_bf16 == round_to_nearest_even(M_f32), bit for bit, for every builder combination."""
    import ctypes as C
    from mrgcn_amd import _lib as L
    plan, A, ref, rng = skewed
    lib = L.load()
    N, R = plan.num_nodes, plan.num_relations
    s = torch.cuda.current_stream().cuda_stream
    S = B if B > 0 else R
    wI = torch.randn((S * N, F), device="cuda")
    comp = torch.randn((R, B), device="cuda") if B > 0 else None
    X = torch.randn((N, K), device="cuda") if K > 0 else None
    W = torch.randn((R, K, F), device="cuda") if K > 0 else None

    def build(bf16, with_I):
        ld = (F + 7) // 8 * 8
        M = torch.zeros((plan.ncols, ld), dtype=torch.bfloat16 if bf16 else torch.float32, device="cuda")
        sfx = "bf16" if bf16 else "f32"
        addend, ldA = 0, 0
        if K > 0:
            if with_I:
                M2 = torch.empty((plan.ncols, ld), device="cuda")
                L.check(lib.mrgcn_rel_transform_fwd_f32(plan.handle, X.data_ptr(), K, K, W.data_ptr(), F,
                                                        M2.data_ptr(), ld, 0, s))
                addend, ldA = M2.data_ptr(), ld
            else:
                L.check(getattr(lib, "mrgcn_rel_transform_fwd_" + sfx)(
                    plan.handle, X.data_ptr(), K, K, W.data_ptr(), F, M.data_ptr(), ld, 1, s))
        if with_I:
            if B > 0:
                L.check(getattr(lib, "mrgcn_basis_mix_fwd_" + sfx)(
                    plan.handle, wI.data_ptr(), comp.data_ptr(), B, F, addend, ldA, M.data_ptr(), ld, s))
            else:
                L.check(getattr(lib, "mrgcn_gather_rows_" + sfx)(
                    plan.handle, wI.data_ptr(), F, addend, ldA, M.data_ptr(), ld, s))
        torch.cuda.synchronize()
        return M

    for with_I in ([True] if K == 0 else [True, False]):
        m32, m16 = build(False, with_I), build(True, with_I)
        assert torch.equal(m16.view(torch.int16), m32.to(torch.bfloat16).view(torch.int16))
        assert float(m16[:, F:].float().abs().max()) == 0.0 if m16.shape[1] > F else True


@pytest.mark.parametrize("name", ["rgcn_small_ft_b3_bias_norm_f32", "rgcn_small_fl_b0_nobias_norm_f32",
                                  "rgcn_smoke_ft_b5_norm_f32", "rgcn_smoke_fl_b5_norm_f32"])
def test_model_with_bf16_operand_close_to_reference(name):
    """Tolerance of the bf16 run: 2e-2 relative to the logits' scale; gradients
    flow through the unchanged fp32 backward."""
    c = util.load_case(name)
    model, dims = util.build_rgcn_from_case(c, "cuda")
    util.load_state_from_case(model, c)
    model = model.cuda()
    model.set_operand_dtype("bf16")
    g, A = util.load_graph(util.graph_of_case(name))
    A = util.coo_tensor(A, str(c["value_mode"]), "cuda")
    X = None if bool(c["meta.featureless"]) else torch.from_numpy(c["X"]).cuda()
    logits = model(X, A)
    ref = c["logits"]
    scale = np.abs(ref).max()
    assert np.abs(logits.detach().cpu().numpy() - ref).max() <= 2e-2 * scale
    assert np.abs(logits.detach().cpu().numpy() - ref).max() > 0  # it really took the bf16 path
    from mrgcn_amd.train import categorical_crossentropy
    loss = categorical_crossentropy(logits, torch.from_numpy(c["labels_idx"]).cuda(),
                                    torch.from_numpy(c["labels_y"]).cuda())
    loss.backward()
    for n, p in model.named_parameters():
        if n != "relations":  # same direction as the reference gradient (cosine), similar size
            gref = c["grad." + n].ravel().astype(np.float64)
            got = util.ref_layout(p.grad, n).cpu().numpy().ravel().astype(np.float64)
            cos = float(got @ gref) / max(np.linalg.norm(got) * np.linalg.norm(gref), 1e-30)
            assert cos > 0.99, (n, cos)
