"""The bf16 pipeline's kernels (BASELINE config 3; SURVEY §8d "dense operands in bf16 with fp32 accumulation"), through
the C ABI.  The reference has no reduced precision (graph.py:93-95 is fp32), so the contracts are stated against the
fp32 kernels and float64 arithmetic ON THE ROUNDED INPUTS:

  * mrgcn_cast_rows_bf16 == torch's round-to-nearest-even, bit for bit, zeros in the row padding;
  * mrgcn_rel_transform_fwd_xbf16 (bf16 rows in, v_mfma_f32_16x16x32_bf16, weights rounded to bf16) == the float64
    product of the rounded rows with the rounded weights to fp32 accuracy (rtol 2e-5), fp32 and bf16 outputs, both
    output orders, K from one to 256 (one to eight k-steps), ragged relation chunks;
  * mrgcn_basis_mix_fwd_abf16 == mrgcn_basis_mix_fwd_f32 / _bf16 on the widened addend, bit for bit;
  * mrgcn_support_rel_transform_bwd_xbf16 == mrgcn_support_rel_transform_bwd_f32 on the widened rows, bit for bit;
  * a model with operand_dtype "bf16" takes the pipeline (mrgcn_amd.stats) and stays within 2e-2 of the fp32 model's
    logits (relative to the largest) and 5e-2 of its gradients in L2 (ReLU's kink switches single terms), eager and
    replayed from a hipGraph."""
import numpy as np
import pytest
import torch

from tests import util
from tests.test_gpu_plan_spmm import _plan_from_coo, _random_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def skewed():
    rng = np.random.default_rng(21)
    N, R, num_rows = 3000, 7, 3000
    rows, cols, vals = _random_graph(rng, num_rows, N, R, 40000, hub_rows=2, hub_len=1500, hub_cols=3)
    plan = _plan_from_coo(rows, cols, vals, num_rows, N, R)
    return plan, util.numpy_plan(rows, cols, vals, num_rows, N, R), rng


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("rows,K", [(1, 1), (5, 7), (1000, 8), (777, 155), (64, 256), (3, 300)])
def test_cast_rows_is_round_to_nearest_even_with_zero_padding(rows, K):
    from mrgcn_amd import _lib as L
    lib = L.load()
    g = torch.Generator("cuda").manual_seed(rows * 1000 + K)
    src = torch.randn((rows, K + 3), device="cuda", generator=g)[:, :K]   # row-strided source
    src[0, 0] = float("inf")
    if rows > 1:
        src[1, K - 1] = 1.00390625   # exactly half way between two bf16 values: ties to even
    ld = (K + 7) // 8 * 8
    dst = torch.full((rows, ld), 7.0, dtype=torch.bfloat16, device="cuda")
    L.check(lib.mrgcn_cast_rows_bf16(src.data_ptr(), src.stride(0), rows, K, dst.data_ptr(), ld, _stream()))
    assert torch.equal(dst[:, :K].view(torch.int16), src.to(torch.bfloat16).contiguous().view(torch.int16))
    assert ld == K or float(dst[:, K:].float().abs().max()) == 0.0


@pytest.mark.parametrize("K,F", [(1, 1), (8, 3), (20, 16), (33, 10), (64, 11), (155, 10), (160, 12), (200, 7), (256, 16)])
def test_transform_on_bf16_rows_equals_the_float64_product_of_the_rounded_inputs(skewed, K, F):
    from mrgcn_amd import _lib as L
    plan, ref, rng = skewed
    lib = L.load()
    N, R = plan.num_nodes, plan.num_relations
    g = torch.Generator("cuda").manual_seed(K * 100 + F)
    X = torch.randn((N, K), device="cuda", generator=g)
    W = torch.randn((R, K, F), device="cuda", generator=g) / np.sqrt(K)
    ldX = (K + 7) // 8 * 8
    Xb = torch.empty((N, ldX), dtype=torch.bfloat16, device="cuda")
    L.check(lib.mrgcn_cast_rows_bf16(X.data_ptr(), K, N, K, Xb.data_ptr(), ldX, _stream()))
    X64 = Xb[:, :K].float().cpu().numpy().astype(np.float64)
    W64 = W.to(torch.bfloat16).float().cpu().numpy().astype(np.float64)
    ulcol, mpos = ref["ulcol"].astype(np.int64), ref["mpos"].astype(np.int64)
    want = np.einsum("ck,ckf->cf", X64[ulcol % N], W64[ulcol // N])           # compact order
    for ld in sorted({F, (F + 3) // 4 * 4, 16}):
        assert lib.mrgcn_rel_transform_xbf16_supported(plan.handle, K, F, ldX, ld)
        for order in (0, 1):
            nrows = plan.nop if order else plan.ncols
            for out_bf16 in (0, 1):
                out = torch.full((nrows, ld), 3.0, dtype=torch.bfloat16 if out_bf16 else torch.float32, device="cuda")
                L.check(lib.mrgcn_rel_transform_fwd_xbf16(plan.handle, Xb.data_ptr(), ldX, K, W.data_ptr(), F,
                                                          out.data_ptr(), ld, order, out_bf16, _stream()))
                got = out.float().cpu().numpy()
                got = got[mpos] if order else got
                if out_bf16:   # one more rounding at the store
                    np.testing.assert_allclose(got[:, :F], want, rtol=2 ** -8, atol=1e-6)
                else:
                    np.testing.assert_allclose(got[:, :F], want, rtol=2e-5, atol=2e-5)
                if ld > F and ld % 4 == 0:   # whole pieces are written: zeros past F
                    assert float(np.abs(got[:, F:]).max()) == 0.0
    assert not lib.mrgcn_rel_transform_xbf16_supported(plan.handle, 257, F, 264, 16)
    assert not lib.mrgcn_rel_transform_xbf16_supported(plan.handle, K, 17, ldX, 20)


@pytest.mark.parametrize("B,F", [(40, 10), (3, 8), (16, 11), (64, 16), (5, 4)])
def test_mix_forward_with_a_bf16_addend_equals_the_fp32_addend_form(skewed, B, F):
    from mrgcn_amd import _lib as L
    plan, ref, rng = skewed
    lib = L.load()
    N, R = plan.num_nodes, plan.num_relations
    g = torch.Generator("cuda").manual_seed(B * 100 + F)
    V = torch.randn((N, B, F), device="cuda", generator=g)
    comp = torch.randn((R, B), device="cuda", generator=g)
    addb = torch.randn((plan.ncols, 16), device="cuda", generator=g).to(torch.bfloat16)
    addb[:, F:] = 0
    add32 = addb.float().contiguous()
    for out_bf16, dt, fn in ((0, torch.float32, lib.mrgcn_basis_mix_fwd_f32), (1, torch.bfloat16, lib.mrgcn_basis_mix_fwd_bf16)):
        ld = (F + 3) // 4 * 4
        want = torch.zeros((plan.nop, ld), dtype=dt, device="cuda")
        got = torch.zeros((plan.nop, ld), dtype=dt, device="cuda")
        L.check(fn(plan.handle, V.data_ptr(), comp.data_ptr(), B, F, add32.data_ptr(), 16, want.data_ptr(), ld, _stream()))
        L.check(lib.mrgcn_basis_mix_fwd_abf16(plan.handle, V.data_ptr(), comp.data_ptr(), B, F, addb.data_ptr(), 16,
                                              got.data_ptr(), ld, out_bf16, _stream()))
        assert torch.equal(got.view(torch.int16 if out_bf16 else torch.int32),
                           want.view(torch.int16 if out_bf16 else torch.int32))


@pytest.mark.parametrize("K,F", [(155, 10), (40, 16), (8, 3)])
def test_support_dw_on_bf16_rows_equals_the_fp32_kernel_on_the_widened_rows(skewed, K, F):
    from mrgcn_amd import _lib as L
    plan, ref, rng = skewed
    lib = L.load()
    N, R = plan.num_nodes, plan.num_relations
    g = torch.Generator("cuda").manual_seed(K + F)
    flags = (torch.rand(plan.num_rows, device="cuda", generator=g) < 0.2).to(torch.uint8)
    sup = plan.support_for(flags)
    X = torch.randn((N, K), device="cuda", generator=g).to(torch.bfloat16)
    ldX = (K + 7) // 8 * 8
    Xb = torch.zeros((N, ldX), dtype=torch.bfloat16, device="cuda")
    Xb[:, :K] = X
    X32 = X.float().contiguous()
    W = torch.randn((R, K, F), device="cuda", generator=g)
    ld = (F + 3) // 4 * 4
    dM = torch.randn((max(sup.L, 1), ld), device="cuda", generator=g)
    nws = int(lib.mrgcn_support_rel_transform_bwd_workspace(sup.handle, K, F, 1, 1))
    assert nws > 0
    outs = []
    for bf in (False, True):
        ws = torch.empty(nws, device="cuda")
        dX = torch.empty((N, K), device="cuda")
        dW = torch.empty((R, K, F), device="cuda")
        if bf:
            L.check(lib.mrgcn_support_rel_transform_bwd_xbf16(sup.handle, dM.data_ptr(), ld, Xb.data_ptr(), ldX, K,
                                                              W.data_ptr(), F, dX.data_ptr(), K, dW.data_ptr(),
                                                              ws.data_ptr(), nws, _stream()))
        else:
            L.check(lib.mrgcn_support_rel_transform_bwd_f32(sup.handle, dM.data_ptr(), ld, X32.data_ptr(), K, K,
                                                            W.data_ptr(), F, dX.data_ptr(), K, dW.data_ptr(),
                                                            ws.data_ptr(), nws, 0, _stream()))
        outs.append((dX, dW))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1].abs().max()) > 0


def _am_like(scale, seed=3):
    from mrgcn_amd import synth
    from mrgcn_amd.models.rgcn import RGCN
    g = synth.make_graph("am", seed=seed, scale=scale)
    N, R = g.num_nodes, g.num_relations
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals), (N, R * N)).cuda()
    torch.manual_seed(seed)
    model = RGCN([(155, 10, "mrgcn", torch.nn.ReLU()), (10, 11, "mrgcn", None)], R, N, 40, 0.0, False, True, False).cuda()
    X = torch.randn((N, 155), device="cuda", generator=torch.Generator("cuda").manual_seed(seed))
    idx, y = synth.make_labels("am", N, seed=0, scale=scale)
    return model, A, X, torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()


def test_model_in_bf16_takes_the_pipeline_and_stays_within_tolerance_of_fp32_eager_and_replayed():
    """AM/16 with the AM model: fp32 epoch against the bf16 pipeline's — logits, loss, every gradient (2e-2 of the
    tensor's largest element); the input's bf16 copy is made once for a constant X and again when X changes in place; a
    replayed hipGraph step equals the eager one bit for bit."""
    import mrgcn_amd
    from mrgcn_amd import functional as Fn
    from mrgcn_amd.train import ClipAdam, GraphedTrainStep, categorical_crossentropy, train_step
    model, A, X, idx, y = _am_like(1 / 16)
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def grads(dtype):
        model.load_state_dict(init)
        model.set_operand_dtype(dtype)
        model.zero_grad(set_to_none=True)
        logits = model(X, A)
        loss = categorical_crossentropy(logits, idx, y)
        loss.backward()
        return logits.detach().clone(), float(loss.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters()}

    l32, loss32, g32 = grads("f32")
    mrgcn_amd.reset_stats()
    l16, loss16, g16 = grads("bf16")
    st = mrgcn_amd.stats()
    assert st.get("bf16.xform_xbf16") == 1 and st.get("bf16.dw_xbf16") == 1 and st.get("bf16.x_cast") == 1, st
    scale = float(l32.abs().max())
    err = float((l16 - l32).abs().max())
    assert 0 < err <= 2e-2 * scale, (err, scale)
    assert abs(loss16 - loss32) <= 2e-2 * abs(loss32)
    for n in g32:   # (the basis coefficients' gradients are sums over every column of a relation that cancel to a
        # small remainder: compared by direction and size, like the round-5 bf16 operand test)
        a, b = g16[n].double().flatten(), g32[n].double().flatten()
        if n.endswith("_comp"):
            cos = float(a @ b) / max(float(a.norm() * b.norm()), 1e-300)
            assert cos > 0.99 and abs(float(a.norm() / b.norm()) - 1) < 5e-2, (n, cos)
        else:   # (a hidden unit at the ReLU's kink switches a whole gradient term on or off: L2 and outlier share)
            rel = float((a - b).norm() / b.norm())
            out = float(((a - b).abs() > 2e-2 * float(b.abs().max())).double().mean())
            assert rel <= 5e-2 and (out <= 0.10 or a.numel() < 1000), (n, rel, out)
    # a constant input is converted once; an in-place change of X is seen
    mrgcn_amd.reset_stats()
    with torch.no_grad():
        a = model(X, A)
        b = model(X, A)
        assert mrgcn_amd.stats().get("bf16.x_cast") is None and mrgcn_amd.stats().get("bf16.x_cached") == 2
        assert torch.equal(a, b)
        X.mul_(2.0)
        c = model(X, A)
        assert mrgcn_amd.stats().get("bf16.x_cast") == 1
        assert not torch.equal(a, c)
        X.mul_(0.5)
    # MRGCN_BF16_PIPELINE=0 is the round-5 form (only M in bf16): still within tolerance of the pipeline
    prev, Fn._BF16_PIPELINE = Fn._BF16_PIPELINE, False
    try:
        with torch.no_grad():
            d = model(X, A)
    finally:
        Fn._BF16_PIPELINE = prev
    assert float((d - l16).abs().max()) <= 2e-2 * scale
    # eager steps against replayed ones
    model.load_state_dict(init)
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0, capturable=True)
    eager = [float(train_step(model, lambda: model(X, A), idx, y, opt)) for _ in range(3)]
    sd_e = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.load_state_dict(init)
    opt2 = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0, capturable=True)
    step = GraphedTrainStep(model, lambda: model(X, A), idx, y, opt2, warmup=1)
    replayed = [float(step()) for _ in range(2)]
    np.testing.assert_allclose(replayed, eager[1:], rtol=1e-6)
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd_e[k]), k
    model.set_operand_dtype("f32")
