# EXPERIMENT RECORD (round 5): the tests that went with tools/lab/xform_band_experiment.hip; not collected by pytest
# (the entry points they exercise took the band-major path only while that kernel was in the library).
"""The band-major, XCD-affine transform of wide input rows (csrc/xform_band.hip; graph.py:93-94 for the touched columns)
against numpy float64 on small graphs whose plans are forced to carry band tiles (tiny bands: many bands per XCD, partial
tiles, single-column groups), in both output orders, f32 and bf16 operands, ragged K / F."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiled_plan():
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphPlan
    old = {k: os.environ.get(k) for k in ("MRGCN_TILE_MIN_COLS", "MRGCN_TILE_BAND")}
    os.environ["MRGCN_TILE_MIN_COLS"], os.environ["MRGCN_TILE_BAND"] = "1", "96"
    try:
        rng = np.random.default_rng(5)
        N, R = 3001, 9
        n = 40000
        rows = np.concatenate([rng.integers(0, N, n), np.arange(N)])
        cols = np.concatenate([(rng.integers(0, R - 1, n) ** 2 // (R - 1)) * N + rng.integers(0, N, n) ** 2 // N,
                               (R - 1) * N + np.arange(N)])
        key = np.unique(rows.astype(np.int64) * (R * N) + cols)
        rows, cols = key // (R * N), key % (R * N)
        vals = rng.uniform(0.1, 1.0, len(rows)).astype(np.float32)
        A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cols])), torch.from_numpy(vals), (N, R * N)).cuda()
        plan = GraphPlan(A, N, R)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return plan, N, R, plan.export(L.ARR_UNODE).astype(np.int64), plan.export(L.ARR_UREL).astype(np.int64), \
        plan.export(L.ARR_MPOS).astype(np.int64)


@pytest.mark.parametrize("K,F,ld", [(155, 10, 12), (33, 1, 4), (64, 16, 16), (200, 11, 12), (256, 7, 8), (100, 10, 10)])
@pytest.mark.parametrize("order", [0, 1])
def test_band_transform_vs_numpy(tiled_plan, K, F, ld, order):
    from mrgcn_amd import _lib as L
    plan, N, R, unode, urel, mpos = tiled_plan
    lib = L.load()
    rng = np.random.default_rng(K * 31 + F)
    ldx = K + 3
    Xf = rng.standard_normal((N, ldx)).astype(np.float32)
    W = rng.standard_normal((R, K, F)).astype(np.float32)
    X = torch.from_numpy(Xf).cuda()
    Wd = torch.from_numpy(W).cuda()
    Out = torch.full((plan.ncols, ld), float("nan"), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    L.check(lib.mrgcn_rel_transform_fwd_f32(plan.handle, X.data_ptr(), ldx, K, Wd.data_ptr(), F, Out.data_ptr(), ld,
                                            order, s), "mrgcn_rel_transform_fwd_f32")
    got = Out.cpu().numpy()
    ref = np.einsum("ck,ckf->cf", Xf[unode, :K].astype(np.float64), W[urel].astype(np.float64))
    pos = mpos if order else np.arange(plan.ncols)
    np.testing.assert_allclose(got[pos, :F], ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    assert not got[:, F:].any()      # the padding of a row is written as zeros


def test_band_transform_bf16_operand(tiled_plan):
    from mrgcn_amd import _lib as L
    plan, N, R, unode, urel, mpos = tiled_plan
    lib = L.load()
    rng = np.random.default_rng(1)
    K, F, ld = 155, 10, 12
    Xf = rng.standard_normal((N, K)).astype(np.float32)
    W = rng.standard_normal((R, K, F)).astype(np.float32)
    X, Wd = torch.from_numpy(Xf).cuda(), torch.from_numpy(W).cuda()
    Out = torch.zeros((plan.ncols, ld), dtype=torch.bfloat16, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    L.check(lib.mrgcn_rel_transform_fwd_bf16(plan.handle, X.data_ptr(), K, K, Wd.data_ptr(), F, Out.data_ptr(), ld, 1, s),
            "mrgcn_rel_transform_fwd_bf16")
    ref = np.einsum("ck,ckf->cf", Xf[unode].astype(np.float64), W[urel].astype(np.float64))
    got = Out.float().cpu().numpy()[mpos, :F]
    assert np.abs(got - ref).max() <= 1e-2 * np.abs(ref).max()


def test_plan_without_tiles_takes_the_relation_major_kernel(tiled_plan):
    """the same product on a plan built without band tiles (the default for a graph this small): equal results"""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.plan import GraphPlan
    plan, N, R, unode, urel, mpos = tiled_plan
    lib = L.load()
    rng = np.random.default_rng(2)
    K, F, ld = 155, 10, 12
    X = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).cuda()
    W = torch.from_numpy(rng.standard_normal((R, K, F)).astype(np.float32)).cuda()
    s = torch.cuda.current_stream().cuda_stream
    lcol = plan.export(L.ARR_ULCOL).astype(np.int64)
    # a second plan over the same adjacency (same compact numbering), no tiles
    rows_t = torch.from_numpy(plan.export(L.ARR_ROWIDX).astype(np.int64))
    cols_t = torch.from_numpy(plan.export(L.ARR_LCOL).astype(np.int64))
    vals_t = torch.from_numpy(plan.export(L.ARR_VAL))
    A = torch.sparse_coo_tensor(torch.stack([rows_t, cols_t]), vals_t, (N, R * N)).cuda()
    plain = GraphPlan(A, N, R)
    assert plain.ncols == plan.ncols and np.array_equal(plain.export(L.ARR_ULCOL).astype(np.int64), lcol)
    outs = []
    for pl in (plan, plain):
        Out = torch.empty((pl.ncols, ld), device="cuda")
        L.check(lib.mrgcn_rel_transform_fwd_f32(pl.handle, X.data_ptr(), K, K, W.data_ptr(), F, Out.data_ptr(), ld, 0, s),
                "mrgcn_rel_transform_fwd_f32")
        outs.append(Out)
    assert torch.equal(outs[0], outs[1])     # same slot <-> k assignment, same MFMA chain: the same bits
