"""Layer-0 feature transform (graph.py:93-94: `torch.einsum('rb,bio->rio', comp, W); torch.mm(X, W_r)` per relation)
with the relations that own the most columns taken NODE-TILE-MAJOR (csrc/xform_mfma.hip: k_xform_hot_fwd — the X tile
and those relations' weight tiles in LDS, an input row read once for all of them) and the others relation-major.

Contract: bit-equal to the relation-major kernel on every column (same MFMA, same k order), both output orders,
f32 and bf16 operand rows, ragged shapes (K not a multiple of 4 or 16, a last tile of fewer nodes, nodes without
columns, relations without columns, a hub node with every relation) — and equal to the float64 product."""
import numpy as np
import pytest
import torch

from tests import util
from tests.test_gpu_plan_spmm import _plan_from_coo

pytestmark = pytest.mark.gpu


def _graph(rng, N, R, nnz, skew=1.3, hub=True):
    """relations drawn with a power law (a few own most columns), the last relation = identity block"""
    w = 1.0 / np.arange(1, R) ** skew
    rel = rng.choice(R - 1, size=nnz, p=w / w.sum())
    src = rng.integers(0, N, nnz)
    dst = rng.integers(0, N, nnz)
    rows = np.concatenate([dst, np.arange(N)])
    cols = np.concatenate([rel * N + src, (R - 1) * N + np.arange(N)])
    if hub:  # one source node under every relation, one node without any column but its own
        rows = np.concatenate([rows, rng.integers(0, N, R)])
        cols = np.concatenate([cols, np.arange(R) * N + 7])
    key = np.unique(rows.astype(np.int64) * (R * N) + cols)
    rows, cols = key // (R * N), key % (R * N)
    vals = rng.random(len(rows)).astype(np.float32) + 0.1
    return rows, cols, vals


def _transform(plan, X, W, F, ld, order, bf16, hot, **cfg):
    from mrgcn_amd import _lib as L
    lib = L.load()
    K = X.shape[1]
    old = L.set_config(xform_hot=hot, **cfg)
    try:
        out = torch.full((plan.nop if order else plan.ncols, ld), float("nan"),
                         dtype=torch.bfloat16 if bf16 else torch.float32, device="cuda")
        fn = lib.mrgcn_rel_transform_fwd_bf16 if bf16 else lib.mrgcn_rel_transform_fwd_f32
        L.check(fn(plan.handle, X.data_ptr(), X.stride(0), K, W.data_ptr(), F, out.data_ptr(), ld, order,
                   torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    finally:
        L.set_config(**old)
    return out


@pytest.mark.parametrize("N,R,K,F,ld", [
    (3000, 12, 155, 10, 10),     # the AM layer shape, rows packed
    (3000, 12, 155, 11, 12),     # padded rows: zeros behind F
    (1000, 40, 64, 16, 16),      # more relations than fit: ranks behind the hot ones stay relation-major
    (2051, 5, 100, 3, 4),        # a last tile of 3 nodes
    (700, 3, 256, 8, 8),         # the widest input the matrix-core transform takes
    (515, 20, 77, 1, 1),
])
def test_hot_relations_node_tile_major_equal_the_relation_major_kernel(N, R, K, F, ld):
    rng = np.random.default_rng(N + K)
    rows, cols, vals = _graph(rng, N, R, 6 * N)
    plan = _plan_from_coo(rows, cols, vals, N, N, R)
    X = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).cuda()
    W = torch.from_numpy(rng.standard_normal((R, K, F)).astype(np.float32)).cuda()
    ref = util.numpy_plan(rows, cols, vals, N, N, R)
    ulcol = ref["ulcol"]
    want = np.einsum("ck,ckf->cf", X.cpu().numpy().astype(np.float64)[ulcol % N],
                     W.cpu().numpy().astype(np.float64)[ulcol // N])
    for order in (0, 1):
        for bf16 in (False, True):
            base = _transform(plan, X, W, F, ld, order, bf16, hot=0)
            for tile, nh, depth in [(64, 0, 2), (16, 2, 1), (96, 3, 3), (32, 16, 2)]:
                got = _transform(plan, X, W, F, ld, order, bf16, hot=2, xform_hot_tile=tile, xform_hot_nh=nh,
                                 xform_hot_depth=depth)
                assert torch.equal(got.view(torch.int16 if bf16 else torch.int32),
                                   base.view(torch.int16 if bf16 else torch.int32)), (order, bf16, tile, nh)
            if not bf16:
                rowsel = torch.from_numpy(ref["mpos"]).long().cuda() if order else slice(None)
                np.testing.assert_allclose(base[rowsel][:, :F].cpu().numpy(), want, rtol=2e-4, atol=2e-4)
                if ld > F:
                    assert float(base[rowsel][:, F:].abs().max()) == 0.0


def test_hot_path_is_the_default_on_a_large_graph_and_counts_every_column():
    """N >= 65 536 switches the path on by itself (xform_hot = 1); every column is written exactly once: the output is
    pre-filled with NaN and must come back equal to the relation-major kernel's."""
    from mrgcn_amd import _lib as L
    rng = np.random.default_rng(5)
    N, R, K, F = 70000, 30, 155, 10
    rows, cols, vals = _graph(rng, N, R, 4 * N)
    plan = _plan_from_coo(rows, cols, vals, N, N, R)
    X = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).cuda()
    W = torch.from_numpy(rng.standard_normal((R, K, F)).astype(np.float32)).cuda()
    assert L.config()["xform_hot"] == 1
    base = _transform(plan, X, W, F, F, 0, False, hot=0)
    got = _transform(plan, X, W, F, F, 0, False, hot=1)
    assert torch.equal(got.view(torch.int32), base.view(torch.int32))
    assert not bool(torch.isnan(got).any())
