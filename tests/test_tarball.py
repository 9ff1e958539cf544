"""Dataset archive (reference mrgcn/data/io/tarball.py): the reader against a
tarball written by the reference's own Tarball.store (tests/golden/make_tarball_golden.py) and the
values the reference reads back from it; writer round trip; CSR -> graph plan ingestion on the GPU."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests import util

HERE = os.path.dirname(__file__)
TAR = os.path.join(HERE, "golden", "dataset_small.tar")
EXP = os.path.join(HERE, "golden", "dataset_small_expected.npz")


def test_reader_matches_what_the_reference_reads_back():
    from mrgcn_amd.data.io.tarball import Tarball
    e = np.load(EXP, allow_pickle=True)
    with Tarball(TAR, "r") as tb:
        assert sorted(tb.list_members()) == ["A", "F", "Y", "class_map", "data", "sample_map"] and len(tb) == 6
        A, F, Y = tb.get("A"), tb.get("F"), tb.get("Y")
        data, sample_map, class_map = tb.get("data"), tb.get("sample_map"), tb.get("class_map")
    assert isinstance(A, sp.csr_matrix) and A.dtype == np.float32 and list(A.shape) == list(e["A.shape"])
    for k in ("data", "indices", "indptr"):
        assert np.array_equal(getattr(A, k), e["A." + k])
    assert np.array_equal(F["xsd.numeric"][0][0], e["F.numeric.enc"])
    assert np.array_equal(F["xsd.numeric"][0][1], e["F.numeric.idx"])
    assert np.array_equal(F["xsd.boolean"][0][0], e["F.boolean.enc"])
    for k in ("train", "valid", "test"):
        assert np.array_equal(Y[k].indices, e[f"Y.{k}.indices"]) and np.array_equal(Y[k].indptr, e[f"Y.{k}.indptr"])
        assert str(Y[k].dtype) == str(e[f"Y.{k}.dtype"])  # CSR members come back as float32 (:151-157)
        assert np.array_equal(data[k], e[f"data.{k}"])
        assert np.array_equal(sample_map[k], e[f"sample_map.{k}"])
    assert list(class_map) == list(e["class_map"])


def test_writer_round_trip(tmp_path):
    from mrgcn_amd.data.io.tarball import Tarball
    A = sp.random(7, 21, density=0.3, format="csr", dtype=np.float32, random_state=1)
    objs = [A, {"a": np.arange(3), "b": {"c": np.ones((2, 2)), "d": "text"}, "e": {}}, [np.arange(2), "x", A],
            torch.arange(4), torch.sparse_coo_tensor(torch.tensor([[0, 1], [1, 0]]), torch.tensor([1.0, 2.0]), (2, 2)),
            {"k": 1}]
    p = str(tmp_path / "t.tar")
    with Tarball(p, "w") as tb:
        tb.store(objs, names=["A", "F", "L", "T", "S", "P"])
    with Tarball(p, "r") as tb:
        assert (tb.get("A") != A).nnz == 0
        F = tb.get("F")
        assert np.array_equal(F["a"], np.arange(3)) and np.array_equal(F["b"]["c"], np.ones((2, 2)))
        assert F["b"]["d"] == "text" and F["e"] == {}
        L = tb.get("L")
        assert np.array_equal(L[0], np.arange(2)) and L[1] == "x" and (L[2] != A).nnz == 0
        assert torch.equal(tb.get("T"), torch.arange(4))
        assert torch.equal(tb.get("S").to_dense(), torch.tensor([[0.0, 1.0], [2.0, 0.0]]))
        assert tb.get("P") == {"k": 1}


def test_labels_of():
    from mrgcn_amd.data.dataset import labels_of, load_tarball
    d = load_tarball(TAR)
    idx, y = labels_of(d["Y"]["train"])
    assert len(idx) == 20 and idx.dtype == np.int64 and y.max() < 4 and np.all(np.diff(idx) >= 0)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["ref_int8", "norm_f32"])
def test_plan_from_csr_equals_plan_from_coo(mode):
    """Every plan array bit-identical whether A arrives as the reference's COO tensor or as the
    archive's CSR arrays; the int8 boundary cast is applied on the device."""
    from mrgcn_amd import _lib as L
    from mrgcn_amd.data.dataset import load_tarball
    from mrgcn_amd.plan import GraphPlan
    A = load_tarball(TAR)["A"]
    N = A.shape[0]
    R = A.shape[1] // N
    p_csr = GraphPlan.from_csr(A, N, R, value_mode=mode)
    p_coo = GraphPlan(util.coo_tensor(A, mode, "cuda"), N, R)
    assert (p_csr.nnz, p_csr.ncols) == (p_coo.nnz, p_coo.ncols)
    for arr in range(17):
        assert np.array_equal(p_csr.export(arr), p_coo.export(arr)), arr
    # rectangular / empty input
    E = sp.csr_matrix((3, 2 * 5), dtype=np.float32)
    assert GraphPlan.from_csr(E, 5, 2).nnz == 0


@pytest.mark.gpu
def test_model_on_adjacency_handle_from_tarball():
    """tarball -> CSR -> plan -> MRGCN(FullBatch) logits equal those on the COO tensor path."""
    from mrgcn_amd.data.batch import FullBatch
    from mrgcn_amd.data.dataset import full_batch, load_tarball
    from mrgcn_amd.models.mrgcn import MRGCN
    d = load_tarball(TAR)
    A = d["A"]
    N = A.shape[0]
    R = A.shape[1] // N
    torch.manual_seed(0)
    modules = [(0, 8, "mrgcn", torch.nn.ReLU()), (8, 4, "mrgcn", None)]
    model = MRGCN(modules, [], R, N, num_bases=3, featureless=True, gcn_gpu_acceleration=True)
    b1 = full_batch(A, value_mode="ref_int8")
    ref = FullBatch(A, [np.empty((N, 0))], np.arange(N))
    ref.as_tensors_()
    ref.to(model.devices)
    out_ref = model(ref)
    b1.X[0] = torch.from_numpy(b1.X[0])
    out = model(b1)
    # (two plans of the same adjacency: built from the CSR without a layout hint and from the COO with the model's —
    # the operand order, hence the order in which a row's terms are added, may differ)
    torch.testing.assert_close(out, out_ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("gname", ["graph_small", "graph_smoke"])
def test_adjacency_from_triples_matches_reference_builder(gname):
    """The vectorised builder against the CSR the reference's graph_structure code produced
    (golden): same pattern, same float32 values."""
    from mrgcn_amd.data.graph_structure import adjacency_from_triples
    g = np.load(os.path.join(HERE, "golden", gname + ".npz"))
    N, P = int(g["num_nodes"]), int(g["num_pred"])
    A = adjacency_from_triples(g["triples"], N, P)
    ref = sp.csr_matrix((g["csr_data"], g["csr_indices"], g["csr_indptr"]), shape=tuple(g["shape"]))
    ref.sort_indices()
    assert A.shape == ref.shape and A.dtype == np.float32
    assert np.array_equal(A.indptr, ref.indptr) and np.array_equal(A.indices, ref.indices)
    assert np.array_equal(A.data, ref.data)


def test_renumbering_a_dataset_archive_keeps_every_reference_consistent():
    """data.reorder.renumber_dataset on the golden archive: adjacency entries, label rows, feature
    node indices and triples all follow the permutation chosen from the training labels."""
    import numpy as np
    from mrgcn_amd.data import reorder
    from mrgcn_amd.data.dataset import labels_of, load_tarball
    d = load_tarball(TAR)
    A = d["A"].tocsr()
    N = A.shape[0]
    R = A.shape[1] // N
    idx, _ = labels_of(d["Y"]["train"])
    coo = A.tocoo()
    order, inv = reorder.label_reach_order(coo.row, coo.col, N, R, idx, hops=2)
    d2 = reorder.renumber_dataset(d, order, inv)
    A2 = d2["A"]
    assert A2.shape == A.shape and A2.nnz == A.nnz
    c2 = A2.tocoo()
    back = dict(zip(zip(order[c2.row].tolist(), ((c2.col // N) * N + order[c2.col % N]).tolist()), c2.data.tolist()))
    assert back == dict(zip(zip(coo.row.tolist(), coo.col.tolist()), coo.data.tolist()))
    for split in d["Y"]:
        assert (d2["Y"][split].toarray() == d["Y"][split].toarray()[order]).all()
    i2, t2 = labels_of(d2["Y"]["train"])
    assert sorted(order[i2].tolist()) == sorted(idx.tolist())
    assert sorted(i2.tolist()) == sorted(inv[idx].tolist())
    for dt in d["F"]:
        for (e, n, s), (e2, n2, s2) in zip(d["F"][dt], d2["F"][dt]):
            assert (e2 == e).all() and (order[n2] == n).all() and (s2 == s).all()
    for split in d["data"]:
        t, t2_ = np.asarray(d["data"][split]), d2["data"][split]
        assert (order[t2_[:, 0]] == t[:, 0]).all() and (t2_[:, 1] == t[:, 1]).all() and (order[t2_[:, 2]] == t[:, 2]).all()
