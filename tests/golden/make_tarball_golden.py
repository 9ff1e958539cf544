#!/usr/bin/env python
"""Dataset tarball fixture, AUTHORING CONTAINER ONLY: written by the reference's own
`mrgcn/data/io/tarball.py::Tarball.store` (:107-135) under the names `mkdataset.py:121-122` uses
(A, F, Y, data, sample_map, class_map), with the 50-node golden graph as A.  The fixture is the
data file the reference produces; nothing of the reference's source is stored.
    python tests/golden/make_tarball_golden.py"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402


def main():
    ref = mg.import_reference()
    if not hasattr(sp, "csr"):  # scipy >= 1.14 dropped the sp.csr alias the reference's type checks use
        import scipy.sparse._csr as _csr
        sp.csr = _csr
    from mrgcn.data.io.tarball import Tarball
    g = np.load(os.path.join(HERE, "graph_small.npz"))
    N, P = int(g["num_nodes"]), int(g["num_pred"])
    A = mg.reference_adjacency(ref, g["triples"], N, P)
    rng = np.random.default_rng(3)
    C = 4
    splits = {}
    perm = rng.permutation(N)
    for name, sl in (("train", perm[:20]), ("valid", perm[20:28]), ("test", perm[28:40])):
        y = rng.integers(0, C, len(sl))
        splits[name] = (np.sort(sl), y[np.argsort(sl)])
    Y = {k: sp.csr_matrix((np.ones(len(i), dtype=np.int8), (i, y)), shape=(N, C), dtype=np.int8)
         for k, (i, y) in splits.items()}  # mk_target_matrices, node_classification.py:351-383
    num_idx = np.sort(rng.choice(N, 30, replace=False))
    F = {"xsd.numeric": [[rng.standard_normal((30, 4)).astype(np.float32), num_idx, np.ones(30, dtype=int)]],
         "xsd.boolean": [[rng.integers(0, 2, (12, 1)).astype(np.float32), np.sort(rng.choice(N, 12, replace=False)),
                          np.ones(12, dtype=int)]]}
    data = {"train": g["triples"][:80], "valid": g["triples"][80:100], "test": g["triples"][100:]}
    sample_map = {k: np.array([f"http://example.org/node/{i}" for i in v[0]]) for k, v in splits.items()}
    class_map = [f"http://example.org/class/{c}" for c in range(C)]
    path = os.path.join(HERE, "dataset_small.tar")
    if os.path.exists(path):
        os.remove(path)
    with Tarball(path, "w") as tb:
        tb.store([A, F, Y, data, sample_map, class_map], names=["A", "F", "Y", "data", "sample_map", "class_map"])
    # what the reference reads back from it (expected values for the reader test)
    with Tarball(path, "r") as tb:
        back = {k: tb.get(k) for k in ("A", "F", "Y", "data", "sample_map", "class_map")}
    exp = {"A.data": back["A"].data, "A.indices": back["A"].indices, "A.indptr": back["A"].indptr,
           "A.shape": np.asarray(back["A"].shape), "class_map": np.asarray(back["class_map"]),
           "F.numeric.enc": back["F"]["xsd.numeric"][0][0], "F.numeric.idx": back["F"]["xsd.numeric"][0][1],
           "F.boolean.enc": back["F"]["xsd.boolean"][0][0]}
    for k, y in back["Y"].items():
        exp[f"Y.{k}.indices"], exp[f"Y.{k}.indptr"], exp[f"Y.{k}.dtype"] = y.indices, y.indptr, np.asarray(str(y.dtype))
    for k in ("train", "valid", "test"):
        exp[f"data.{k}"] = back["data"][k]
        exp[f"sample_map.{k}"] = back["sample_map"][k]
    np.savez_compressed(os.path.join(HERE, "dataset_small_expected.npz"), **exp)
    import tarfile
    print(sorted(tarfile.open(path).getnames()))
    print({k: type(v).__name__ for k, v in back.items()}, os.path.getsize(path))


if __name__ == "__main__":
    main()
