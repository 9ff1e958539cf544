#!/usr/bin/env python
"""Golden vectors for `Batch.pad_` / `Batch.to_dense_` (AUTHORING CONTAINER ONLY).

Drives the reference's `FullBatch.pad_(pad_symbols=...)` and `to_dense_()`
(mrgcn/data/batch.py:25-68 with mrgcn/data/utils.py:109-152) on a feature list with one
token-sequence encoding set (object array of int arrays), one set of CSR members (object array of
scipy matrices, the WKT / temporal-CNN form) and one fixed-width set, and stores inputs + outputs in
`pad_batch.npz`.  Only data is stored.

    python tests/golden/make_pad_goldens.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_goldens import REF, _stub_rdflib  # noqa: E402


def inputs(seed=0):
    rng = np.random.default_rng(seed)
    n_tok, n_mat = 7, 5
    tok_len = np.array([3, 9, 1, 6, 12, 2, 4])
    toks = np.empty(n_tok, dtype=object)
    for i, L in enumerate(tok_len):
        toks[i] = rng.integers(1, 30000, L).astype(np.int64)
    toks[3][2] = -1  # the reference turns a token id of -1 into the pad symbol
    mat_w = np.array([4, 11, 7, 2, 9])
    mats = np.empty(n_mat, dtype=object)
    for i, w in enumerate(mat_w):
        mats[i] = sp.random(3, w, density=0.5, format="csr", dtype=np.float32, random_state=int(rng.integers(1 << 30)))
    num = rng.standard_normal((6, 4)).astype(np.float32)
    return toks, tok_len, mats, mat_w, num


def feature_list(toks, tok_len, mats, mat_w, num):
    return [np.zeros((20, 0), dtype=np.float32),
            ["xsd.string", [[toks, np.arange(len(toks)), tok_len.copy()]], False],
            ["ogc.wktLiteral", [[mats, np.arange(len(mats)), mat_w.copy()]], False],
            ["xsd.numeric", [[num, np.arange(len(num)), np.full(len(num), 4)]], False]]


def main():
    _stub_rdflib()
    sys.path.insert(0, REF)
    from mrgcn.data.batch import FullBatch
    toks, tok_len, mats, mat_w, num = inputs()
    out = {"tok_flat": np.concatenate(list(toks)), "tok_len": tok_len, "mat_w": mat_w, "num": num}
    for i, m in enumerate(mats):
        out[f"mat{i}.data"], out[f"mat{i}.indices"], out[f"mat{i}.indptr"] = m.data, m.indices, m.indptr
        out[f"mat{i}.shape"] = np.array(m.shape)
    for name, pads, seq_override in (("default", {}, None), ("pad101", {"xsd.string": 101}, None),
                                     ("wide", {"xsd.string": 7}, 16)):
        X = feature_list(*inputs())
        if seq_override is not None:  # seq_length larger than every member: the width follows it
            X[1][1][0][2] = np.full(len(toks), seq_override)
            X[2][1][0][2] = np.full(len(mats), seq_override)
        b = FullBatch(None, X, np.arange(20))
        b.pad_(pad_symbols=pads)
        out[f"{name}.tok_padded"] = np.asarray(b.X[1][1][0][0])
        padded = b.X[2][1][0][0]
        out[f"{name}.mat_shapes"] = np.array([m.shape for m in padded])
        b.to_dense_()
        out[f"{name}.mat_dense"] = np.asarray(b.X[2][1][0][0])
        out[f"{name}.num"] = np.asarray(b.X[3][1][0][0])
    np.savez_compressed(os.path.join(HERE, "pad_batch.npz"), **out)
    print({k: getattr(v, "shape", None) for k, v in out.items() if "." in k and not k.startswith("mat")})


if __name__ == "__main__":
    main()
