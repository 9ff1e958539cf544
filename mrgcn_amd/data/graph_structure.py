"""Stacked adjacency from integer triples — the layout contract of `A` (reference:
mrgcn/encodings/graph_structure.py:13-38 identity block + hstack, :70-108 one row-normalised block
per predicate and per inverse, :162-169 normalisation): N x (R*N) CSR, R = 2P + 1, block order
[p0, p0^-1, p1, p1^-1, ..., identity], column r*N + j, values 1 / deg_r(i) in float32 (what the
dataset archive stores, tarball.py:151-157).  One vectorised pass instead of a loop over predicates."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def stacked_coo(triples, N, P, value_mode="norm_f32"):
    """COO of the N x (R*N) stacked adjacency, R = 2P + 1 (inverses + identity)."""
    s, p, o = triples[:, 0], triples[:, 1], triples[:, 2]
    R = 2 * P + 1
    # forward block r = 2p: row s, col o, value 1 / #{o' : (s, p, o')}
    kf = p * N + s
    _, inv_f, cnt_f = np.unique(kf, return_inverse=True, return_counts=True)
    vf = 1.0 / cnt_f[inv_f]
    # inverse block r = 2p + 1: row o, col s, value 1 / #{s' : (s', p, o)}
    ki = p * N + o
    _, inv_i, cnt_i = np.unique(ki, return_inverse=True, return_counts=True)
    vi = 1.0 / cnt_i[inv_i]
    ident = np.arange(N, dtype=np.int64)
    rows = np.concatenate([s, o, ident])
    cols = np.concatenate([(2 * p) * N + o, (2 * p + 1) * N + s, (R - 1) * N + ident])
    vals = np.concatenate([vf, vi, np.ones(N)]).astype(np.float32)
    if value_mode == "ref_int8":
        vals = vals.astype(np.int8)  # truncation toward zero: only exact ones survive
    elif value_mode != "norm_f32":
        raise ValueError(value_mode)
    return rows, cols, vals, R


def adjacency_from_triples(triples: np.ndarray, num_nodes: int, num_pred: int) -> sp.csr_matrix:
    """`triples`: unique integer rows (s, p, o).  Column indices are sorted within a row (the
    reference's hstack output is not guaranteed to be: compare index *sets*)."""
    triples = np.asarray(triples, dtype=np.int64)
    rows, cols, vals, R = stacked_coo(triples, int(num_nodes), int(num_pred), value_mode="norm_f32")
    A = sp.csr_matrix((vals.astype(np.float32), (rows, cols)), shape=(num_nodes, R * num_nodes), dtype=np.float32)
    A.sort_indices()
    return A
