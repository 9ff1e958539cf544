"""Stacked adjacency from integer triples — the layout contract of `A` (reference:
mrgcn/encodings/graph_structure.py:13-38 identity block + hstack, :70-108 one row-normalised block
per predicate and per inverse, :162-169 normalisation): N x (R*N) CSR, R = 2P + 1, block order
[p0, p0^-1, p1, p1^-1, ..., identity], column r*N + j, values 1 / deg_r(i) in float32 (what the
dataset archive stores, tarball.py:151-157).  One vectorised pass instead of a loop over predicates."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from ..synth import stacked_coo


def adjacency_from_triples(triples: np.ndarray, num_nodes: int, num_pred: int) -> sp.csr_matrix:
    """`triples`: unique integer rows (s, p, o).  Column indices are sorted within a row (the
    reference's hstack output is not guaranteed to be: compare index *sets*)."""
    triples = np.asarray(triples, dtype=np.int64)
    rows, cols, vals, R = stacked_coo(triples, int(num_nodes), int(num_pred), value_mode="norm_f32")
    A = sp.csr_matrix((vals.astype(np.float32), (rows, cols)), shape=(num_nodes, R * num_nodes), dtype=np.float32)
    A.sort_indices()
    return A
