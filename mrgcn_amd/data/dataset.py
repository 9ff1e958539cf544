"""Dataset archive -> boundary objects (reference: run.py:63-69 + tasks/node_classification.py:
330-350 `mkbatches`): reads the tarball `mkdataset.py` writes (A, F, Y, data, sample_map, class_map)
and hands the adjacency to the device as a graph plan built from its CSR arrays."""
from __future__ import annotations

import numpy as np

from .batch import FullBatch
from .io.tarball import Tarball


def load_tarball(path: str) -> dict:
    with Tarball(path, "r") as tb:
        return {k: tb.get(k) for k in tb.list_members()}


def labels_of(Y_split):
    """(idx, targets) of one split's N x C label matrix, as `Y.nonzero()` gives them
    (node_classification.py:166-168)."""
    idx, targets = Y_split.nonzero()
    return np.asarray(idx, dtype=np.int64), np.asarray(targets, dtype=np.int64)


def full_batch(A_csr, X=None, value_mode: str = "ref_int8", device="cuda") -> FullBatch:
    """FullBatch whose `A` is a device adjacency handle carrying the plan built from the CSR arrays
    (`GraphPlan.from_csr`): the COO tensor of `FullBatch.as_tensors_` is never materialised."""
    from ..plan import GraphPlan
    N = A_csr.shape[0]
    R = A_csr.shape[1] // N
    plan = GraphPlan.from_csr(A_csr, N, R, value_mode=value_mode, device=device)
    batch = FullBatch(None, X if X is not None else [np.empty((N, 0), dtype=float)], np.arange(N),
                      value_mode=value_mode)
    batch.A = plan.as_adjacency_handle()
    return batch
