"""Boundary object handed to `MRGCN.forward` (reference: mrgcn/data/batch.py:13-149).

`FullBatch(A, X, batch_node_idx)` holds the scipy CSR adjacency N x (R*N), the feature list
`X = [X0, [datatype, encoding_sets, gpu_flag], ...]` and the node index.  `as_tensors_` turns
A into the sparse COO tensor the layers receive — with the reference's int8 cast by default
(`value_mode="ref_int8"`: the row-normalised floats truncate to 0/1) or
as float32 (`"norm_f32"`, the intended maths).  Unlike the reference's `Batch.to`, whose
`self.A.to(device)` discards its result (batch.py:122-123), `to` here really moves A: the
kernels need it in HBM."""
from __future__ import annotations

import numpy as np
import torch

VALUE_MODES = ("ref_int8", "norm_f32")


def scipy_sparse_to_pytorch_sparse(sp_input, dtype):
    """CSR -> uncoalesced COO with row-major `nonzero()` order (mrgcn/data/utils.py:165-170)."""
    indices = np.array(sp_input.nonzero())
    return torch.sparse_coo_tensor(torch.LongTensor(indices), torch.Tensor(sp_input.data),
                                   sp_input.shape, dtype=dtype)


class Batch:
    A = None
    X = None
    node_index = None
    device = None

    def __init__(self, batch_node_idx=None):
        self.device = torch.device("cpu")
        if batch_node_idx is not None:
            self.node_index = np.copy(batch_node_idx)

    def pad_(self, time_dim=1, pad_symbols=dict()):
        """Only variable-length (object-array) encodings need padding; the datatypes this
        package encodes (numeric / boolean / temporal vectors) are fixed width."""
        if self.X is None:
            return
        for _, encoding_sets, _ in self.X[1:]:
            for encodings, _, _ in encoding_sets:
                if getattr(encodings, "dtype", None) == np.dtype("O"):
                    raise NotImplementedError("variable-length encodings are outside mrgcn_amd's scope")

    def to_dense_(self):
        return

    def as_tensors_(self):
        self.node_index = torch.from_numpy(np.asarray(self.node_index))
        if self.X is None or isinstance(self.X[0], torch.Tensor):
            return
        self.X[0] = torch.from_numpy(self.X[0])
        for i, (_, encoding_sets, _) in enumerate(self.X[1:], 1):
            for j, (encodings, node_idx, seq_lengths) in enumerate(encoding_sets):
                self.X[i][1][j][0] = torch.from_numpy(encodings)
                self.X[i][1][j][1] = torch.from_numpy(node_idx)
                self.X[i][1][j][2] = torch.from_numpy(seq_lengths)

    def to(self, devices):
        if self.X is not None:
            for i, (datatype, encoding_sets, _) in enumerate(self.X[1:], 1):
                device = devices[datatype]
                for j, (encodings, node_idx, seq_lengths) in enumerate(encoding_sets):
                    self.X[i][1][j][0] = encodings.to(device)
                    self.X[i][1][j][1] = node_idx.to(device)
                    self.X[i][1][j][2] = seq_lengths.to(device)
        gcn_device = devices["relational"]
        if isinstance(self.A, torch.Tensor):
            self.A = self.A.to(gcn_device)
        kinds = {str(d) for d in devices.values()}
        self.device = next(iter(devices.values())) if len(kinds) == 1 else "ambigious"
        return self


class FullBatch(Batch):
    def __init__(self, A=None, X=None, batch_node_idx=None, value_mode="ref_int8"):
        super().__init__(batch_node_idx)
        assert value_mode in VALUE_MODES
        self.value_mode = value_mode
        if A is not None:
            self.A = A
        if X is not None:
            self.X = X

    def as_tensors_(self):
        super().as_tensors_()
        dtype = torch.int8 if self.value_mode == "ref_int8" else torch.float32
        self.A = scipy_sparse_to_pytorch_sparse(self.A, dtype=dtype)
