"""Boundary object handed to `MRGCN.forward` (reference: mrgcn/data/batch.py:13-149).

`FullBatch(A, X, batch_node_idx)` holds the scipy CSR adjacency N x (R*N), the feature list
`X = [X0, [datatype, encoding_sets, gpu_flag], ...]` and the node index.  `as_tensors_` turns
A into the sparse COO tensor the layers receive — with the reference's int8 cast by default
(`value_mode="ref_int8"`: the row-normalised floats truncate to 0/1) or
as float32 (`"norm_f32"`, the intended maths).  Unlike the reference's `Batch.to`, whose
`self.A.to(device)` discards its result (batch.py:122-123), `to` here really moves A: the
kernels need it in HBM.

`MiniBatch` / `A_Batch` and their helpers (batch.py:150-316) follow below: the batch structure
(row slices, neighbour sets) is built on the host with vectorised numpy instead of the
reference's per-entry Python loops; the layers turn every slice into a device graph plan once."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L_

VALUE_MODES = ("ref_int8", "norm_f32")


def scipy_sparse_to_pytorch_sparse(sp_input, dtype):
    """CSR -> uncoalesced COO with row-major `nonzero()` order (mrgcn/data/utils.py:165-170)."""
    indices = np.array(sp_input.nonzero())
    return torch.sparse_coo_tensor(torch.LongTensor(indices), torch.Tensor(sp_input.data),
                                   sp_input.shape, dtype=dtype)


_MAX_BATCH_LENGTH = 999  # data/utils.py:109,:135: longer members do not widen the batch further


def pad_token_sequences(seqs, pad_symbol=0, min_width=5):
    """`seqs`: object array of 1-D integer arrays.  One `[n, L]` int64 matrix, member i left-aligned
    in row i, the rest = `pad_symbol` (data/utils.py:135-152).  The reference fills with -1 first and
    replaces every -1 afterwards, so a token id of -1 turns into the pad symbol as well; a member
    longer than L is an error there (shape mismatch) and here."""
    n = len(seqs)
    lens = np.fromiter((len(a) for a in seqs), dtype=np.int64, count=n)
    L = max(int(min_width), min(int(lens.max()) if n else 0, _MAX_BATCH_LENGTH))
    if n and int(lens.max()) > L:
        raise ValueError(f"a sequence of {int(lens.max())} tokens does not fit the padded width {L}")
    out = np.full((n, L), pad_symbol, dtype=np.int64)
    if n:
        flat = np.concatenate([np.asarray(a).reshape(-1) for a in seqs]).astype(np.int64)
        out[np.arange(L)[None, :] < lens[:, None]] = flat
        out[out == -1] = pad_symbol
    return out


def pad_sparse_members(mats, time_dim=1, min_width=5):
    """`mats`: object array of scipy CSR matrices that differ in length along `time_dim`.  Each is
    re-declared with that dimension = `L` (no entry moves; data/utils.py:109-133)."""
    import scipy.sparse as sp
    n = len(mats)
    longest = max((m.shape[time_dim] for m in mats), default=0)
    L = max(int(min_width), min(int(longest), _MAX_BATCH_LENGTH))
    out = np.empty(n, dtype=object)
    for i, m in enumerate(mats):
        if time_dim == 1:
            out[i] = sp.csr_matrix((m.data, m.indices, m.indptr), shape=(m.shape[0], L), dtype=m.dtype)
        else:  # rows are the time axis: the row pointer grows by empty rows
            indptr = np.concatenate([m.indptr, np.full(max(L - m.shape[0], 0), m.indptr[-1], m.indptr.dtype)])
            out[i] = sp.csr_matrix((m.data, m.indices, indptr), shape=(L, m.shape[1]), dtype=m.dtype)
    return out


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
        """Variable-length encodings (object arrays) are brought to one width per encoding set
        (batch.py:25-54): token sequences become an int64 matrix `[n, L]` filled with the datatype's
        pad symbol (0 when none is given), CSR members (WKT / image rows) are widened with empty
        columns to `L`; `L = max(max(seq_length), min(longest member, 999))`
        (data/utils.py:109-152).  Fixed-width (numeric) encodings are left alone."""
        if self.X is None:
            return
        for i, (datatype, encoding_sets, _) in enumerate(self.X[1:], 1):
            for j, (encodings, _, seq_length) in enumerate(encoding_sets):
                if getattr(encodings, "dtype", None) != np.dtype("O"):
                    continue
                width = int(max(seq_length))
                if isinstance(encodings[0], np.ndarray):
                    padded = pad_token_sequences(encodings, pad_symbols.get(datatype, 0), width)
                else:
                    padded = pad_sparse_members(encodings, time_dim, width)
                self.X[i][1][j][0] = padded

    def to_dense_(self):
        """Object arrays of CSR members become one dense `[n, rows, cols]` array (batch.py:56-68)."""
        if self.X is None:
            return
        import scipy.sparse as sp
        for i, (_, encoding_sets, _) in enumerate(self.X[1:], 1):
            for j, (encodings, _, _) in enumerate(encoding_sets):
                if getattr(encodings, "dtype", None) != np.dtype("O") or len(encodings) == 0:
                    continue
                if not isinstance(encodings[0], sp.csr_matrix):
                    continue
                self.X[i][1][j][0] = np.stack([np.asarray(m.toarray()) for m in encodings])

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


# ---- mini-batch boundary objects (reference: mrgcn/data/batch.py:150-316) ------------------------
def getNeighboursSparse(A, idx):
    """Sorted global node ids of every source node the rows `idx` of the scipy CSR `A` touch,
    irrespective of relation (batch.py:233-249; vectorised, same result)."""
    import scipy.sparse as sp
    assert isinstance(A, sp.csr_matrix)
    idx = np.asarray(idx, dtype=np.int64)
    num_nodes = A.shape[0]
    lo, hi = A.indptr[idx].astype(np.int64), A.indptr[idx + 1].astype(np.int64)
    n = hi - lo
    if n.sum() == 0:
        return np.zeros(0, dtype=np.int64)
    pos = np.repeat(lo - np.concatenate([[0], np.cumsum(n)[:-1]]), n) + np.arange(int(n.sum()))
    return np.unique(A.indices[pos].astype(np.int64) % num_nodes)


def getAdjacencyNodeColumnIdx(idx, num_nodes, num_relations):
    """Columns r*num_nodes + i of every node i in `idx` for every relation r, relation-major
    (batch.py:251-256)."""
    idx = torch.as_tensor(idx, dtype=torch.long)
    r = torch.arange(num_relations, dtype=torch.long, device=idx.device)
    return (r[:, None] * num_nodes + idx[None, :]).reshape(-1)


def sliceSparseCOO(t, idx):
    """Entries of the sparse COO `t` whose column is in (the ascending) `idx`, columns renumbered
    to their position in `idx`, values replaced by float32 ONES (batch.py:258-270: the stored
    values — incl. entries the int8 cast truncated to 0 — are discarded)."""
    ind = t._indices()
    idx = idx.to(ind.device)
    pos = torch.searchsorted(idx, ind[1])
    pos_c = pos.clamp(max=max(idx.numel() - 1, 0))
    keep = (idx[pos_c] == ind[1]) if idx.numel() else torch.zeros_like(ind[1], dtype=torch.bool)
    row, col = ind[0][keep], pos[keep]
    return torch.sparse_coo_tensor(torch.vstack([row, col]),
                                   torch.ones(col.numel(), dtype=torch.float32, device=ind.device),
                                   size=[t.shape[0], idx.numel()])


def mksubset(X, sample_idx):
    """Rows / encodings of the nodes in `sample_idx`, same list structure (batch.py:272-316);
    fixed-width (numeric array) encodings only."""
    X0, F = X[0], X[1:]
    X_sample = [X0[sample_idx]]
    for modality, F_set, gpu_acceleration in F:
        F_set_sample = []
        for encodings, nodes_idx, seq_lengths in F_set:
            if getattr(encodings, "dtype", None) == np.dtype("O"):
                raise NotImplementedError("variable-length encodings are outside mrgcn_amd's scope")
            common = np.intersect1d(nodes_idx, sample_idx)
            if len(common) <= 0:
                F_set_sample.append([np.empty(0), np.empty(0), np.empty(0)])
                continue
            mask = np.isin(nodes_idx, common)
            F_set_sample.append([encodings[mask], np.array(sorted(common)), seq_lengths[mask]])
        X_sample.append([modality, F_set_sample, gpu_acceleration])
    return X_sample


class A_Batch:
    """Row slices of A and neighbour sets for an L-layer mini-batch (batch.py:166-231):
    `row[i]` = A[sample_i] (scipy CSR, later sparse COO), `neighbours[i]` = sorted source nodes
    those rows touch; sample_0 = the batch nodes, sample_{i+1} = neighbours[i]."""

    def __init__(self, A=None, batch_idx=None, num_layers=0, value_mode="ref_int8"):
        assert value_mode in VALUE_MODES
        self.value_mode = value_mode
        self.neighbours, self.row = [], []
        self.node_index = None
        self.device = torch.device("cpu")
        self._a_idx = {}  # layer slot -> cached A_idx tensor (keeps the layers' slice cache warm)
        if batch_idx is not None:
            self.node_index = np.copy(batch_idx)
        if A is not None:
            self._populate(A, num_layers)

    def _populate(self, A, num_layers):
        sample_idx = self.node_index
        for _ in range(num_layers):
            self.row.append(A[sample_idx])
            neighbours_idx = getNeighboursSparse(A, sample_idx)
            self.neighbours.append(neighbours_idx)
            sample_idx = neighbours_idx

    def as_tensors_(self):
        dtype = torch.int8 if self.value_mode == "ref_int8" else torch.float32
        self.node_index = torch.from_numpy(np.asarray(self.node_index))
        self.row = [scipy_sparse_to_pytorch_sparse(a, dtype=dtype) for a in self.row]
        self.neighbours = [torch.from_numpy(np.asarray(a)) for a in self.neighbours]

    def to(self, device):
        self.node_index = self.node_index.to(device)
        self.neighbours = [t.to(device) for t in self.neighbours]
        self.row = [t.to(device) for t in self.row]
        self._a_idx = {}
        self.device = device
        return self


class MiniBatch(Batch):
    def __init__(self, A=None, X=None, batch_node_idx=None, num_layers=None, value_mode="ref_int8", plan=None):
        """`plan` (a GraphPlan of the FULL graph on the GPU) instead of `A`: the batch structure is a masked batch on
        that plan (A_BatchMasked: no slices, no per-batch plans); everything else — `X` subset to the outermost
        neighbours, `as_tensors_`, `to(devices)`, `MRGCN.forward(batch)` — is unchanged."""
        super().__init__(batch_node_idx)
        if plan is not None:
            self.A = A_BatchMasked(plan, self.node_index, num_layers)
            if X is not None:
                self.X = mksubset(X, self.A.neighbours[-1].cpu().numpy())
        elif A is not None:
            self.A = A_Batch(A, self.node_index, num_layers, value_mode=value_mode)
            if X is not None:  # not featureless: features of the outermost neighbours
                self.X = mksubset(X, self.A.neighbours[-1])

    def as_tensors_(self):
        super().as_tensors_()
        self.A.as_tensors_()

    def to(self, devices):
        if self.X is not None:
            for i, (datatype, encoding_sets, _) in enumerate(self.X[1:], 1):
                device = devices[datatype]
                for j, (encodings, node_idx, seq_lengths) in enumerate(encoding_sets):
                    self.X[i][1][j][0] = encodings.to(device)
                    self.X[i][1][j][1] = node_idx.to(device)
                    self.X[i][1][j][2] = seq_lengths.to(device)
        self.A.to(devices["relational"])
        kinds = {str(d) for d in devices.values()}
        self.device = next(iter(devices.values())) if len(kinds) == 1 else "ambigious"
        return self


# ---- batch structure built on the device -----------------------------------------------------------
class DeviceCSR:
    """The stacked adjacency's CSR arrays resident in HBM (uploaded once), for building mini-batches
    without going back to scipy on the host every time batches are re-sampled."""

    def __init__(self, A, device="cuda"):
        self.shape = tuple(A.shape)
        self.indptr = torch.from_numpy(np.ascontiguousarray(A.indptr, dtype=np.int64)).to(device)
        self.indices = torch.from_numpy(np.ascontiguousarray(A.indices, dtype=np.int64)).to(device)
        self.data = torch.from_numpy(np.ascontiguousarray(A.data, dtype=np.float32)).to(device)
        self._ws = None  # workspace of the frontier kernels, grown on demand and reused by every batch

    def workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.indptr.device)
        return self._ws

    def frontier(self, sample: torch.Tensor, value_mode: str = "ref_int8"):
        """One layer of a batch through the frontier kernels (csrc/minibatch.hip; C ABI mrgcn_frontier_count /
        _emit): the rows `sample` of A as COO (row in sample, global column, value), the same entries' columns
        in the sliced numbering r * n_b + position(node), and the ascending neighbour list.  One two-word
        readback (the output sizes) is the only host synchronisation."""
        from .. import _lib as L
        lib = L.load()
        dev = self.indptr.device
        N = self.shape[0]
        sample = sample.to(dev, torch.long).contiguous()
        n_s = int(sample.numel())
        ws = self.workspace(lib.mrgcn_frontier_workspace_bytes(N, n_s))
        row_off = torch.empty(n_s + 1, dtype=torch.long, device=dev)
        node_pos = torch.empty(N + 1, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            L.check(lib.mrgcn_frontier_count(self.indptr.data_ptr(), self.indices.data_ptr(), N, sample.data_ptr(),
                                             n_s, row_off.data_ptr(), node_pos.data_ptr(), ws.data_ptr(),
                                             ws.numel(), s), "mrgcn_frontier_count")
            sizes = torch.stack([row_off[-1], node_pos[-1].long()]).cpu()   # the one readback
            nnz, n_b = int(sizes[0]), int(sizes[1])
            row = torch.empty(nnz, dtype=torch.long, device=dev)
            col = torch.empty(nnz, dtype=torch.long, device=dev)
            col_sl = torch.empty(nnz, dtype=torch.long, device=dev)
            i8 = value_mode == "ref_int8"
            val = torch.empty(nnz, dtype=torch.int8 if i8 else torch.float32, device=dev)
            nb = torch.empty(n_b, dtype=torch.long, device=dev)
            L.check(lib.mrgcn_frontier_emit(self.indptr.data_ptr(), self.indices.data_ptr(), self.data.data_ptr(), N,
                                            sample.data_ptr(), n_s, row_off.data_ptr(), node_pos.data_ptr(), n_b,
                                            L.VAL_I8 if i8 else L.VAL_F32, row.data_ptr(), col.data_ptr(),
                                            val.data_ptr(), col_sl.data_ptr(), nb.data_ptr(), s),
                    "mrgcn_frontier_emit")
        return row, col, val, col_sl, nb


class A_BatchDevice(A_Batch):
    """`A_Batch` whose row slices and neighbour sets are produced on the GPU from a `DeviceCSR` by the frontier
    kernels: same `row` / `neighbours` / `node_index` attributes, already tensors on the device (as after
    `as_tensors_()` + `to(device)`), same contents as the host build (bit-exact against the reference's goldens).
    The column-compacted slice of every layer (what `sliceSparseCOO(row[i], A_idx_i)` returns, batch.py:258-270)
    comes out of the same kernel pass and is attached to `row[i]`, so the layers never run the searchsorted
    slicer."""

    def __init__(self, A_dev: DeviceCSR, batch_idx, num_layers, value_mode="ref_int8", short_lived=False):
        """`short_lived`: the batch is used for one step and dropped (batches re-sampled every step): its slices ask for
        the quick plan build (plan.GraphPlan(lean=True)) instead of the layout passes that pay off over many epochs
        on the same batch (the reference's flow: `mkbatches` once, then every epoch over the same batches)."""
        assert value_mode in VALUE_MODES
        self.value_mode = value_mode
        self.short_lived = bool(short_lived)
        self.neighbours, self.row = [], []
        self._a_idx = {}
        dev = A_dev.indptr.device
        self.device = dev
        self.node_index = torch.as_tensor(np.asarray(batch_idx), dtype=torch.long, device=dev)
        num_nodes = A_dev.shape[0]
        R = A_dev.shape[1] // num_nodes
        sample = self.node_index
        for i in range(num_layers):
            row, col, val, col_sl, nb = A_dev.frontier(sample, value_mode)
            a = torch.sparse_coo_tensor(torch.stack([row, col]), val, (sample.numel(), A_dev.shape[1]))
            a_idx = getAdjacencyNodeColumnIdx(nb, num_nodes, R)
            sliced = torch.sparse_coo_tensor(torch.stack([row, col_sl]),
                                             torch.ones(col_sl.numel(), dtype=torch.float32, device=dev),
                                             (sample.numel(), R * nb.numel()))
            self._a_idx[i] = a_idx
            if self.short_lived:
                a._mrgcn_lean = sliced._mrgcn_lean = True
            a._mrgcn_slice = (a_idx, sliced)   # what GraphConvolution._forward_mini_batch looks up
            self.row.append(a)
            self.neighbours.append(nb)
            sample = nb

    def as_tensors_(self):
        return

    def to(self, device):
        return self if torch.device(device) == self.device else super().to(device)


class A_BatchMasked:
    """The mini-batch structure of `A_Batch` (batch.py:166-231) WITHOUT slices: every layer's sample is a row set on
    the full graph's plan (a forward gradient support, csrc/masked.hip) —
        supports[0]   the batch nodes' rows,   neighbours[0] = the source nodes those rows touch
        supports[i+1] the rows of neighbours[i], ...
    `neighbours[i]` are the same sorted node ids `getNeighboursSparse` returns (int64, on the plan's device), so the
    caller's `X[batch.neighbours[-1]]` is unchanged; `row[i]` is the support (what `RGCN.forward` hands to layer
    num_layers-1-i).  A re-sampled batch costs `num_layers` support builds on the existing plan: no slice tensors, no
    per-batch plans, no worker threads.  Rows of the output follow `batch_idx` (any order, repeats allowed).

    `full_batch_values=True`: the feature term multiplies the adjacency's stored values like the input term does —
    not the reference's mini-batch arithmetic (its slices drop the values: batch.py:258-270) but its FULL-batch
    arithmetic (graph.py:93-95) restricted to the batch's receptive field: a batch that holds every labelled node then
    trains exactly like the full-batch epoch while computing only the rows the loss can see."""

    def __init__(self, plan, batch_idx, num_layers, full_batch_values=False):
        from ..plan import GraphSupport
        dev = plan.device
        if plan.num_rows != plan.num_nodes:
            raise ValueError("the masked batch needs the square stacked adjacency (rows = nodes)")
        idx = torch.as_tensor(np.asarray(batch_idx) if not isinstance(batch_idx, torch.Tensor) else batch_idx,
                              dtype=torch.long).to(dev)
        self.plan, self.node_index, self.device = plan, idx, dev
        self.value_mode = None
        flags = torch.zeros(plan.num_rows, dtype=torch.uint8, device=dev)
        flags[idx] = 1
        # every layer's support in one build (one host wait): level i+1's rows are level i's NODE_FLAGS
        self.supports = GraphSupport.chain(plan, flags, num_layers, forward=True) if num_layers else []
        for sup in self.supports:
            sup.feature_values = bool(full_batch_values)
        self.neighbours = [sup.view(L_.SUP_LNODE).long() for sup in self.supports]
        self.row = self.supports
        # position of every batch node among the (sorted, distinct) rows the top layer computes
        self.out_rank = self.supports[0].view(L_.SUP_ROWRANK)[idx].long() if num_layers else idx

    def as_tensors_(self):
        return None

    def to(self, device):
        if torch.device(device) != self.device and torch.device(device).index is not None:
            raise ValueError("a masked batch lives on its plan's device")
        return self

    def close(self):
        for s in self.supports:
            s.close()
        self.supports, self.row = [], []


class BatchPrefetcher:
    """Iterates over device-built mini-batches that worker threads prepare ahead — batch structure (frontier
    kernels) and, when `model` is given, the slice plans — on their own streams while the caller trains on the
    previous one:

        for ab in BatchPrefetcher(dcsr, sampler, num_layers, model=model):
            loss = criterion(model(X[ab.neighbours[-1]], ab), ...)
            ...

    `batches`: an iterable of node-index arrays (one per step: a sampler); batches come out in its order.  Every
    batch is short-lived (lean slice plans).  A batch build is mostly waiting (host round trips of sizes between
    short device passes), so `workers` threads (default 3) build consecutive batches side by side; at most `depth`
    batches (default: `workers`) exist ahead of the caller.  What the hand-over guarantees: the caller's current
    stream waits for the batch's build (an event, no host wait); every tensor of the batch is registered with the
    caller's stream (`record_stream`), so the allocator does not hand its memory to a worker again before the
    caller's kernels have run; the plans of a finished batch are released by a worker behind an event the iterator
    records on the caller's stream when the NEXT batch is asked for (mrgcn_plan_destroy_after: no device-wide wait,
    the builds under way are not stalled).
    The reference has no counterpart (its batches are built once on the host, node_classification.py:128)."""

    def __init__(self, A_dev: DeviceCSR, batches, num_layers, value_mode="ref_int8", model=None, workers=3,
                 depth=None):
        import queue
        import threading
        self.num_layers, self.value_mode, self.model = num_layers, value_mode, model
        self.dev = A_dev.indptr.device
        self._it = enumerate(iter(batches))
        self._lock = threading.Lock()            # the sampler
        self._cv = threading.Condition()         # slots / counters below
        self._slots = {}                         # step -> (batch, event behind its build)
        self._taken = 0                          # steps handed to the caller
        self._next_step = 0                      # steps drawn from the sampler
        self._total = None                       # number of steps, once the sampler is exhausted
        self._depth = max(1, int(depth if depth is not None else workers))
        self._done = queue.Queue()               # (batch, event behind its last use) for a worker to release
        self._err = None
        self._last = None
        self._closed = False
        self._threads = []
        for _ in range(max(1, int(workers))):
            csr = DeviceCSR.__new__(DeviceCSR)   # the resident arrays, a workspace of its own
            csr.shape, csr.indptr, csr.indices, csr.data, csr._ws = A_dev.shape, A_dev.indptr, A_dev.indices, A_dev.data, None
            t = threading.Thread(target=self._work, args=(csr, torch.cuda.Stream(device=self.dev)), daemon=True)
            t.start()
            self._threads.append(t)

    # -- workers --------------------------------------------------------------------------------------------
    def _release_finished(self):
        import queue
        while True:
            try:
                ab, ev = self._done.get_nowait()
            except queue.Empty:
                return
            for a in ab.row:
                for t in (a, getattr(a, "_mrgcn_slice", (None, None))[1]):
                    p = getattr(t, "_mrgcn_plan", None) if t is not None else None
                    if p is not None:
                        p.close(after_event=ev)

    def _work(self, csr, stream):
        try:
            with torch.cuda.device(self.dev), torch.cuda.stream(stream):
                while self._err is None:
                    with self._lock:
                        try:
                            k, idx = next(self._it)
                        except StopIteration:
                            with self._cv:
                                if self._total is None:
                                    self._total = self._next_step
                                self._cv.notify_all()
                            break
                        self._next_step = k + 1
                    with self._cv:               # bounded look-ahead
                        self._cv.wait_for(lambda: k < self._taken + self._depth or self._err is not None)
                    if self._err is not None:
                        break
                    ab = A_BatchDevice(csr, idx, self.num_layers, self.value_mode, short_lived=True)
                    if self.model is not None:
                        self.model.prepare_batch(ab)
                    ev = torch.cuda.Event()
                    ev.record(stream)
                    with self._cv:
                        self._slots[k] = (ab, ev)
                        self._cv.notify_all()
                    self._release_finished()
        except BaseException as e:  # noqa: BLE001  (re-raised on the caller's thread)
            with self._cv:
                self._err = e
                self._cv.notify_all()
        finally:
            self._release_finished()

    # -- caller ---------------------------------------------------------------------------------------------
    @staticmethod
    def _tensors(ab):
        yield ab.node_index
        for i, a in enumerate(ab.row):
            yield ab.neighbours[i]
            yield ab._a_idx[i]
            for t in (a, a._mrgcn_slice[1]):
                yield t._indices()
                yield t._values()

    def _retire_last(self, cur):
        if self._last is not None:
            ev = torch.cuda.Event()
            ev.record(cur)               # behind everything the caller submitted with the previous batch
            self._done.put((self._last, ev))
            self._last = None

    def __iter__(self):
        return self

    def __next__(self):
        cur = torch.cuda.current_stream(self.dev)
        self._retire_last(cur)
        k = self._taken
        with self._cv:
            self._cv.wait_for(lambda: k in self._slots or self._err is not None
                              or (self._total is not None and k >= self._total))
            if self._err is not None:
                err, self._err = self._err, StopIteration()  # (raised once, here; the workers have stopped)
                if isinstance(err, StopIteration):
                    raise StopIteration
                raise err
            if k not in self._slots:
                done = True
            else:
                done = False
                ab, ev = self._slots.pop(k)
                self._taken = k + 1
                self._cv.notify_all()
        if done:
            # the workers exit with the sampler: what the caller retired since then is released here, not at a later
            # close() nobody may call
            self._release_finished()
            raise StopIteration
        cur.wait_event(ev)
        for t in self._tensors(ab):
            t.record_stream(cur)
        self._last = ab
        return ab

    def close(self):
        """Hands the last batch back, stops the workers and waits for them (call after the loop)."""
        self._retire_last(torch.cuda.current_stream(self.dev))
        with self._cv:
            if self._err is None and self._total is None:
                self._err = StopIteration()      # (an early exit of the loop: no further batch is started)
            self._cv.notify_all()
        for t in self._threads:
            t.join(timeout=30.0)
        # batches built ahead that the caller never took: their plans go back behind the events of their builds (not
        # through GraphPlan.__del__ from the garbage collector, which waits for the whole device)
        with self._cv:
            left, self._slots = list(self._slots.values()), {}
        for ab, ev in left:
            self._done.put((ab, ev))
        self._release_finished()
        self._closed = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            if not getattr(self, "_closed", True):
                self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass
