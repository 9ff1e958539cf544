"""GraphPlan — host-side handle of the device-resident graph plan (built once per
adjacency).  Input is the reference's own boundary object: the sparse COO tensor that
`FullBatch.as_tensors_` produces (mrgcn/data/batch.py:144-149; int8 values) or a float32
COO (the layer is value-generic: mrgcn/layers/graph.py:75 `A.float()`)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# Handles whose owner was dropped while a stream capture was under way in this thread (typically the garbage collector
# running a __del__ in the middle of a capture): releasing them waits for the device, which would invalidate the
# capture, so they wait here for the next release outside one.
_PARKED: list = []


def _capturing() -> bool:
    try:
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    except Exception:  # noqa: BLE001  (interpreter shutdown)
        return False


def _release(kind: str, handle, after_event=None):
    lib = L.load()
    if _capturing():
        _PARKED.append((kind, handle))
        return
    while _PARKED:
        k, h = _PARKED.pop()
        (lib.mrgcn_support_destroy if k.startswith("support") else lib.mrgcn_plan_destroy)(h)
    if kind == "support_ordered":
        lib.mrgcn_support_destroy_ordered(handle)
    elif kind == "support":
        lib.mrgcn_support_destroy(handle)
    elif after_event is not None:
        lib.mrgcn_plan_destroy_after(handle, after_event.cuda_event)
    else:
        lib.mrgcn_plan_destroy(handle)


def _flags(prune_zeros, replicate, lean=False) -> int:
    """replicate: None = the library's default (env MRGCN_REPLICATE), True / False = force."""
    f = (L.PLAN_PRUNE_ZEROS if prune_zeros else 0) | (L.PLAN_LEAN if lean else 0)
    if replicate is True:
        f |= L.PLAN_REPLICATE
    elif replicate is False:
        f |= L.PLAN_NO_REPLICATE
    return f


def _row_bytes_arg(operand_row_bytes):
    """(ctypes array or None, count) for the hinted constructors: at most four distinct sizes."""
    sizes = sorted({int(b) for b in (operand_row_bytes or ())})[:4]
    if not sizes:
        return None, 0
    return (C.c_int32 * len(sizes))(*sizes), len(sizes)


# row stride (floats) of a hidden layer's padded output at least this (16: rows of one 64-byte sector — the next layer's
# transform gathers them)
_PAD_OUT_LD = int(__import__("os").environ.get("MRGCN_PAD_OUT_LD", "0"))


class GraphPlan:
    def __init__(self, A: torch.Tensor, num_nodes: int, num_relations: int,
                 prune_zeros: bool = False, replicate=None, operand_row_bytes=None, lean: bool = False):
        """`operand_row_bytes`: row sizes (bytes) of the compact operands the plan's products will read
        (mrgcn_plan_create_hinted): a layout hint for the operand order, never for results.
        `lean` (MRGCN_PLAN_LEAN): the quick build for a small, short-lived adjacency — the slices of a re-sampled
        mini-batch — without the layout passes that pay off on a graph that is multiplied many times."""
        if not A.is_sparse:
            raise TypeError("A must be a torch sparse COO tensor")
        if not A.is_cuda:
            raise L.MrgcnError("GraphPlan needs A on the GPU (no CPU path exists in mrgcn_amd)")
        lib = L.load()
        self.device = A.device
        idx = A._indices()
        val = A._values()
        if val.dtype == torch.int8:
            vd = L.VAL_I8
        elif val.dtype == torch.float32:
            vd = L.VAL_F32
        else:
            raise TypeError(f"unsupported adjacency dtype {val.dtype}")
        rows = idx[0].contiguous()
        cols = idx[1].contiguous()
        val = val.contiguous()
        if rows.dtype != torch.int64:
            rows, cols = rows.long(), cols.long()
        self.num_rows = int(A.shape[0])
        self.num_nodes = int(num_nodes)
        self.num_relations = int(num_relations)
        if int(A.shape[1]) != self.num_nodes * self.num_relations:
            raise ValueError("A.shape[1] != num_relations * num_nodes")
        handle = C.c_void_p()
        rb, nrb = _row_bytes_arg(operand_row_bytes)
        with torch.cuda.device(self.device):
            L.check(lib.mrgcn_plan_create_hinted(
                C.byref(handle), self.num_rows, self.num_nodes, self.num_relations,
                int(val.numel()), rows.data_ptr(), cols.data_ptr(), val.data_ptr(), vd,
                _flags(prune_zeros, replicate, lean), C.cast(rb, C.c_void_p) if nrb else None, nrb,
                _stream_ptr(self.device)), "mrgcn_plan_create_hinted")
        self.lean = bool(lean)
        self.operand_row_bytes = tuple(rb) if nrb else ()
        self._adopt(handle)

    def _adopt(self, handle):
        lib = L.load()
        self._h = handle
        info = L.PlanInfo()
        L.check(lib.mrgcn_plan_info(self._h, C.byref(info)))
        self.nnz, self.ncols = int(info.nnz), int(info.ncols)
        self.max_row_nnz, self.max_col_nnz = int(info.max_row_nnz), int(info.max_col_nnz)
        self.long_rows, self.long_cols = int(info.long_rows), int(info.long_cols)
        self.device_bytes = int(info.device_bytes)
        # rows of the compact operand M (= ncols unless the plan keeps replicas) and how many are copies
        self.nop, self.n_rep = int(info.operand_rows), int(info.replicas)
        self._ptr_cache = {}

    @classmethod
    def from_csr(cls, A_csr, num_nodes: int, num_relations: int, value_mode: str = "ref_int8",
                 device="cuda", prune_zeros: bool = False, replicate=None, operand_row_bytes=None) -> "GraphPlan":
        """Plan straight from a scipy CSR (what the dataset archive holds, tarball.py:151-157): the
        three arrays are uploaded as they are and expanded on the device — no host `.nonzero()`, no
        int64 COO.  `value_mode="ref_int8"` applies the reference's boundary cast (batch.py:144-149)."""
        lib = L.load()
        device = torch.device(device)
        if device.type != "cuda":
            raise L.MrgcnError("GraphPlan needs a GPU (no CPU path exists in mrgcn_amd)")
        if int(A_csr.shape[1]) != int(num_nodes) * int(num_relations):
            raise ValueError("A.shape[1] != num_relations * num_nodes")
        self = cls.__new__(cls)
        self.device = device if device.index is not None else torch.device("cuda", torch.cuda.current_device())
        self.num_rows, self.num_nodes, self.num_relations = int(A_csr.shape[0]), int(num_nodes), int(num_relations)
        indptr = torch.from_numpy(np.ascontiguousarray(A_csr.indptr, dtype=np.int32)).to(self.device)
        indices = torch.from_numpy(np.ascontiguousarray(A_csr.indices, dtype=np.int32)).to(self.device)
        data = torch.from_numpy(np.ascontiguousarray(A_csr.data, dtype=np.float32)).to(self.device)
        handle = C.c_void_p()
        rb, nrb = _row_bytes_arg(operand_row_bytes)
        with torch.cuda.device(self.device):
            L.check(lib.mrgcn_plan_create_csr_hinted(
                C.byref(handle), self.num_rows, self.num_nodes, self.num_relations, int(data.numel()),
                indptr.data_ptr(), indices.data_ptr(), data.data_ptr(), 1 if value_mode == "ref_int8" else 0,
                _flags(prune_zeros, replicate), C.cast(rb, C.c_void_p) if nrb else None, nrb,
                _stream_ptr(self.device)), "mrgcn_plan_create_csr_hinted")
        self.lean = False
        self.operand_row_bytes = tuple(rb) if nrb else ()
        self._adopt(handle)
        return self

    def as_adjacency_handle(self) -> torch.Tensor:
        """An (entry-less) sparse tensor of A's shape that carries this plan: what `model(X, A)` /
        `FullBatch.A` take when the adjacency never existed as a COO tensor.  Full-batch only (the
        mini-batch slicer needs the entries)."""
        A = torch.sparse_coo_tensor(torch.zeros((2, 0), dtype=torch.long, device=self.device),
                                    torch.zeros(0, dtype=torch.float32, device=self.device),
                                    (self.num_rows, self.num_relations * self.num_nodes))
        A._mrgcn_plan = self
        return A

    # -- lifetime -------------------------------------------------------------------
    def close(self, after_event=None):
        """Releases the plan's device memory.  Default: after a wait for everything in flight on the device.
        `after_event`: a torch.cuda.Event recorded behind the last work that uses the plan — the host then waits for
        that event only and other streams keep running (mrgcn_plan_destroy_after)."""
        if getattr(self, "_h", None) is not None and self._h:
            for sup in self.__dict__.pop("_supports", {}).values():  # (they read the plan's arrays: first)
                sup.close()
            _release("plan", self._h, after_event)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if self._h is None:
            raise L.MrgcnError("plan already destroyed")
        return self._h

    # -- introspection ----------------------------------------------------------------
    def export(self, which: int) -> np.ndarray:
        """Copies one plan array to the host (index arrays int32, value arrays float32)."""
        ptr, n = self.array_ptr(which)
        out = np.empty(n, dtype=np.float32 if which in L.FLOAT_ARRAYS else np.int32)
        L.check(L.load().mrgcn_plan_export(self.handle, which, out.ctypes.data, out.nbytes))
        return out

    def array_ptr(self, which: int):
        if which not in self._ptr_cache:
            p, n = C.c_void_p(), C.c_int64()
            L.check(L.load().mrgcn_plan_array(self.handle, which, C.byref(p), C.byref(n)))
            self._ptr_cache[which] = (p.value or 0, int(n.value))
        return self._ptr_cache[which]

    # -- products ---------------------------------------------------------------------
    def view_rows(self, view: int) -> int:
        return self.ncols if view == L.VIEW_TRANSPOSED else self.num_rows

    def spmm(self, view: int, D: torch.Tensor, F: int | None = None, out: torch.Tensor | None = None,
             bias: torch.Tensor | None = None, relu: bool = False, out_index: int = 0,
             out_rows: int | None = None, pad_writable: bool = False, padded_rows: bool = False,
             two_pass: bool = False) -> torch.Tensor:
        """Y[i, :F] = sum_e val[e] * D[idx[e], :F] over row i of `view` (see mrgcn_spmm_f32).
        `out_index` is a device pointer (int) to an int32 row redirection table or 0.
        `padded_rows` (without `out`): the COMPACT product of a narrow layer whose F is not a multiple of
        four returns the first F columns of a buffer with rows padded to whole 16-byte pieces (row stride
        4*ceil(F/4)): the kernel then stores whole pieces and consecutive rows fill their lines
        (MRGCN_SPMM_PAD_WRITABLE) — what a hidden layer's output is kept in.  `pad_writable` says the same of
        a caller's `out`.  `two_pass`: MRGCN_SPMM_TWO_PASS (rows of several chunks finished by a second launch)."""
        assert D.is_cuda and D.dtype in (torch.float32, torch.bfloat16) and D.dim() == 2 and D.stride(1) == 1
        bf16 = D.dtype == torch.bfloat16  # dense operand in bf16: fp32 values, accumulation and Y
        F = int(D.shape[1] if F is None else F)
        if out is None:
            rows = self.view_rows(view) if out_rows is None else out_rows
            if padded_rows and view == L.VIEW_COMPACT and F <= 16 and F % 4 and not out_index:
                out = torch.empty((rows, max((F + 3) // 4 * 4, _PAD_OUT_LD)), dtype=torch.float32, device=D.device)[:, :F]
                pad_writable = True
            else:
                out = torch.empty((rows, F), dtype=torch.float32, device=D.device)
        assert out.dtype == torch.float32 and out.stride(1) == 1 and out.is_cuda
        if bias is not None:
            assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() >= F
        flags = ((L.SPMM_RELU if relu else 0) | (L.SPMM_PAD_WRITABLE if pad_writable else 0)
                 | (L.SPMM_TWO_PASS if two_pass else 0))
        with torch.cuda.device(D.device):
            fn = L.load().mrgcn_spmm_bf16 if bf16 else L.load().mrgcn_spmm_f32
            L.check(fn(self.handle, view, D.data_ptr(), D.stride(0), F, out.data_ptr(), out.stride(0),
                       bias.data_ptr() if bias is not None else 0, flags, out_index,
                       _stream_ptr(D.device)), "mrgcn_spmm_bf16" if bf16 else "mrgcn_spmm_f32")
        return out

    def replicate(self, M: torch.Tensor) -> torch.Tensor:
        """Fills the replica rows of a compact operand whose primary rows (MPOS) are written
        (mrgcn_operand_replicate); a no-op on a plan without replicas."""
        if self.n_rep:
            assert M.is_cuda and M.is_contiguous() and M.shape[0] == self.nop
            with torch.cuda.device(M.device):
                L.check(L.load().mrgcn_operand_replicate(self.handle, M.data_ptr(), M.stride(0) * M.element_size(),
                                                         _stream_ptr(M.device)), "mrgcn_operand_replicate")
        return M

    def ulcol_long(self) -> torch.Tensor:
        """int64 device tensor: literal column r*N + j of every compact column."""
        t = getattr(self, "_ulcol_long", None)
        if t is None:
            t = torch.from_numpy(self.export(L.ARR_ULCOL).astype(np.int64)).to(self.device)
            self._ulcol_long = t
        return t

    def ulcol_sorted(self):
        """(rows, perm): the touched literal columns in rising order (int32) and the compact id of each (int32) — what
        mrgcn_scatter_rows_zero_fill_f32 walks.  Built once (one device sort), kept."""
        ent = self.__dict__.get("_ulcol_sorted")
        if ent is None:
            rows, perm = torch.sort(self.ulcol_long())
            ent = self.__dict__["_ulcol_sorted"] = (rows.to(torch.int32).contiguous(), perm.to(torch.int32).contiguous())
        return ent

    # -- per-node entry units of the wide-layer backward (mrgcn_wide_input_bwd_f32) ---------------------
    def wide_units(self, unit_entries: int = 256):
        """(erel, unit_node, unit_beg, unit_end, unit_multi, n_units): the plan's CSC entries cut into per-node units
        of at most `unit_entries` entries (a source node's entries are contiguous), built once and kept.  (Measured on
        the FB15k-237 epoch: 1.42 / 1.23 / 1.17 / 1.18 / 1.18 / 1.20 / 1.24 / 1.36 ms with units of 16 / 32 / 48 / 64 /
        96 / 128 / 256 / 512 entries — hub nodes stop being the tail of the launch.  Round 5 kept 256 because with 64
        the step oracle once saw 29 of 5.8 M elements of a replayed step outside their interval and blamed the order of
        the float atomics; round 6 found the cause — one hidden unit whose pre-activation is within 1e-10 of zero, its
        ReLU mask decided by rounding: `tests/test_gpu_step_oracle.py::_kink_mask` — and the reproducible form
        (`wide_units_det`) runs with 64.)"""
        ent = self.__dict__.get("_wide_units")
        if ent is None:
            nptr = self.export(L.ARR_NPTR).astype(np.int64)
            cptr = self.export(L.ARR_CPTR).astype(np.int64)
            eptr = cptr[nptr]                                   # entry range of every node
            n = np.diff(eptr)
            k = (n + unit_entries - 1) // unit_entries          # units per node (0 for a node without entries)
            node = np.repeat(np.arange(self.num_nodes), k)
            first = np.cumsum(k) - k
            idx = np.arange(int(k.sum())) - np.repeat(first, k)  # unit number inside its node
            beg = eptr[node] + idx * unit_entries
            end = np.minimum(beg + unit_entries, eptr[node + 1])
            dev = self.device
            erel = torch.empty((max(self.nnz, 1),), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                L.check(L.load().mrgcn_plan_entry_relations(self.handle, erel.data_ptr(), _stream_ptr(dev)),
                        "mrgcn_plan_entry_relations")
            up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(dev)  # noqa: E731
            ent = (erel, up(node, np.int32), up(beg, np.int32), up(end, np.int32), up(k[node] > 1, np.uint8),
                   int(len(node)))
            self.__dict__["_wide_units"] = ent
        return ent

    def wide_units_det(self, unit_entries: int = 64):
        """The units of the BITWISE-REPRODUCIBLE wide-layer backward (mrgcn_wide_input_bwd_det_f32): per-node units of at
        most `unit_entries` entries; a unit of a node with several units gets a slot in the hub workspace (`unit_slot`,
        -1 otherwise), `hub_node` / `hub_ptr` list those nodes and their slot ranges (a node's units, hence its slots,
        are consecutive).  Returns a dict, built once and kept."""
        ent = self.__dict__.get("_wide_units_det")
        if ent is None:
            nptr = self.export(L.ARR_NPTR).astype(np.int64)
            cptr = self.export(L.ARR_CPTR).astype(np.int64)
            eptr = cptr[nptr]                                   # entry range of every node
            n = np.diff(eptr)
            k = (n + unit_entries - 1) // unit_entries          # units per node (0 for a node without entries)
            node = np.repeat(np.arange(self.num_nodes), k)
            first = np.cumsum(k) - k
            idx = np.arange(int(k.sum())) - np.repeat(first, k)  # unit number inside its node
            beg = eptr[node] + idx * unit_entries
            end = np.minimum(beg + unit_entries, eptr[node + 1])
            multi = k[node] > 1
            slot = np.full(len(node), -1, dtype=np.int64)
            slot[multi] = np.arange(int(multi.sum()))
            hubs = np.flatnonzero(k > 1)
            hub_ptr = np.concatenate([[0], np.cumsum(k[hubs])])
            dev = self.device
            erel = torch.empty((max(self.nnz, 1),), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                L.check(L.load().mrgcn_plan_entry_relations(self.handle, erel.data_ptr(), _stream_ptr(dev)),
                        "mrgcn_plan_entry_relations")
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(dev)  # noqa: E731
            ent = dict(erel=erel, node=up(node), beg=up(beg), end=up(end), slot=up(slot), n_units=int(len(node)),
                       hub_node=up(hubs), hub_ptr=up(hub_ptr), n_hubs=int(len(hubs)), n_slots=int(multi.sum()), ws={})
            self.__dict__["_wide_units_det"] = ent
        return ent

    # -- gradient support (include/mrgcn_hip.h: mrgcn_support_*) --------------------------------
    def support_for(self, row_flags: torch.Tensor):
        """The gradient support of the output rows flagged in `row_flags` (uint8 [num_rows], device), built on first
        use and kept on the plan under the identity of the flags tensor — a STRUCTURAL row set (the labelled rows, or
        the node flags of the support of the layer above) is the same tensor, unchanged, every epoch.  None while a
        stream capture is under way and the support does not exist yet (the build synchronises): the caller then
        takes the per-epoch marking path."""
        if self.lean or row_flags is None:
            return None
        key = (row_flags.data_ptr(), row_flags._version, int(row_flags.numel()))
        cache = self.__dict__.setdefault("_supports", {})
        sup = cache.get(key)
        if sup is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            if row_flags.dtype != torch.uint8 or row_flags.numel() != self.num_rows or row_flags.device != self.device:
                return None
            sup = GraphSupport(self, row_flags)
            while len(cache) >= 4:  # (a handful of label sets per plan: train / valid / test)
                cache.pop(next(iter(cache))).close()
            cache[key] = sup
        return sup

    # -- algorithmic traffic ------------------------------------------------
    def spmm_bytes(self, F: int, value_bytes: int = 4, elem_bytes: int = 4) -> int:
        """nnz*(4+v) + (rows+1)*4 + ncols*F*e + rows*F*e"""
        return (self.nnz * (4 + value_bytes) + (self.num_rows + 1) * 4
                + self.ncols * F * elem_bytes + self.num_rows * F * elem_bytes)


class GraphSupport:
    """Host handle of a gradient support (mrgcn_support_t): the live columns / entries / nodes of a plan for a fixed
    set of output rows that can carry gradient.  Holds on to the flags tensor it was built from (its identity is the
    cache key) and to its plan."""

    def __init__(self, plan: GraphPlan, row_flags, forward: bool = False, _handle=None):
        """`forward`: also keep the flagged rows' forward arrays (MRGCN_SUPPORT_FORWARD: a mini-batch layer as a
        masked pass over the plan, csrc/masked.hip).  (`_handle`: adopt a support that exists — GraphSupport.chain.)"""
        import weakref
        lib = L.load()
        if _handle is None and (row_flags.dtype != torch.uint8 or row_flags.numel() != plan.num_rows
                                or not row_flags.is_contiguous()):
            raise L.MrgcnError(f"row_flags must be a contiguous uint8 tensor of {plan.num_rows} rows")
        # (a weak reference: the plan keeps its supports, not the other way round — no reference cycle, so both are
        # released by reference counting when the plan goes, not by a collector run at an arbitrary moment)
        self._plan = weakref.ref(plan)
        self.row_flags, self.device = row_flags, plan.device
        h = _handle
        if h is None:
            h = C.c_void_p()
            with torch.cuda.device(self.device):
                L.check(lib.mrgcn_support_create_ex(C.byref(h), plan.handle, row_flags.data_ptr(),
                                                    L.SUPPORT_FORWARD if forward else 0, _stream_ptr(self.device)),
                        "mrgcn_support_create")
        self._h = h
        info = L.SupportInfo()
        L.check(lib.mrgcn_support_info(self._h, C.byref(info)))
        self.L, self.E, self.NL = int(info.live_cols), int(info.live_entries), int(info.live_nodes)
        self.device_bytes = int(info.device_bytes)
        self.chunks_wide, self.chunks_narrow = int(info.chunks_wide), int(info.chunks_narrow)
        self.forward, self.NR = bool(forward), int(info.flagged_rows)
        # a forward support is a mini-batch's: it lives for one step on the stream that built it, and is released
        # under that stream's order, without a device-wide wait (mrgcn_support_destroy_ordered)
        self.ordered_release = bool(forward)
        self._node_flags = None
        self._ws = {}

    @classmethod
    def chain(cls, plan: GraphPlan, row_flags: torch.Tensor, n: int, forward: bool = True):
        """`n` supports in one build (one host wait): the first on `row_flags`, each next one on the NODE_FLAGS of the
        one before — the samples of the n layers of a mini-batch (mrgcn_support_create_chain)."""
        lib = L.load()
        if row_flags.dtype != torch.uint8 or row_flags.numel() != plan.num_rows or not row_flags.is_contiguous():
            raise L.MrgcnError(f"row_flags must be a contiguous uint8 tensor of {plan.num_rows} rows")
        hs = (C.c_void_p * n)()
        with torch.cuda.device(plan.device):
            L.check(lib.mrgcn_support_create_chain(hs, n, plan.handle, row_flags.data_ptr(),
                                                   L.SUPPORT_FORWARD if forward else 0, _stream_ptr(plan.device)),
                    "mrgcn_support_create_chain")
        out = []
        for i in range(n):
            # (level i+1 reads the flags array level i owns: it keeps that support alive through `row_flags`)
            out.append(cls(plan, row_flags if i == 0 else out[-1], forward, _handle=C.c_void_p(hs[i])))
        return out

    @property
    def plan(self):
        return self._plan()

    @property
    def handle(self):
        if self._h is None:
            raise L.MrgcnError("support already destroyed (its plan was closed)")
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _release("support_ordered" if getattr(self, "ordered_release", False) else "support", self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def array_ptr(self, which: int):
        p, n = C.c_void_p(), C.c_int64()
        L.check(L.load().mrgcn_support_array(self.handle, which, C.byref(p), C.byref(n)))
        return p.value or 0, int(n.value)

    def export(self, which: int) -> np.ndarray:
        ptr, n = self.array_ptr(which)
        dt = {L.SUP_COL_FLAGS: torch.uint8, L.SUP_NODE_FLAGS: torch.uint8, L.SUP_LVAL: torch.float32,
              L.SUP_FVAL: torch.float32}.get(which, torch.int32)
        out = torch.empty((n,), dtype=dt, device=self.device)
        if n:
            out = _device_array(ptr, n, dt, self.device).clone()
        return out.cpu().numpy()

    def node_flags(self) -> torch.Tensor:
        """uint8 [num_nodes]: the nodes that own a live column (a copy of the support's array, made once: an ordinary
        tensor with a lifetime of its own) — the row set of the layer below, and `row_cur` of the row-sparse Adam.
        The same tensor object every call: its identity keys the support of the layer below."""
        if self._node_flags is None:
            ptr, n = self.array_ptr(L.SUP_NODE_FLAGS)
            self._node_flags = _device_array(ptr, n, torch.uint8, self.device).clone()
        return self._node_flags

    def view(self, which: int) -> torch.Tensor:
        """One of the support's int32 arrays as a tensor over the library's memory (valid while the support lives)."""
        ptr, n = self.array_ptr(which)
        return _device_array(ptr, n, torch.int32, self.device)

    def workspace(self, key, numel: int) -> torch.Tensor:
        """A float32 scratch tensor kept on the support (one per use: the same buffer every epoch)."""
        t = self._ws.get(key)
        if t is None or t.numel() < numel:
            t = torch.empty((max(int(numel), 2),), dtype=torch.float32, device=self.device)
            self._ws[key] = t
        return t


def _device_array(ptr: int, n: int, dtype, device) -> torch.Tensor:
    """A tensor over `n` elements of device memory the library owns (no copy, no ownership)."""
    if n == 0:
        return torch.empty((0,), dtype=dtype, device=device)
    itemsize = torch.empty((), dtype=dtype).element_size()
    typestr = {torch.uint8: "|u1", torch.int32: "<i4", torch.float32: "<f4"}[dtype]

    class _Raw:
        __cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2,
                                    "strides": (itemsize,)}
    return torch.as_tensor(_Raw(), device=device)


def plan_of(A: torch.Tensor, num_nodes: int, num_relations: int, operand_row_bytes=None, lean: bool = False) -> GraphPlan:
    """Returns the plan cached on the adjacency tensor, building it on first use.  `operand_row_bytes` (the row
    sizes of the compact operands its users will multiply with: a layout hint, see GraphPlan) and `lean` (the quick
    build of a mini-batch slice) only matter to the call that builds the plan."""
    p = getattr(A, "_mrgcn_plan", None)
    if p is None or p._h is None or p.num_nodes != num_nodes or p.num_relations != num_relations:
        # (a slice of a short-lived batch carries the request itself: data/batch.py A_BatchDevice(short_lived=True))
        lean = lean or bool(getattr(A, "_mrgcn_lean", False))
        p = GraphPlan(A, num_nodes, num_relations, operand_row_bytes=operand_row_bytes, lean=lean)
        A._mrgcn_plan = p
    return p


_BUILD_STREAMS: dict = {}
_PLAN_THREADS = __import__("os").environ.get("MRGCN_PLAN_THREADS", "1") != "0"   # 0: one after the other on the caller's stream


def build_plans_parallel(jobs) -> None:
    """Builds the plans of several adjacency tensors at once: `jobs` = [(A, num_nodes, num_relations,
    operand_row_bytes[, lean]), ...]; those that carry a plan already are skipped.  A plan build is a chain of short device
    passes with host read-backs of sizes in between (a few ms of mostly waiting for a small slice); the builds of a
    re-sampled mini-batch's slices (two per layer) are independent, so each runs in its own host thread on its own
    stream — the waits overlap — and the caller's stream waits for all of them.  The C ABI is thread-safe per plan
    handle; ctypes releases the GIL for the duration of a call."""
    import threading
    todo = [j for j in jobs if getattr(j[0], "_mrgcn_plan", None) is None or j[0]._mrgcn_plan._h is None]
    if len(todo) <= 1 or not _PLAN_THREADS:
        for j in todo:
            plan_of(*j)
        return
    dev = todo[0][0].device
    cur = torch.cuda.current_stream(dev)
    # the same few streams every time: the stream-ordered pool hands a freed block back without a driver call only to
    # the stream that freed it
    pool = _BUILD_STREAMS.setdefault((dev, threading.get_ident()), [])   # (per calling thread: batch prefetch workers)
    while len(pool) < len(todo):
        pool.append(torch.cuda.Stream(device=dev))
    streams = pool[: len(todo)]
    errors = []

    def work(job, st):
        try:
            with torch.cuda.device(dev), torch.cuda.stream(st):
                plan_of(*job)
        except BaseException as e:  # noqa: BLE001  (re-raised on the caller's thread)
            errors.append(e)

    threads = []
    for job, st in zip(todo, streams):
        st.wait_stream(cur)
        t = threading.Thread(target=work, args=(job, st), daemon=True)
        t.start()
        threads.append(t)
    for t in threads:
        t.join()
    for st in streams:
        cur.wait_stream(st)
    if errors:
        raise errors[0]
