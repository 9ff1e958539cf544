"""autograd wiring of the HIP kernels.  Every op here calls the C ABI; none has a PyTorch
or CPU fallback."""
from __future__ import annotations

import os

import torch

from . import _lib as L
from .plan import GraphPlan
from .stats import bump


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# measured on the AM shape: 14.6 ms with the overlap vs 14.3 ms without (every one of these
# kernels already saturates the memory system on its own), so it is opt-in
_OVERLAP = os.environ.get("MRGCN_OVERLAP", "1") != "0"  # AM epoch 6.72-6.76 -> 6.67-6.68 ms (same box, alternating runs)
# Skip the compact columns without gradient in the backward (exact: they add zeros).
_LIVE_COLS = os.environ.get("MRGCN_LIVE_COLS", "1") != "0"

# The backward on a gradient support (plan.GraphPlan.support_for): when the rows of a layer's output gradient that can
# hold anything are known structurally — the labelled rows, and below them the nodes of the upper layer's support —
# the live columns, their kept entries and the relation-major lists are built once and every epoch runs on dense index
# spaces (csrc/support.hip).  MRGCN_SUPPORT=0: the per-epoch marking path for every backward (A/B).
_SUPPORT = os.environ.get("MRGCN_SUPPORT", "1") != "0"
# the wide featureless layer's backward in its bitwise-reproducible form (64-entry units, no float atomics); 0: the
# round-5 form (256-entry units, atomics for hub nodes)
_WIDE_DET = os.environ.get("MRGCN_WIDE_DET", "1") != "0"
_WIDE_UNIT = int(os.environ.get("MRGCN_WIDE_UNIT", "64"))   # entries per unit of the reproducible form
# bf16 layers read a wide input as bf16 rows (0: only the compact operand M is bf16, the round-5 form — the A/B switch)
_BF16_PIPELINE = os.environ.get("MRGCN_BF16_PIPELINE", "1") != "0"

# tests: start dM as NaNs, so that any read of a row the producer left unwritten shows
_POISON_DEAD = False

# Row-sparse gradient of a node-major weight_I (train_step turns it on for ClipAdam without weight decay):
# nodes without a live compact column have a zero gradient block, and with a fixed label set never any
# other — their Adam moments stay zero and their parameters never move.  The backward then leaves those
# blocks unwritten (and their V blocks unread) and hands the gradient to the optimizer through
# `weight_I._mrgcn_rows` (buffer, per-node flags, squared norm) instead of `weight_I.grad`; ClipAdam runs
# mrgcn_adam_step_rows_f32, which never touches a node that never had gradient.  The state hangs on the
# Parameter object itself.  Outside train_step (plain `loss.backward()`) the gradient is dense and arrives
# in `weight_I.grad` as usual.
# The same happens under a plain `loss.backward()` when the parameter's optimizer announced that it reads the
# row-sparse form (mrgcn_amd.optim.Adam registers itself as `weight_I._mrgcn_row_consumer`: the reference's own loop,
# tasks/node_classification.py:190-193, then runs the fast path with `optim.Adam` / `clip_grad_norm_` swapped).
# None = decided per parameter (consumer registered?), True / False = forced (train_step).
_ROW_SPARSE = None
_ROW_SPARSE_ENV = os.environ.get("MRGCN_ROW_SPARSE", "1") != "0"


def row_sparse_weight_grad(enabled):
    """Forces the row-sparse form on / off (None: per parameter).  Returns the previous setting."""
    global _ROW_SPARSE
    prev, _ROW_SPARSE = _ROW_SPARSE, (None if enabled is None else bool(enabled))
    return prev


def _row_sparse_for(param) -> bool:
    if not _ROW_SPARSE_ENV:
        return False
    if _ROW_SPARSE is not None:
        return _ROW_SPARSE
    consumer = getattr(param, "_mrgcn_row_consumer", None)
    return consumer is not None and consumer() is not None


def dense_from_rows(param: torch.Tensor, ent) -> torch.Tensor:
    """The dense gradient a row-sparse entry stands for (zeros for the nodes without gradient): for the rare step
    that needs it after all — another term (a weight regulariser) put a dense gradient on the same parameter."""
    lib = L.load()
    if ent.get("kind") == "index":   # compact rows of a literal operand (_SpmmLiteral)
        g = torch.zeros_like(param)
        g.index_copy_(0, ent["index"], ent["g"])
        return g
    fz = ent.get("fused")
    if fz is None:
        return torch.where(ent["cur"].bool().view(-1, *([1] * (param.dim() - 1))), ent["g"], torch.zeros_like(ent["g"]))
    g = torch.empty_like(param)
    d_comp = torch.empty_like(fz["comp"])
    if fz.get("sup") is not None:
        with torch.cuda.device(param.device):
            L.check(lib.mrgcn_support_mix_bwd_f32(
                fz["sup"].handle, fz["dM"].data_ptr(), fz["ld"], param.data_ptr(), fz["comp"].data_ptr(), fz["B"],
                fz["F"], g.data_ptr(), 1, d_comp.data_ptr(), 0, 0, 0, _stream(param.device)),
                "mrgcn_support_mix_bwd_f32")
        return g
    with torch.cuda.device(param.device):
        L.check(lib.mrgcn_basis_mix_bwd_f32(
            fz["plan"].handle, fz["dM"].data_ptr(), fz["ld"], fz["live"].data_ptr(), param.data_ptr(),
            fz["comp"].data_ptr(), fz["B"], fz["F"], g.data_ptr(), 0, d_comp.data_ptr(), 0, _stream(param.device)),
            "mrgcn_basis_mix_bwd_f32")
    return g


def pop_row_grad(param: torch.Tensor):
    """The row-sparse gradient the last backward left on `param`, or None; consumed by the call."""
    ent = getattr(param, "_mrgcn_rows", None)
    if ent is None or not ent["fresh"]:
        return None
    ent["fresh"] = False
    return ent


def clear_row_grads(params):
    """Gradients left by a backward whose optimizer step never came must not survive into the next."""
    for p in params:
        ent = getattr(p, "_mrgcn_rows", None)
        if ent is not None:
            ent["fresh"] = False


class _LiveGauge:
    """How many rows of a layer's output gradient held anything the last time it was looked at.
    The sparse transposed product (mrgcn_spmm_transposed_live_f32) wins while few rows are live
    (semi-supervised: a handful of labelled nodes) and loses to the general one when most are
    (AM shape: 211 vs 467 us at 0.3 %, 860 vs 467 us at 100 %).  The count travels device ->
    pinned host without a synchronisation, so the choice lags one step; when the general product
    is in use the sparse one is tried again every `_RETRY` calls to refresh the count."""
    _RETRY = 32
    _DENSE = 0.25

    _pinned = None  # one pinned page for all gauges (pinning per gauge would cost ~100 us each)
    _free: list = []

    def __init__(self, num_rows, dev):
        cls = _LiveGauge
        if cls._pinned is None:
            cls._pinned = torch.full((1024,), -1, dtype=torch.int32).pin_memory()
            cls._free = list(range(1024))
        self.num_rows = num_rows
        self.dev = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.slot = cls._free.pop() if cls._free else None
        if self.slot is not None:
            cls._pinned[self.slot] = -1
            self.host = cls._pinned[self.slot:self.slot + 1]
        else:  # out of slots: a private, unpinned count that never learns anything (always sparse)
            self.host = torch.full((1,), -1, dtype=torch.int32)
        self.calls = 0

    def __del__(self):
        if getattr(self, "slot", None) is not None and _LiveGauge is not None:  # (None at interpreter exit)
            _LiveGauge._free.append(self.slot)
            self.slot = None

    def sparse(self) -> bool:
        self.calls += 1
        last = int(self.host[0])  # pinned host memory: no synchronisation
        return last < 0 or last <= self._DENSE * self.num_rows or self.calls % self._RETRY == 0

    def publish(self):
        if self.slot is not None:
            self.host.copy_(self.dev, non_blocking=True)


def _live_gauge(plan, F, relu, dev):
    """The gauge of (plan, layer width, activation): kept on the plan object, dies with it."""
    gauges = plan.__dict__.setdefault("_gauges", {})
    g = gauges.get((F, bool(relu)))
    if g is None:
        g = gauges[(F, bool(relu))] = _LiveGauge(plan.num_rows, dev)
    return g


_SIDE_STREAMS: dict = {}


def _side_stream(device) -> torch.cuda.Stream:
    st = _SIDE_STREAMS.get(device)
    if st is None:
        # MRGCN_SIDE_PRIORITY (default 0 = the current stream's): -1 lets the side stream's kernels (the transforms'
        # dW / dX) take compute units ahead of the mix backward they run beside
        st = torch.cuda.Stream(device=device, priority=int(os.environ.get("MRGCN_SIDE_PRIORITY", "0")))
        _SIDE_STREAMS[device] = st
    return st


def _ld_for(F: int) -> int:
    """Leading dimension of the fp32 compact operand.  Layers of 4 <= F <= 16 outputs with F not a multiple of
    four keep PACKED rows (ld = F: 40-byte rows at F = 10): the product's gathers stay single 16-byte loads — the
    row's last vector overlaps its neighbour's (k_spmm3 `pack`) — and the first-touch stream of the operand
    shrinks by the pad (AM shape: 158 -> 148 us with the plan's operand order hinted for 40-byte rows).  Other
    widths: rows padded to a multiple of 4 floats.  MRGCN_LDM_ALIGN overrides (floats: 4 = always padded)."""
    a = os.environ.get("MRGCN_LDM_ALIGN")
    if a is None:
        return F if (F % 4 and 4 <= F <= 16) else (F + 3) // 4 * 4
    a = int(a)
    return (F + a - 1) // a * a


def operand_row_bytes(F: int, bf16: bool = False) -> int:
    """Row size of a layer's compact operand — the hint GraphPlan takes."""
    return 2 * _ld_for_bf16(F) if bf16 else 4 * _ld_for(F)


def _ld_for_bf16(F: int) -> int:
    """bf16 operand rows.  Narrow layers whose F is even and not a multiple of four keep PACKED rows (ld = F: 20-byte
    rows at F = 10, 8-byte gathers with the row's last vector overlapping its neighbour's — k_spmm3 `pack`, as for the
    fp32 operand); other widths are padded to a multiple of MRGCN_LDM_BF16_ALIGN elements (4: 8-byte gathers, rows no
    wider than the fp32 operand's; measured 0.15 ms per AM epoch better than 8).  MRGCN_LDM_BF16_ALIGN set: always padded."""
    a = os.environ.get("MRGCN_LDM_BF16_ALIGN")
    if a is None:
        if 4 <= F <= 16 and F % 4 and F % 2 == 0:
            return F
        a = "4"
    a = int(a)
    return (F + a - 1) // a * a


# ---- the bf16 pipeline's layer input ---------------------------------------------------------------------------------
# BASELINE config 3 / SURVEY §8d: activations stored in bf16, fp32 accumulation.  A layer whose `operand_dtype` is
# "bf16" reads a WIDE input (K > 16) as bf16 rows padded to whole 16-byte pieces: half the bytes per gathered row in the
# transform and in its dW.  An input that is DATA (no gradient: the feature matrix of a run without encoders, the same
# tensor every epoch) is converted once and the copy kept while the tensor object lives unchanged — like the graph
# plan, preparation of a constant input; an input that is an activation (it requires grad, or its version moves) is
# converted by every forward.  A bf16 tensor is taken as it is.
_XB_CACHE = {}   # id(X) -> (weakref(X), version, data_ptr, Xb)


def bf16_rows(X: torch.Tensor) -> torch.Tensor:
    """[n, ld] bf16 copy of the fp32 rows X[n, K], ld = K rounded up to 8 elements, zeros past K."""
    if X.dtype == torch.bfloat16:
        if X.dim() != 2 or X.stride(1) != 1 or X.stride(0) % 8 or X.data_ptr() % 16:
            raise L.MrgcnError("a bf16 layer input needs rows of whole 16-byte pieces")
        return X
    import weakref
    # (kept only when made outside a capture: a tensor allocated inside one lives in the graph's pool)
    cacheable = not X.requires_grad and not torch.cuda.is_current_stream_capturing()
    if not X.requires_grad:
        ent = _XB_CACHE.get(id(X))
        if ent is not None and ent[0]() is X and ent[1] == X._version and ent[2] == X.data_ptr():
            bump("bf16.x_cached")
            return ent[3]
    Xc = X if (X.dim() == 2 and X.stride(1) == 1) else X.contiguous()
    n, K = Xc.shape
    ld = (K + 7) // 8 * 8
    Xb = torch.empty((n, ld), dtype=torch.bfloat16, device=X.device)
    with torch.cuda.device(X.device):
        L.check(L.load().mrgcn_cast_rows_bf16(Xc.data_ptr(), Xc.stride(0), n, K, Xb.data_ptr(), ld, _stream(X.device)),
                "mrgcn_cast_rows_bf16")
    bump("bf16.x_cast")
    if cacheable:
        for k in [k for k, e in _XB_CACHE.items() if e[0]() is None]:
            del _XB_CACHE[k]
        _XB_CACHE[id(X)] = (weakref.ref(X), X._version, X.data_ptr(), Xb)
    return Xb


_XPAD_CACHE = {}   # id(X) -> (weakref(X), version, data_ptr, padded copy)
_XPAD = os.environ.get("MRGCN_X_LINE_ROWS", "1") != "0"


def line_aligned_rows(X: torch.Tensor) -> torch.Tensor:
    """A wide fp32 layer input that is DATA (no gradient; the same tensor every epoch) whose rows are not whole 128-byte
    lines, as a view of a copy with rows padded to whole lines (K = 155: 620 -> 640 bytes) — made once, kept while the
    tensor object lives unchanged, like the plan and the bf16 copy.  The relation transforms gather whole rows: an
    unaligned 620-byte row touches 5.9 lines on average, an aligned one 5 (layer-0 transform 875 -> 834 us, its dW
    945 -> 873 at the AM shape).  Anything else is returned as it is."""
    if (not _XPAD or X.dtype != torch.float32 or X.requires_grad or X.dim() != 2 or X.shape[1] < 64
            or X.stride(1) != 1 or (X.stride(0) * 4) % 128 == 0 or not X.is_cuda):
        return X
    import weakref
    ent = _XPAD_CACHE.get(id(X))
    if ent is not None and ent[0]() is X and ent[1] == X._version and ent[2] == X.data_ptr():
        return ent[3]
    if torch.cuda.is_current_stream_capturing():
        return X   # (a copy made inside a capture would live in the graph's pool: the first eager step makes it)
    n, K = X.shape
    ld = (K + 31) // 32 * 32
    buf = torch.zeros((n, ld), dtype=torch.float32, device=X.device)
    buf[:, :K].copy_(X)
    Xp = buf[:, :K]
    bump("x_line_rows.copy")
    for k in [k for k, e in _XPAD_CACHE.items() if e[0]() is None]:
        del _XPAD_CACHE[k]
    _XPAD_CACHE[id(X)] = (weakref.ref(X), X._version, X.data_ptr(), Xp)
    return Xp


class _SpmmLiteral(torch.autograd.Function):
    """Y = A . D with D the dense (R*N) x F operand in the reference's row order
    (mrgcn/layers/graph.py:75, :95).  Backward = A^T dY scattered into a dense
    (R*N) x F gradient, as SparseAddmmBackward produces for the reference."""

    @staticmethod
    def forward(ctx, plan: GraphPlan, D: torch.Tensor, bias, relu: bool, owner=None):
        # `owner`: the layer whose `weight_I` parameter IS the operand (a featureless layer without bases): its
        # gradient may then travel in compact form (below)
        ctx.param = None
        if owner is not None and D is getattr(owner, "weight_I", None) and D.is_contiguous():
            ctx.param = D
        D = D.contiguous()
        Y = plan.spmm(L.VIEW_LITERAL, D, bias=bias, relu=relu)
        ctx.plan, ctx.relu, ctx.has_bias = plan, relu, bias is not None
        ctx.d_rows = D.shape[0]
        ctx.save_for_backward(Y if relu else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        plan = ctx.plan
        dY = dY.contiguous()
        (Y,) = ctx.saved_tensors
        # (the layer above may hand its input gradient on already masked: `_grad_meta`, as for _RgcnLayer)
        meta = _grad_meta(dY)
        if meta and meta.get("sparse_rows"):
            raise L.MrgcnError("internal: an output gradient with unwritten rows reached the literal product")
        if ctx.relu and not (meta and meta["relu_applied"]):
            dY = relu_bwd(dY, Y)
        dbias = _bias_grad(dY) if ctx.has_bias else None
        dD = None
        if ctx.needs_input_grad[1]:
            F = dY.shape[1]
            param = ctx.param
            rows = getattr(param, "_mrgcn_rows", None) if param is not None else None
            if rows is not None and rows.get("kind") == "index" and rows.get("plan") is not plan:
                # another adjacency (mini-batches, a second graph): other compact columns — rows outside THIS set may
                # hold moments that must keep decaying: dense gradients for this parameter from now on
                rows["dense_only"] = True
            compact = (param is not None and F % 4 == 0 and plan.ncols > 0 and not plan.lean and _row_sparse_for(param)
                       # (moments outside the compact columns: the optimizer asked for dense gradients from now on)
                       and not (rows is not None and rows.get("kind") == "index" and rows.get("dense_only")))
            if compact:
                # The rows of the (R*N) x F operand that are columns of A — the plan's compact columns — are the only
                # ones this product's autograd ever gives gradient to, whatever the labels: the gradient stays in
                # compact order ([ncols, F], no zero fill of the table) and the consumer on the parameter
                # (ClipAdam: mrgcn_adam_step_index_rows_f32) updates those rows only.  `.grad` stays None.
                if rows is not None and rows["fresh"]:
                    raise L.MrgcnError("row-sparse weight_I gradient: the layer ran twice in one train_step "
                                       "(use train_step(..., row_sparse=False))")
                if rows is None or rows.get("kind") != "index":
                    if torch.cuda.is_current_stream_capturing() and getattr(plan, "_ulcol_long", None) is None:
                        raise L.MrgcnError("the literal column list of this plan is built on first use: run one "
                                           "backward of the layer before capturing it")
                    rows = dict(kind="index", plan=plan, shape=tuple(param.shape), g=None, index=plan.ulcol_long(),
                                index_ptr=plan.array_ptr(L.ARR_ULCOL)[0], fresh=False, seeded_for=None, dense_only=False)
                    param._mrgcn_rows = rows
                g = torch.empty((plan.ncols, F), dtype=torch.float32, device=dY.device)
                plan.spmm(L.VIEW_TRANSPOSED, dY, out=g)
                rows["g"], rows["fresh"] = g, True
                bump("weight_I.index_rows")
                return None, None, dbias, None, None
            dD = torch.zeros((ctx.d_rows, F), dtype=torch.float32, device=dY.device)
            ulcol_ptr, _ = plan.array_ptr(L.ARR_ULCOL)
            plan.spmm(L.VIEW_TRANSPOSED, dY, out=dD, out_index=ulcol_ptr)
        return None, dD, dbias, None, None


def spmm_literal(plan: GraphPlan, D: torch.Tensor, bias=None, relu: bool = False, owner=None) -> torch.Tensor:
    return _SpmmLiteral.apply(plan, D, bias, relu, owner)


def relu_bwd(dY: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    """dY * (Y > 0); Y may be the first F columns of a buffer with padded rows (plan.spmm)."""
    out = torch.empty(dY.shape, dtype=torch.float32, device=dY.device)
    with torch.cuda.device(dY.device):
        if dY.is_contiguous() and Y.is_contiguous():
            L.check(L.load().mrgcn_relu_bwd_f32(dY.data_ptr(), Y.data_ptr(), dY.numel(), out.data_ptr(),
                                                _stream(dY.device)), "mrgcn_relu_bwd_f32")
        else:
            assert dY.dim() == 2 and dY.stride(1) == 1 and Y.stride(1) == 1 and Y.shape == dY.shape
            L.check(L.load().mrgcn_relu_bwd_rows_f32(dY.data_ptr(), dY.stride(0), Y.data_ptr(), Y.stride(0),
                                                     dY.shape[0], dY.shape[1], out.data_ptr(), out.stride(0),
                                                     _stream(dY.device)), "mrgcn_relu_bwd_rows_f32")
    return out


# Side channel between the backward passes of two stacked layers.  The upper layer's backward returns its input
# gradient with a note attached to the tensor object: the ReLU mask of the lower layer's output is already applied,
# and a byte per row says which rows hold anything.  The lower layer's backward trusts the note only while the tensor
# is the very object, unchanged, that the upper layer returned: autograd hands a sole consumer's gradient through as
# it is, and sums the gradients of several consumers either into a new tensor (no note) or in place (the tensor's
# version counter moves) — in both cases the lower layer falls back to masking and scanning by itself.
def _bias_grad(dY: torch.Tensor, row_flags=None) -> torch.Tensor:
    """`dY.sum(0)`: the gradient of a layer's bias (autograd of `AFW + self.b`, graph.py:98-101).  Narrow layers take
    one pass of this package (fixed summation order; rows flagged 0 are not read — a gradient whose unflagged rows
    were never written is fine); wide ones torch's reduction."""
    F = dY.shape[1]
    if dY.is_cuda and dY.dtype == torch.float32 and F <= 16 and dY.stride(1) == 1 and dY.shape[0] > 0:
        lib = L.load()
        out = torch.empty(F, dtype=torch.float32, device=dY.device)
        ws = torch.empty(int(lib.mrgcn_colsum_rows_workspace(F)), dtype=torch.float32, device=dY.device)
        with torch.cuda.device(dY.device):
            L.check(lib.mrgcn_colsum_rows_f32(dY.data_ptr(), dY.stride(0), dY.shape[0], F,
                                              row_flags.data_ptr() if row_flags is not None else 0, out.data_ptr(),
                                              ws.data_ptr(), ws.numel(), _stream(dY.device)), "mrgcn_colsum_rows_f32")
        return out
    if row_flags is not None:
        # unflagged rows may never have been written: select, do not multiply (NaN * 0 is NaN)
        return torch.where(row_flags.bool()[:, None], dY, torch.zeros((), dtype=dY.dtype, device=dY.device)).sum(0)
    return dY.sum(0)


def _set_grad_meta(t, row_live, relu_applied: bool, structural: bool = False, sparse_rows: bool = False):
    """`structural`: `row_live` is a row set fixed by the label set and the graph (the same tensor every epoch; rows
    outside it are certainly zero, rows inside it may be): the key of a gradient support.  `sparse_rows`: the rows
    outside `row_live` were not even written — only a consumer that goes by the flags may read this tensor."""
    t._mrgcn_grad_meta = {"version": t._version, "row_live": row_live, "relu_applied": bool(relu_applied),
                          "structural": bool(structural), "sparse_rows": bool(sparse_rows)}


def _grad_meta(t):
    m = getattr(t, "_mrgcn_grad_meta", None)
    return m if (m is not None and m["version"] == t._version) else None


class _RgcnLayer(torch.autograd.Function):
    """Y = relu?( A' . M + b ),  M[c] = comp_I[r_c] . V_I[j_c, :, :]  (or weight_I[r_c*N + j_c])
                                       + X[j_c] . W_F[r_c]
    i.e. graph.py:62-102 without the (R*N) x out intermediates.  `weight_I` with bases is the layer's
    node-major (N, B, out) parameter; without bases the reference's (R*N, out)."""

    @staticmethod
    def forward(ctx, plan: GraphPlan, F: int, weight_I, comp_I, X, W_F, bias, relu: bool, bf16: bool = False,
                owner=None):
        lib = L.load()
        dev = plan.device
        # bf16: the compact operand M is stored in bf16 (one rounding at its store); a WIDE input (K > 16) is read as
        # bf16 rows by the transform on v_mfma_f32_16x16x32_bf16 and the feature term's rows travel in bf16 too (the
        # bf16 pipeline, `bf16_rows`); parameters, every accumulation, Y and the backward's sums stay fp32
        ld = _ld_for_bf16(F) if bf16 else _ld_for(F)
        M = torch.empty((plan.nop, ld), dtype=torch.bfloat16 if bf16 else torch.float32, device=dev)
        sfx = "bf16" if bf16 else "f32"
        xform_fwd = getattr(lib, "mrgcn_rel_transform_fwd_" + sfx)
        mix_fwd = getattr(lib, "mrgcn_basis_mix_fwd_" + sfx)
        gather_rows = getattr(lib, "mrgcn_gather_rows_" + sfx)
        s = _stream(dev)
        Xc = Wc = Xb = None
        with torch.cuda.device(dev):
            addend, ldA, addend_bf16 = 0, 0, False
            if X is not None:
                K = X.shape[1]
                Wc = W_F.contiguous()
                if bf16 and (K > 16 or X.dtype == torch.bfloat16) and _BF16_PIPELINE:
                    ldo = 16 if weight_I is not None else ld
                    if lib.mrgcn_rel_transform_xbf16_supported(plan.handle, K, F, (K + 7) // 8 * 8, ldo):
                        Xb = bf16_rows(X)
                if Xb is None and X.dtype != torch.float32:
                    raise L.MrgcnError("a bf16 layer input needs operand_dtype 'bf16' and a shape the bf16 transform takes")
                Xc = X if (X.dim() == 2 and X.stride(1) == 1) else X.contiguous()  # row-strided X is taken as it is
                if Xb is None:
                    Xc = line_aligned_rows(Xc)   # (a constant feature matrix: rows of whole 128-byte lines, copied once)
                if weight_I is not None:
                    # feature term in plain compact order (sequential writes); the input-term
                    # pass below adds it while it emits the final rows in operand order
                    addend_bf16 = (Xb is not None and comp_I is not None and comp_I.shape[1] <= 64
                                   and (comp_I.shape[1] * F) % 4 == 0)
                    ldA = 16 if addend_bf16 else (F + 3) // 4 * 4
                    M2 = torch.empty((plan.ncols, ldA), dtype=torch.bfloat16 if addend_bf16 else torch.float32,
                                     device=dev)
                    out, ldo, order = M2, ldA, 0
                    addend = M2.data_ptr()
                else:
                    out, ldo, order = M, ld, 1
                if Xb is not None:
                    bump("bf16.xform_xbf16")
                    L.check(lib.mrgcn_rel_transform_fwd_xbf16(plan.handle, Xb.data_ptr(), Xb.stride(0), K, Wc.data_ptr(), F,
                                                              out.data_ptr(), ldo, order,
                                                              int(out.dtype == torch.bfloat16), s),
                            "mrgcn_rel_transform_fwd_xbf16")
                else:
                    fwd = xform_fwd if out is M else lib.mrgcn_rel_transform_fwd_f32  # M2 stays fp32
                    L.check(fwd(plan.handle, Xc.data_ptr(), Xc.stride(0), Xc.shape[1], Wc.data_ptr(), F,
                                out.data_ptr(), ldo, order, s), "mrgcn_rel_transform_fwd_" + sfx)
            if weight_I is not None:
                wI = weight_I.contiguous()
                if comp_I is not None:
                    cI = comp_I.contiguous()
                    if addend_bf16:
                        L.check(lib.mrgcn_basis_mix_fwd_abf16(plan.handle, wI.data_ptr(), cI.data_ptr(), cI.shape[1], F,
                                                              addend, ldA, M.data_ptr(), ld, 1, s),
                                "mrgcn_basis_mix_fwd_abf16")
                    else:
                        L.check(mix_fwd(plan.handle, wI.data_ptr(), cI.data_ptr(), cI.shape[1], F, addend, ldA,
                                        M.data_ptr(), ld, s), "mrgcn_basis_mix_fwd_" + sfx)
                else:
                    L.check(gather_rows(plan.handle, wI.data_ptr(), F, addend, ldA, M.data_ptr(), ld, s),
                            "mrgcn_gather_rows_" + sfx)
        plan.replicate(M)  # (plans with operand replicas only)
        # a hidden layer's output stays in rows padded to whole 16-byte pieces (the product stores whole pieces: -5 %);
        # an output the caller sees is dense like the reference's
        Y = plan.spmm(L.VIEW_COMPACT, M, F=F, bias=bias, relu=relu,
                      padded_rows=bool(getattr(owner, "padded_output", False)))
        ctx.plan, ctx.F, ctx.ld, ctx.relu, ctx.owner = plan, F, ld, relu, owner
        ctx.Xb = Xb   # the input's bf16 rows (the bf16 pipeline): what the backward's dW gathers
        # the layer's input is the output of a fused ReLU (marked by the layer that made it): its own sign is that
        # ReLU's mask, so this layer's backward can hand its input gradient on already masked
        ctx.x_is_relu_out = X is not None and bool(getattr(X, "_mrgcn_relu_out", False))
        ctx.has = (weight_I is not None, comp_I is not None, X is not None, bias is not None)
        ctx.save_for_backward(weight_I, comp_I, Xc, Wc, Y if relu else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = L.load()
        plan, F = ctx.plan, ctx.F
        weight_I, comp_I, X, W_F, Y = ctx.saved_tensors
        has_I, has_comp, has_X, has_bias = ctx.has
        dev = plan.device
        s = _stream(dev)
        dY = dY.contiguous()
        # what the layer above knows about this gradient (see `_grad_meta`): its ReLU mask is applied already, and
        # which of its rows hold anything
        meta = _grad_meta(dY)
        row_flags = meta["row_live"] if meta else None
        if ctx.relu and not (meta and meta["relu_applied"]):
            dY = relu_bwd(dY, Y)
            if not (meta and meta.get("structural")):
                row_flags = None   # (a structural row set stays a valid superset under the mask; a scanned one is stale)
        sparse_rows = bool(meta and meta.get("sparse_rows"))
        if sparse_rows and row_flags is None:
            raise L.MrgcnError("internal: an output gradient with unwritten rows arrived without its row flags")
        # (rows outside the flags are zeros — or, with `sparse_rows`, unwritten: the flagged rows are all there is to add)
        dbias = _bias_grad(dY, row_flags) if has_bias else None
        if meta is None and _SUPPORT and _LIVE_COLS and _DISCOVER and F <= 16:
            # a plain dense gradient (the reference's own loss: CrossEntropyLoss on Y_hat[idx] leaves zeros + the
            # labelled rows): which rows hold anything is looked up, and while that set equals last epoch's the
            # backward runs on the support built for it
            meta = _discovered_rows(ctx.owner, plan, dY, F, dev)
            if meta:
                bump("discovered_rows")
            row_flags = meta["row_live"] if meta else row_flags
        sup = _support_of(plan, meta, F, dev) if (_SUPPORT and _LIVE_COLS) else None
        if sup is not None:
            out = _RgcnLayer._backward_on_support(ctx, sup, dY, dbias)
            if out is not None:
                bump("backward.support")
                return out
        # (the units of the wide-layer backward are built on first use, with host round trips: not inside a capture)
        if (has_I and has_comp and not has_X and not sparse_rows and weight_I.dim() == 3 and not plan.lean
                and dY.stride(0) % 4 == 0
                and (("_wide_units_det" if _WIDE_DET else "_wide_units") in plan.__dict__
                     or not torch.cuda.is_current_stream_capturing())):
            Bn = weight_I.shape[1]
            param = getattr(ctx.owner, "weight_I", None)
            if (lib.mrgcn_wide_input_bwd_supported(plan.handle, Bn, F)
                    and not (param is not None and _row_sparse_for(param) and _LIVE_COLS)):
                # a wide featureless layer with few bases (the link-prediction encoder): dV and dcomp straight from
                # dY over the plan's entries — the 4 F-byte rows of dM are never written (csrc/wide_input.hip)
                wI = weight_I.contiguous()
                d_wI, d_comp = torch.empty_like(wI), torch.empty_like(comp_I)
                if _WIDE_DET:
                    # units of 64 entries, no float atomics: bitwise reproducible (plan.wide_units_det)
                    u = plan.wide_units_det(_WIDE_UNIT)
                    nws = int(lib.mrgcn_wide_input_bwd_det_workspace(plan.handle, u["n_slots"], Bn, F))
                    ws = u["ws"].get((Bn, F))
                    if ws is None or ws.numel() < nws:
                        ws = u["ws"][(Bn, F)] = torch.empty((max(nws, 4),), dtype=torch.float32, device=dev)
                    with torch.cuda.device(dev):
                        L.check(lib.mrgcn_wide_input_bwd_det_f32(
                            plan.handle, u["erel"].data_ptr(), u["node"].data_ptr(), u["beg"].data_ptr(),
                            u["end"].data_ptr(), u["slot"].data_ptr(), u["n_units"], u["hub_node"].data_ptr(),
                            u["hub_ptr"].data_ptr(), u["n_hubs"], u["n_slots"], dY.data_ptr(), dY.stride(0), wI.data_ptr(),
                            comp_I.contiguous().data_ptr(), Bn, F, d_wI.data_ptr(), d_comp.data_ptr(), ws.data_ptr(),
                            ws.numel(), s), "mrgcn_wide_input_bwd_det_f32")
                else:
                    erel, un, ub, ue, um, nu = plan.wide_units()
                    with torch.cuda.device(dev):
                        L.check(lib.mrgcn_wide_input_bwd_f32(
                            plan.handle, erel.data_ptr(), un.data_ptr(), ub.data_ptr(), ue.data_ptr(), um.data_ptr(), nu,
                            dY.data_ptr(), dY.stride(0), wI.data_ptr(), comp_I.contiguous().data_ptr(), Bn, F,
                            d_wI.data_ptr(), d_comp.data_ptr(), s), "mrgcn_wide_input_bwd_f32")
                bump("backward.wide_input")
                return None, None, d_wI, d_comp, None, None, dbias, None, None, None
        # dM = A'^T dY over touched columns only, plain compact order (its consumers are node-major)
        ld = (F + 3) // 4 * 4
        dM = torch.empty((plan.ncols, ld), dtype=torch.float32, device=dev)
        if _POISON_DEAD:
            dM.fill_(float("nan"))
        live = node_live = None
        gauge = _live_gauge(plan, F, ctx.relu, dev) if _LIVE_COLS else None
        if sparse_rows and (gauge is None or F > 16):
            raise L.MrgcnError("internal: an output gradient with unwritten rows needs the live-row backward")
        # (unwritten rows outside the flags: only the live-row form may read this gradient, whatever the gauge says)
        if gauge is not None and (sparse_rows or (F <= 16 and gauge.sparse())):
            # with few labelled nodes most rows of dY are zeros: gather the others only, and keep one
            # byte per compact column: does it carry any gradient?
            live = torch.empty((plan.ncols,), dtype=torch.uint8, device=dev)
            # (a layer whose input wants a gradient also gets a flag per source node: its dX pass skips by it)
            node_live = (torch.empty((plan.num_nodes,), dtype=torch.uint8, device=dev)
                         if (has_X and ctx.needs_input_grad[4]) else None)
            scratch = torch.empty((int(lib.mrgcn_spmm_transposed_live_scratch(plan.handle)),),
                                  dtype=torch.uint8, device=dev)
            # rows of dM without gradient are not even written when every consumer goes by the flags
            # (the no-bases scatter reads dM itself)
            write_dead = int(has_I and not has_comp)
            if row_flags is not None and (row_flags.numel() != plan.num_rows or row_flags.device != dev):
                row_flags = None
            with torch.cuda.device(dev):
                L.check(lib.mrgcn_spmm_transposed_live_flagged_f32(
                    plan.handle, dY.data_ptr(), dY.stride(0), F, dM.data_ptr(), ld, scratch.data_ptr(),
                    live.data_ptr(), gauge.dev.data_ptr(), write_dead,
                    row_flags.data_ptr() if row_flags is not None else 0,
                    node_live.data_ptr() if node_live is not None else 0, s), "mrgcn_spmm_transposed_live_flagged_f32")
            gauge.publish()
            bump("backward.marking")
        else:
            plan.spmm(L.VIEW_TRANSPOSED, dY, F=F, out=dM)
            bump("backward.general")
        d_wI = d_comp = dX = dW = None
        # The consumers of dM are independent of each other and bound by different resources
        # (dV: HBM writes, dcomp: vector-memory issue, dW/dX: matrix cores + gathers), so the
        # input-term and feature-term backward run on two HIP streams (MRGCN_OVERLAP=0: one).
        overlap = has_I and has_X and _OVERLAP
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if overlap else main
        if overlap:
            side.wait_stream(main)  # dM is ready on `main`
        with torch.cuda.device(dev):
            if has_I and has_comp:
                N_, Bn, _ = weight_I.shape
                wI = weight_I.contiguous()
                d_comp = torch.empty_like(comp_I)
                param = getattr(ctx.owner, "weight_I", None)
                rows = None
                if (live is not None and param is not None and weight_I.is_contiguous()
                        and param.shape == weight_I.shape and _row_sparse_for(param)):
                    rows = getattr(param, "_mrgcn_rows", None)
                    if rows is not None and rows["fresh"]:
                        raise L.MrgcnError("row-sparse weight_I gradient: the layer ran twice in one train_step "
                                           "(use train_step(..., row_sparse=False))")
                    # when the shape allows it no gradient tensor exists at all: the backward keeps flags, dcomp
                    # and ||dV||^2, and ClipAdam rebuilds each live block from dM inside the Adam pass
                    # (mrgcn_adam_step_rows_fused_f32); otherwise the blocks of the live nodes go to rows["g"]
                    fused = bool(lib.mrgcn_adam_rows_fused_supported(plan.handle, Bn, F))
                    if rows is None or rows["shape"] != tuple(weight_I.shape) or rows["ever"].device != dev:
                        rows = dict(g=None, shape=tuple(weight_I.shape), cur=None, cur_owned=False,
                                    ever=torch.zeros(N_, dtype=torch.uint8, device=dev), sumsq=None, fresh=False,
                                    seeded_for=None, fused=None)
                        param._mrgcn_rows = rows
                    if not rows.get("cur_owned"):
                        # (`cur` may be a gradient support's own node flags — _support_weight_I_grads — which this path
                        # must not overwrite: the flags written below are this entry's own)
                        rows["cur"], rows["cur_owned"] = torch.zeros(N_, dtype=torch.uint8, device=dev), True
                    if not fused and rows["g"] is None:
                        rows["g"] = torch.empty_like(wI)
                if rows is not None:
                    sq = torch.zeros((), dtype=torch.float64, device=dev)
                    L.check(lib.mrgcn_basis_mix_bwd_f32(
                        plan.handle, dM.data_ptr(), ld, live.data_ptr(), wI.data_ptr(), comp_I.data_ptr(), Bn, F,
                        0 if fused else rows["g"].data_ptr(), rows["cur"].data_ptr(), d_comp.data_ptr(),
                        sq.data_ptr(), s), "mrgcn_basis_mix_bwd_f32")
                    rows["sumsq"], rows["fresh"] = sq, True
                    bump("weight_I.fused_rows" if fused else "weight_I.rows")
                    # what the fused update reads: dM and the column flags of this backward, and the coefficients
                    # as they were (the optimizer may update weight_I_comp before weight_I)
                    rows["fused"] = dict(plan=plan, dM=dM, ld=ld, live=live, comp=comp_I.detach().clone(), B=Bn,
                                         F=F) if fused else None
                    d_wI = None  # travels in param._mrgcn_rows
                else:
                    bump("weight_I.dense")
                    d_wI = torch.empty_like(wI)
                    L.check(lib.mrgcn_basis_mix_bwd_f32(
                        plan.handle, dM.data_ptr(), ld, live.data_ptr() if live is not None else 0, wI.data_ptr(),
                        comp_I.data_ptr(), Bn, F, d_wI.data_ptr(), 0, d_comp.data_ptr(), 0, s),
                        "mrgcn_basis_mix_bwd_f32")
            elif has_I:
                # dense (R*N) x F gradient in one pass: the touched rows from dM, zeros everywhere else
                d_wI = torch.empty(weight_I.shape, dtype=torch.float32, device=dev)
                if torch.cuda.is_current_stream_capturing() and "_ulcol_sorted" not in plan.__dict__:
                    raise L.MrgcnError("the sorted column list of this plan is built on first use: run one backward "
                                       "of the layer before capturing it")
                srows, sperm = plan.ulcol_sorted()
                L.check(lib.mrgcn_scatter_rows_zero_fill_f32(srows.data_ptr(), sperm.data_ptr(), srows.numel(),
                                                             dM.data_ptr(), ld, F, d_wI.data_ptr(), d_wI.shape[0], s),
                        "mrgcn_scatter_rows_zero_fill_f32")
            if has_X:
                need_dX = ctx.needs_input_grad[4]
                need_dW = ctx.needs_input_grad[5]
                K = X.shape[1]
                dx_flags = None
                with torch.cuda.stream(side):
                    if need_dX:
                        dX = torch.empty((X.shape[0], K), dtype=torch.float32, device=dev)
                    if need_dW:
                        dW = torch.empty_like(W_F)
                    if need_dX or need_dW:
                        ws = None
                        nws = int(lib.mrgcn_rel_transform_bwd_workspace(plan.handle, K, F, int(need_dX),
                                                                        int(need_dW)))
                        if nws > 0:
                            ws = torch.empty((nws,), dtype=torch.float32, device=dev)
                        # the input gradient leaves finished: masked by the ReLU that produced X (when X is such an
                        # output) and with a flag per row — the layer below then neither masks nor scans it
                        finish = need_dX and bool(lib.mrgcn_rel_transform_bwd_masked_supported(
                            plan.handle, K, F, ws.numel() if ws is not None else 0))
                        dx_flags = torch.empty((X.shape[0],), dtype=torch.uint8, device=dev) if finish else None
                        L.check(lib.mrgcn_rel_transform_bwd_masked_f32(
                            plan.handle, dM.data_ptr(), ld, live.data_ptr() if live is not None else 0,
                            X.data_ptr(), X.stride(0), K, W_F.data_ptr(), F,
                            dX.data_ptr() if need_dX else 0, K, dW.data_ptr() if need_dW else 0,
                            ws.data_ptr() if ws is not None else 0, ws.numel() if ws is not None else 0,
                            int(finish and ctx.x_is_relu_out), dx_flags.data_ptr() if finish else 0,
                            node_live.data_ptr() if (finish and node_live is not None and live is not None) else 0,
                            side.cuda_stream), "mrgcn_rel_transform_bwd_masked_f32")
                        if finish:
                            _set_grad_meta(dX, dx_flags, finish and ctx.x_is_relu_out)
                if overlap:
                    main.wait_stream(side)
                    for t in (dX, dW, ws, dM, live, dx_flags, node_live):  # allocated / used on `side`: keep the allocator honest
                        if t is not None:
                            t.record_stream(side if (t is dM or t is live or t is node_live) else main)
        return None, None, d_wI, d_comp, dX, dW, dbias, None, None, None


_DISCOVER = os.environ.get("MRGCN_DISCOVER_ROWS", "1") != "0"


def _discovered_rows(owner, plan, dY, F, dev):
    """Row flags for a dense output gradient that carries no note about its live rows (it comes from torch's own
    autograd: the reference's loop, `criterion(model(X, A)[idx], y)`), as a `structural` note: the UNION of the rows
    that held anything whenever this layer looked.  The label set of a training run does not change, so the union stops
    growing after an epoch or two (a labelled row whose gradient underflows to exact zeros in some epoch stays in it: a
    structural set may contain zero rows) and the gradient support built for it is reused from then on.  One pass
    over dY and a 1-byte read-back per call — hence not inside a stream capture, and only for the layer whose
    gradient arrives without a note: the layers below get theirs from the layer above.  None: not applicable."""
    if (owner is None or torch.cuda.is_current_stream_capturing() or dY.shape[0] != plan.num_rows
            or getattr(plan, "lean", False)):   # (a lean plan is a one-step mini-batch slice: its row set never comes back)
        return None
    if dY.dim() != 2 or dY.stride(1) != 1 or dY.dtype != torch.float32:
        return None
    if owner.__dict__.get("_mrgcn_found_rows_dense"):
        return None   # (most rows carry gradient: the general transposed product is the right backward, see below)
    flags = torch.empty((dY.shape[0],), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(L.load().mrgcn_rows_nonzero_f32(dY.data_ptr(), dY.stride(0), F, dY.shape[0], flags.data_ptr(),
                                                _stream(dev)), "mrgcn_rows_nonzero_f32")
    ent = owner.__dict__.get("_mrgcn_found_rows")
    grown = False
    if ent is None or ent.shape != flags.shape or ent.device != flags.device:
        ent, grown = flags, True
    elif bool((flags > ent).any()):   # a row outside the set so far: the set grows (a new tensor: a new support)
        ent, grown = torch.maximum(ent, flags), True
    if grown and int(ent.sum(dtype=torch.int64)) > _LiveGauge._DENSE * plan.num_rows:
        # a dense gradient (every node labelled, a loss over all rows): a support over nearly the whole graph would be a
        # second copy of the transposed arrays for nothing — from now on this layer's plain gradients take the general
        # transposed product without being looked at
        owner.__dict__["_mrgcn_found_rows_dense"] = True
        owner.__dict__.pop("_mrgcn_found_rows", None)
        return None
    owner.__dict__["_mrgcn_found_rows"] = ent
    # (rows outside the set are exact zeros in THIS gradient: it was just looked at)
    return {"version": dY._version, "row_live": ent, "relu_applied": False, "structural": True, "sparse_rows": False}


def _support_of(plan, meta, F, dev):
    """The gradient support for an output gradient whose live rows are known structurally, or None."""
    if not meta or not meta.get("structural") or F > 16:
        return None
    rf = meta["row_live"]
    if rf is None or rf.numel() != plan.num_rows or rf.device != dev or rf.dtype != torch.uint8:
        return None
    return plan.support_for(rf)


def _support_weight_I_grads(owner, sup, plan, dM, ld, weight_I, comp_I, F, s):
    """(d weight_I, d weight_I_comp) of the input term from dM ([L, ld] by the support's live numbers).  With a
    row-sparse consumer on the parameter (mrgcn_amd.optim / ClipAdam) d weight_I is None: the entry left on the
    parameter (`_mrgcn_rows`) holds what the optimizer's row update reads.  (Call under torch.cuda.device.)"""
    lib = L.load()
    dev = plan.device
    d_wI = None
    N_, Bn, _ = weight_I.shape
    wI = weight_I.contiguous()
    d_comp = torch.empty_like(comp_I)
    param = getattr(owner, "weight_I", None)
    rows = None
    if (param is not None and weight_I.is_contiguous() and param.shape == weight_I.shape
            and _row_sparse_for(param)):
        rows = getattr(param, "_mrgcn_rows", None)
        if rows is not None and rows["fresh"]:
            raise L.MrgcnError("row-sparse weight_I gradient: the layer ran twice in one train_step "
                               "(use train_step(..., row_sparse=False))")
        fused = bool(lib.mrgcn_adam_rows_fused_supported(plan.handle, Bn, F))
        if rows is None or rows["shape"] != tuple(weight_I.shape) or rows["ever"].device != dev:
            rows = dict(g=None, shape=tuple(weight_I.shape), cur=None,
                        ever=torch.zeros(N_, dtype=torch.uint8, device=dev), sumsq=None, fresh=False,
                        seeded_for=None, fused=None)
            param._mrgcn_rows = rows
        if not fused and rows["g"] is None:
            rows["g"] = torch.empty_like(wI)
    if rows is not None:
        # the nodes of the support: the same set every epoch (the support's own tensor, read-only here: the marking
        # path allocates its own when it takes over — `cur_owned`)
        rows["cur"], rows["cur_owned"] = sup.node_flags(), False
        if fused:
            # norm-only: dcomp and ||dV||^2 are written whole (no zero fills), nothing is accumulated
            sq = torch.empty((), dtype=torch.float64, device=dev)
            ws = sup.workspace(("mix", Bn), int(lib.mrgcn_support_mix_bwd_workspace(sup.handle, Bn)))
            L.check(lib.mrgcn_support_mix_bwd_f32(
                sup.handle, dM.data_ptr(), ld, wI.data_ptr(), comp_I.data_ptr(), Bn, F, 0, 0,
                d_comp.data_ptr(), sq.data_ptr(), ws.data_ptr(), ws.numel(), s), "mrgcn_support_mix_bwd_f32")
        else:
            sq = torch.zeros((), dtype=torch.float64, device=dev)
            L.check(lib.mrgcn_support_mix_bwd_f32(
                sup.handle, dM.data_ptr(), ld, wI.data_ptr(), comp_I.data_ptr(), Bn, F, rows["g"].data_ptr(), 0,
                d_comp.data_ptr(), sq.data_ptr(), 0, 0, s), "mrgcn_support_mix_bwd_f32")
        rows["sumsq"], rows["fresh"] = sq, True
        bump("weight_I.fused_rows" if fused else "weight_I.rows")
        # what the fused update reads: dM of this backward and the coefficients as they were (ClipAdam steps
        # the node table before weight_I_comp; the version is checked there)
        rows["fused"] = dict(sup=sup, plan=plan, dM=dM, ld=ld, live=None, comp=comp_I.detach(),
                             comp_version=comp_I._version, B=Bn, F=F) if fused else None
    else:
        bump("weight_I.dense")
        d_wI = torch.empty_like(wI)
        L.check(lib.mrgcn_support_mix_bwd_f32(
            sup.handle, dM.data_ptr(), ld, wI.data_ptr(), comp_I.data_ptr(), Bn, F, d_wI.data_ptr(), 1,
            d_comp.data_ptr(), 0, 0, 0, s), "mrgcn_support_mix_bwd_f32")
    return d_wI, d_comp


def _support_backward_workspace(ctx, sup):
    """floats of workspace the transform's backward on `sup` needs (0: no feature term), or None when a shape is outside
    what the support calls take (the caller then runs another path)."""
    lib = L.load()
    weight_I, comp_I, X, W_F, Y = ctx.saved_tensors
    has_I, has_comp, has_X, has_bias = ctx.has
    F = ctx.F
    if has_I and not has_comp:
        return None  # (the literal (R*N) x F gradient is scattered from plain compact order)
    if has_I and (comp_I.shape[1] > 64 or F > 16):
        return None  # (outside the support's node-major mix backward: mrgcn_support_mix_bwd_f32)
    need_dX = has_X and ctx.needs_input_grad[4]
    need_dW = has_X and ctx.needs_input_grad[5]
    nws = 0
    if need_dX or need_dW:
        nws = int(lib.mrgcn_support_rel_transform_bwd_workspace(sup.handle, X.shape[1], F, int(need_dX), int(need_dW)))
        if nws < 0:
            return None
    return nws


def _backward_on_support(ctx, sup, dY, dbias):
    """_RgcnLayer.backward on a gradient support: dM and every per-column product are [L, ld] arrays by live number;
    no marking, no flags, no zero fills (csrc/support.hip).  None when a shape is outside what the support calls
    take (the caller then runs the per-epoch marking path)."""
    lib = L.load()
    nws = _support_backward_workspace(ctx, sup)
    if nws is None:
        return None
    plan, F = ctx.plan, ctx.F
    dev = plan.device
    s = _stream(dev)
    ld = (F + 3) // 4 * 4
    dM = torch.empty((max(sup.L, 1), ld), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.mrgcn_support_spmm_t_f32(sup.handle, dY.data_ptr(), dY.stride(0), F, dM.data_ptr(), ld, s),
                "mrgcn_support_spmm_t_f32")
    return _support_backward_from_dM(ctx, sup, dM, ld, dbias, nws)


def _support_backward_from_dM(ctx, sup, dM, ld, dbias, nws):
    """The layer's parameter / input gradients from dM [L, ld] (the gradient of the live columns' operand rows, by the
    support's live numbers): mix backward, the transform's dW / dX.  `ctx`: anything with _RgcnLayer's context fields
    (plan, F, saved_tensors, has, needs_input_grad, owner, x_is_relu_out, Xb) — the node-partitioned halo engine sums
    the ranks' contributions into dM first and calls this on its own columns' support (partition_halo.py)."""
    lib = L.load()
    plan, F = ctx.plan, ctx.F
    weight_I, comp_I, X, W_F, Y = ctx.saved_tensors
    has_I, has_comp, has_X, has_bias = ctx.has
    dev = plan.device
    need_dX = has_X and ctx.needs_input_grad[4]
    need_dW = has_X and ctx.needs_input_grad[5]
    K = X.shape[1] if has_X else 0
    s = _stream(dev)
    d_wI = d_comp = dX = dW = None
    overlap = has_I and has_X and _OVERLAP
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev) if overlap else main
    if overlap:
        side.wait_stream(main)
    with torch.cuda.device(dev):
        if has_I:
            d_wI, d_comp = _support_weight_I_grads(ctx.owner, sup, plan, dM, ld, weight_I, comp_I, F, s)
        if need_dX or need_dW:
            with torch.cuda.stream(side):
                if need_dX:
                    dX = torch.empty((X.shape[0], K), dtype=torch.float32, device=dev)
                if need_dW:
                    dW = torch.empty_like(W_F)
                ws = sup.workspace(("xform", K, F), nws)
                mask = bool(need_dX and ctx.x_is_relu_out and K <= 16)
                Xb = getattr(ctx, "Xb", None)
                if Xb is not None and not mask:
                    bump("bf16.dw_xbf16")
                    L.check(lib.mrgcn_support_rel_transform_bwd_xbf16(
                        sup.handle, dM.data_ptr(), ld, Xb.data_ptr(), Xb.stride(0), K, W_F.data_ptr(), F,
                        dX.data_ptr() if need_dX else 0, K, dW.data_ptr() if need_dW else 0, ws.data_ptr(), ws.numel(),
                        side.cuda_stream), "mrgcn_support_rel_transform_bwd_xbf16")
                else:
                    L.check(lib.mrgcn_support_rel_transform_bwd_f32(
                        sup.handle, dM.data_ptr(), ld, X.data_ptr(), X.stride(0), K, W_F.data_ptr(), F,
                        dX.data_ptr() if need_dX else 0, K, dW.data_ptr() if need_dW else 0, ws.data_ptr(), ws.numel(),
                        int(mask), side.cuda_stream), "mrgcn_support_rel_transform_bwd_f32")
                if need_dX:
                    # the rows that can hold anything: the nodes of this support — the row set of the layer below
                    _set_grad_meta(dX, sup.node_flags(), mask, structural=True)
            if overlap:
                main.wait_stream(side)
                for t in (dX, dW):
                    if t is not None:
                        t.record_stream(main)
                dM.record_stream(side)
    return None, None, d_wI, d_comp, dX, dW, dbias, None, None, None


_RgcnLayer._backward_on_support = staticmethod(_backward_on_support)


class _MaskedLayer(torch.autograd.Function):
    """One `GraphConvolution` of a mini-batch (graph.py:62-102 with A_idx) as a masked pass over the FULL graph's plan
    (csrc/masked.hip): `sup` is the forward support of the layer's sample on that plan — Y has one row per flagged row
    (rising row id), X one row per live node (the layer's neighbours, rising node id).  The input term multiplies the
    stored values, the feature term the all-ones slice (batch.py:258-270)."""

    @staticmethod
    def forward(ctx, sup, F: int, weight_I, comp_I, X, W_F, bias, relu: bool, owner=None):
        lib = L.load()
        plan = sup.plan
        dev = sup.device
        s = _stream(dev)
        ld = (F + 3) // 4 * 4
        NR, Lc = sup.NR, max(sup.L, 1)
        has_I, has_X = weight_I is not None, X is not None
        Y = torch.empty((NR, F), dtype=torch.float32, device=dev)
        Xc = Wc = wI = cI = None
        with torch.cuda.device(dev):
            if has_X:
                Xc = X if (X.dim() == 2 and X.stride(1) == 1) else X.contiguous()
                Wc = W_F.contiguous()
                # X: one row per neighbour (the reference's mksubset form) or the WHOLE feature matrix, one row per node
                # — the transform then picks the neighbours' rows itself (no X[batch.neighbours[-1]] copy in front)
                x_by_node = Xc.shape[0] == plan.num_nodes and sup.NL != plan.num_nodes
                if Xc.shape[0] != sup.NL and not x_by_node:
                    raise L.MrgcnError(f"masked layer: X has {Xc.shape[0]} rows, the sample has {sup.NL} neighbours "
                                       f"(or hand over all {plan.num_nodes} rows)")
                T = torch.empty((Lc, ld), dtype=torch.float32, device=dev)
                L.check(lib.mrgcn_support_rel_transform_fwd_f32(sup.handle, Xc.data_ptr(), Xc.stride(0), int(x_by_node),
                                                                Xc.shape[1], Wc.data_ptr(), F, T.data_ptr(), ld, s),
                        "mrgcn_support_rel_transform_fwd_f32")
            if has_I:
                wI = weight_I.contiguous()
                M = torch.empty((Lc, ld), dtype=torch.float32, device=dev)
                if comp_I is not None:
                    cI = comp_I.contiguous()
                    L.check(lib.mrgcn_support_mix_fwd_f32(sup.handle, wI.data_ptr(), cI.data_ptr(), cI.shape[1], F,
                                                          M.data_ptr(), ld, s), "mrgcn_support_mix_fwd_f32")
                else:  # no bases: the literal (R*N) x F table, a row per live column
                    L.check(lib.mrgcn_support_literal_rows_f32(sup.handle, 0, wI.data_ptr(), F, M.data_ptr(), ld, s),
                            "mrgcn_support_literal_rows_f32")
            b = bias.data_ptr() if bias is not None else 0
            # the feature term multiplies the all-ones slice (the reference's mini-batch arithmetic: `sliceSparseCOO`
            # drops the values, batch.py:258-270) — or, for a batch built with `full_batch_values`, the stored values
            # like the input term: the FULL-batch arithmetic of graph.py:93-95 restricted to the batch's receptive field
            fv = int(bool(getattr(sup, "feature_values", False)))
            if has_I and has_X:
                YI = torch.empty((NR, F), dtype=torch.float32, device=dev)
                L.check(lib.mrgcn_support_spmm_fwd_f32(sup.handle, 1, M.data_ptr(), ld, F, YI.data_ptr(), F, 0, 0, s),
                        "mrgcn_support_spmm_fwd_f32")
                L.check(lib.mrgcn_support_spmm_fwd_f32(sup.handle, fv, T.data_ptr(), ld, F, Y.data_ptr(), F, b, 0, s),
                        "mrgcn_support_spmm_fwd_f32")
                # (graph.py:95-101: AIW + AFW, the bias inside AFW)
                Y = torch.add(YI, Y, out=Y)
                if relu:
                    Y.relu_()
            else:
                L.check(lib.mrgcn_support_spmm_fwd_f32(sup.handle, int(has_I or fv), (M if has_I else T).data_ptr(), ld, F,
                                                       Y.data_ptr(), F, b, int(relu), s), "mrgcn_support_spmm_fwd_f32")
        ctx.sup, ctx.plan, ctx.F, ctx.ld, ctx.relu, ctx.owner = sup, plan, F, ld, relu, owner
        ctx.x_by_node = bool(has_X and x_by_node)
        ctx.x_is_relu_out = has_X and bool(getattr(X, "_mrgcn_relu_out", False))
        ctx.has = (has_I, has_X, bias is not None)
        ctx.save_for_backward(wI, cI, Xc, Wc, Y if relu else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = L.load()
        sup, plan, F, ld = ctx.sup, ctx.plan, ctx.F, ctx.ld
        weight_I, comp_I, X, W_F, Y = ctx.saved_tensors
        has_I, has_X, has_bias = ctx.has
        dev = sup.device
        s = _stream(dev)
        dY = dY.contiguous()
        meta = _grad_meta(dY)
        if ctx.relu and not (meta and meta["relu_applied"]):
            dY = relu_bwd(dY, Y)
        dbias = _bias_grad(dY) if has_bias else None
        d_wI = d_comp = dX = dW = None
        Lc = max(sup.L, 1)
        with torch.cuda.device(dev):
            if has_I:
                dM = torch.empty((Lc, ld), dtype=torch.float32, device=dev)
                L.check(lib.mrgcn_support_spmm_t_compact_f32(sup.handle, 1, dY.data_ptr(), dY.stride(0), F,
                                                             dM.data_ptr(), ld, s), "mrgcn_support_spmm_t_compact_f32")
                if comp_I is not None:
                    d_wI, d_comp = _support_weight_I_grads(ctx.owner, sup, plan, dM, ld, weight_I, comp_I, F, s)
                else:
                    d_wI = torch.empty_like(weight_I)
                    L.check(lib.mrgcn_support_literal_rows_f32(sup.handle, 1, d_wI.data_ptr(), F, dM.data_ptr(), ld, s),
                            "mrgcn_support_literal_rows_f32")
            need_dX = has_X and ctx.needs_input_grad[4]
            need_dW = has_X and ctx.needs_input_grad[5]
            if need_dX or need_dW:
                K = X.shape[1]
                dT = torch.empty((Lc, ld), dtype=torch.float32, device=dev)
                L.check(lib.mrgcn_support_spmm_t_compact_f32(sup.handle, int(bool(getattr(sup, "feature_values", False))),
                                                             dY.data_ptr(), dY.stride(0), F, dT.data_ptr(), ld, s),
                        "mrgcn_support_spmm_t_compact_f32")
                nws = int(lib.mrgcn_support_rel_transform_bwd_workspace(sup.handle, K, F, int(need_dX), int(need_dW)))
                ws = sup.workspace(("xform", K, F), nws)
                if need_dX:
                    dX = torch.empty((sup.NL, K), dtype=torch.float32, device=dev)
                if need_dW:
                    dW = torch.empty_like(W_F)
                mask = bool(need_dX and ctx.x_is_relu_out and K <= 16 and not ctx.x_by_node)
                L.check(lib.mrgcn_support_rel_transform_bwd_compact_f32(
                    sup.handle, dT.data_ptr(), ld, X.data_ptr(), X.stride(0), int(ctx.x_by_node), K, W_F.data_ptr(), F,
                    dX.data_ptr() if need_dX else 0, K, dW.data_ptr() if need_dW else 0, ws.data_ptr(), ws.numel(),
                    int(mask), s), "mrgcn_support_rel_transform_bwd_compact_f32")
                if need_dX and mask:
                    _set_grad_meta(dX, None, True)
                if need_dX and ctx.x_by_node:  # the whole matrix wants its gradient: zeros outside the neighbours
                    full = torch.zeros((X.shape[0], K), dtype=torch.float32, device=dev)
                    full.index_copy_(0, sup.view(L.SUP_LNODE).long(), dX)
                    dX = full
        return None, None, d_wI, d_comp, dX, dW, dbias, None, None


def masked_layer_supported(sup, layer, K: int, need_dX: bool = False) -> bool:
    """Can `masked_layer` run this layer?  (narrow f32 rows, the matrix-core transforms' shapes.)"""
    if layer.outdim > 16 or getattr(layer, "operand_dtype", "f32") != "f32":
        return False
    if not (layer.input_layer and layer.featureless):
        return bool(L.load().mrgcn_support_rel_transform_supported(sup.handle, int(K), int(layer.outdim), int(need_dX)))
    return True


def masked_layer(sup, layer, X, relu: bool = False) -> torch.Tensor:
    """One `GraphConvolution` on a mini-batch sample given as a forward support of the full graph's plan
    (data.batch.A_BatchMasked): graph.py:62-102 with A_idx, both terms."""
    F, B = layer.outdim, layer.num_bases
    weight_I = comp_I = Xin = W_F = None
    if layer.input_layer:
        weight_I, comp_I = layer.weight_I, (layer.weight_I_comp if B > 0 else None)
    if not (layer.input_layer and layer.featureless):
        if X is None:
            raise L.MrgcnError("masked_layer: the feature term needs X")
        Xin, W_F = X, layer.weight_F
        if B > 0:
            W_F = _BasisContract.apply(layer.weight_F_comp, W_F)
    bias = layer.b if layer.bias else None
    Y = _MaskedLayer.apply(sup, F, weight_I, comp_I, Xin, W_F, bias, relu, layer)
    if relu:
        Y._mrgcn_relu_out = True
    return Y


class _BasisContract(torch.autograd.Function):
    """W_F[r] = sum_b comp[r, b] V_F[b] (graph.py:83-85) and its backward on this package's kernels
    (mrgcn_basis_contract_f32 / _bwd_f32): comp (R, B), V (B, in, out) -> (R, in, out)."""

    @staticmethod
    def forward(ctx, comp, V):
        comp_c, V_c = comp.contiguous(), V.contiguous()
        R, B = comp_c.shape
        X = V_c.numel() // B
        W = torch.empty((R,) + tuple(V_c.shape[1:]), dtype=torch.float32, device=V_c.device)
        with torch.cuda.device(V_c.device):
            L.check(L.load().mrgcn_basis_contract_f32(comp_c.data_ptr(), V_c.data_ptr(), R, B, X, W.data_ptr(),
                                                      _stream(V_c.device)), "mrgcn_basis_contract_f32")
        ctx.save_for_backward(comp_c, V_c)
        return W

    @staticmethod
    def backward(ctx, dW):
        comp, V = ctx.saved_tensors
        R, B = comp.shape
        X = V.numel() // B
        dW = dW.contiguous()
        dcomp = torch.empty_like(comp) if ctx.needs_input_grad[0] else None
        dV = torch.empty_like(V) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(V.device):
            L.check(L.load().mrgcn_basis_contract_bwd_f32(
                comp.data_ptr(), V.data_ptr(), dW.data_ptr(), R, B, X, dcomp.data_ptr() if dcomp is not None else 0,
                dV.data_ptr() if dV is not None else 0, _stream(V.device)), "mrgcn_basis_contract_bwd_f32")
        return dcomp, dV


def rgcn_layer(plan: GraphPlan, layer, X, relu: bool = False, input_term: bool = True,
               feature_term: bool = True, use_bias: bool = True) -> torch.Tensor:
    """Fused forward of one `GraphConvolution` (graph.py:62-102).  `input_term` / `feature_term`
    select the two summands (mini-batch mode runs them on different column spaces)."""
    F = layer.outdim
    B = layer.num_bases
    weight_I = comp_I = W_F = None
    if layer.input_layer and input_term:
        weight_I = layer.weight_I
        comp_I = layer.weight_I_comp if B > 0 else None
    Xin = None
    if feature_term and not (layer.input_layer and layer.featureless):
        if X is None:
            raise L.MrgcnError("rgcn_layer: the feature term needs X")
        Xin = X
        W_F = layer.weight_F
        if B > 0:  # graph.py:83-85: the (R x B) . (B x in*out) contraction
            W_F = _BasisContract.apply(layer.weight_F_comp, W_F)
    if weight_I is None and Xin is None:
        raise L.MrgcnError("rgcn_layer: neither the input term nor the feature term is selected "
                           "(a featureless layer only has the input term)")
    bf16 = getattr(layer, "operand_dtype", "f32") == "bf16"
    bias = layer.b if (layer.bias and use_bias) else None
    if weight_I is not None and comp_I is None and Xin is None and not bf16:
        # featureless layer without bases: weight_I already *is* the literal operand
        Y = spmm_literal(plan, weight_I, bias=bias, relu=relu, owner=layer)
        if relu:
            Y._mrgcn_relu_out = True   # (the layer above masks its input gradient with this output's sign)
        return Y
    Y = _RgcnLayer.apply(plan, F, weight_I, comp_I, Xin, W_F, bias, relu, bf16, layer)
    if relu:
        Y._mrgcn_relu_out = True  # (a Python attribute of this tensor object: a copy or a view does not carry it)
    if F <= 16 and _SUPPORT and _LIVE_COLS and not plan.lean:
        # this layer's backward reads the flagged rows of its output gradient only: a loss that knows the rows it
        # touches (train.categorical_crossentropy) need not zero-fill the rest (AM shape: 73 MB per epoch)
        Y._mrgcn_sparse_grad_ok = True
    return Y
