"""autograd wiring of the HIP kernels.  Every op here calls the C ABI; none has a PyTorch
or CPU fallback."""
from __future__ import annotations

import os
import weakref

import torch

from . import _lib as L
from .plan import GraphPlan


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# ||grad||^2 values a producing kernel already accumulated (device doubles), keyed by the
# gradient's storage pointer; `ClipAdam` consumes them instead of re-reading the gradient.
# Safe by construction: autograd either adopts the returned tensor's storage as `p.grad`
# (pointer matches) or hands the optimizer a different tensor (sum of several contributions,
# a clone) whose pointer does not match, in which case the norm is recomputed.  The registry is
# emptied by every optimizer step.
_GRAD_SUMSQ: dict = {}


def register_grad_sumsq(grad: torch.Tensor, sumsq: torch.Tensor):
    _GRAD_SUMSQ[grad.data_ptr()] = (grad.numel(), sumsq)


def pop_grad_sumsq(grad: torch.Tensor):
    """The precomputed sum of squares of the gradient living at this storage, or None."""
    ent = _GRAD_SUMSQ.pop(grad.data_ptr(), None)
    if ent is None or ent[0] != grad.numel():
        return None
    return ent[1]


def clear_grad_sumsq():
    _GRAD_SUMSQ.clear()
    _DEFERRED.clear()


# Deferred update of weight_I (opt-in): with it on, the backward of a bases layer does not store
# dV (weight_I.grad stays None); it leaves ||dV||^2 plus what is needed to recompute dV here, and
# `ClipAdam.step` applies Adam to weight_I inside the kernel that recomputes it
# (mrgcn_basis_mix_bwd_adam_f32).  Same arithmetic, 2 x 2.67 GB less HBM traffic per AM epoch —
# but measured NOT faster on MI355X (AM shape: 14.96 ms vs 14.66 ms with the stored gradient):
# the recomputing kernel streams p/m/v as 4-byte lanes over many basis slabs at ~3.8 TB/s where
# the plain float4 Adam kernel reaches ~4.7 TB/s, which eats the saved traffic.  Kept as an
# option (it halves the peak memory of the step: no 2.67 GB gradient tensor).
# Only valid when nothing else contributes to weight_I's gradient (no L1/L2 term, one use of the
# layer per step) — `train_step` switches it off when a regulariser is active.
_DEFER = False
_DEFERRED: dict = {}


def defer_input_grad(on: bool) -> bool:
    """Switches the deferred weight_I update on/off; returns the previous setting."""
    global _DEFER
    prev, _DEFER = _DEFER, bool(on)
    return prev


def pop_deferred(param: torch.Tensor):
    ent = _DEFERRED.pop(param.data_ptr(), None)
    if ent is None or ent["numel"] != param.numel():
        return None
    return ent


# measured on the AM shape: 14.6 ms with the overlap vs 14.3 ms without (every one of these
# kernels already saturates the memory system on its own), so it is opt-in
_OVERLAP = os.environ.get("MRGCN_OVERLAP", "0") != "0"
# Skip the compact columns without gradient in the transform backward (exact: they add zeros).
_LIVE_COLS = os.environ.get("MRGCN_LIVE_COLS", "1") != "0"


# tests: start dM as NaNs, so that any read of a row the producer left unwritten shows
_POISON_DEAD = False

# Chunk-sparse gradient of weight_I (train_step turns it on for ClipAdam without weight decay): nodes
# without a live compact column have a zero gradient row in every basis, and with a fixed label set
# never any other — their Adam moments stay zero and their parameters never move.  The backward then
# leaves the 4 KB chunks of dV without any live node unwritten and ClipAdam runs
# mrgcn_adam_step_chunked_f32, which never touches chunks that never had gradient.  How much that
# saves depends on how the nodes are numbered: see mrgcn_amd.data.reorder.
_SPARSE_WGRAD = False
_WCHUNKS: dict = {}
_WCHUNK_DENSE = 0.75  # fraction of chunks that ever had gradient above which the masks are dropped


def sparse_weight_grad(enabled: bool) -> bool:
    """Returns the previous setting."""
    global _SPARSE_WGRAD
    prev, _SPARSE_WGRAD = _SPARSE_WGRAD, bool(enabled)
    return prev


# Node-major optimizer space (preferred over the chunk masks when the shape allows): the gradient of
# weight_I is written as [N][B][F] for the nodes with gradient only and ClipAdam keeps its moments in
# the same layout (mrgcn_adam_step_nodemajor_f32) — nodes that never had gradient are skipped whatever
# the node numbering.  weight_I.grad stays None; the entry below carries the gradient to the optimizer.
_NODEMAJOR: dict = {}
_NODE_MAJOR = os.environ.get("MRGCN_NODE_MAJOR", "1") != "0"


def pop_nodemajor(param: torch.Tensor):
    ent = _NODEMAJOR.get(param.data_ptr())
    if ent is None or not ent["fresh"] or ent["numel"] != param.numel():
        return None
    ent["fresh"] = False
    return ent


def pop_weight_chunks(param: torch.Tensor):
    """(cur, ever, slab_elems, B) when this step's gradient of `param` was produced chunk-sparse."""
    ent = _WCHUNKS.get(param.data_ptr())
    if ent is None or not ent["fresh"] or ent["numel"] != param.numel():
        return None
    ent["fresh"] = False
    return ent


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

    def release(self):
        if self.slot is not None:
            _LiveGauge._free.append(self.slot)
            self.slot = None

    def sparse(self) -> bool:
        self.calls += 1
        last = int(self.host[0])  # pinned host memory: no synchronisation
        return last < 0 or last <= self._DENSE * self.num_rows or self.calls % self._RETRY == 0

    def publish(self):
        if self.slot is not None:
            self.host.copy_(self.dev, non_blocking=True)


_GAUGES = {}


def _live_gauge(plan, F, relu, dev):
    key = (id(plan), F, bool(relu))
    g = _GAUGES.get(key)
    if g is None or g.plan_ref() is not plan:
        if g is not None:
            g.release()
        if len(_GAUGES) >= 256:  # mini-batch training builds plans by the thousand: drop the dead ones
            for k in [k for k, v in _GAUGES.items() if v.plan_ref() is None]:
                _GAUGES.pop(k).release()
        g = _LiveGauge(plan.num_rows, dev)
        g.plan_ref = weakref.ref(plan)
        _GAUGES[key] = g
    return g
_SIDE_STREAMS: dict = {}


def _side_stream(device) -> torch.cuda.Stream:
    st = _SIDE_STREAMS.get(device)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _SIDE_STREAMS[device] = st
    return st


def _ld_for(F: int) -> int:
    """Leading dimension of the compact operand: rows padded to a multiple of 4 floats (16-byte
    float4 gathers).  Measured on the AM shape: 12 vs 16 floats per 10-feature row gather within
    2 % of each other, while the 12-float rows make the producers' scattered pass ~8 % cheaper.
    MRGCN_LDM_ALIGN overrides (floats, multiple of 4)."""
    a = int(os.environ.get("MRGCN_LDM_ALIGN", "4"))
    return (F + a - 1) // a * a


def _ld_for_bf16(F: int) -> int:
    """bf16 operand rows: padded to a multiple of MRGCN_LDM_BF16_ALIGN elements.  4 (8-byte gathers,
    rows no wider than the fp32 operand's, so the producers run no extra lanes) measured 0.15 ms per
    AM epoch better than 8 (16-byte gathers)."""
    a = int(os.environ.get("MRGCN_LDM_BF16_ALIGN", "4"))
    return (F + a - 1) // a * a


class _SpmmLiteral(torch.autograd.Function):
    """Y = A . D with D the dense (R*N) x F operand in the reference's row order
    (mrgcn/layers/graph.py:75, :95).  Backward = A^T dY scattered into a dense
    (R*N) x F gradient, as SparseAddmmBackward produces for the reference."""

    @staticmethod
    def forward(ctx, plan: GraphPlan, D: torch.Tensor, bias, relu: bool):
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
        if ctx.relu:
            dY = relu_bwd(dY, Y)
        dbias = dY.sum(0) if ctx.has_bias else None
        dD = None
        if ctx.needs_input_grad[1]:
            F = dY.shape[1]
            dD = torch.zeros((ctx.d_rows, F), dtype=torch.float32, device=dY.device)
            ulcol_ptr, _ = plan.array_ptr(L.ARR_ULCOL)
            plan.spmm(L.VIEW_TRANSPOSED, dY, out=dD, out_index=ulcol_ptr)
        return None, dD, dbias, None


def spmm_literal(plan: GraphPlan, D: torch.Tensor, bias=None, relu: bool = False) -> torch.Tensor:
    return _SpmmLiteral.apply(plan, D, bias, relu)


def relu_bwd(dY: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(dY)
    with torch.cuda.device(dY.device):
        L.check(L.load().mrgcn_relu_bwd_f32(dY.data_ptr(), Y.data_ptr(), dY.numel(), out.data_ptr(),
                                            _stream(dY.device)), "mrgcn_relu_bwd_f32")
    return out


class _RgcnLayer(torch.autograd.Function):
    """Y = relu?( A' . M + b ),  M[c] = comp_I[r_c] . V_I[:, j_c, :]  (or weight_I[r_c*N + j_c])
                                       + X[j_c] . W_F[r_c]
    i.e. graph.py:62-102 without the (R*N) x out intermediates."""

    @staticmethod
    def forward(ctx, plan: GraphPlan, F: int, weight_I, comp_I, X, W_F, bias, relu: bool, bf16: bool = False):
        lib = L.load()
        dev = plan.device
        # bf16: only the compact operand M is stored in bf16 (one rounding at its store); inputs,
        # every accumulation, Y and the whole backward stay fp32
        ld = _ld_for_bf16(F) if bf16 else _ld_for(F)
        M = torch.empty((plan.ncols, ld), dtype=torch.bfloat16 if bf16 else torch.float32, device=dev)
        sfx = "bf16" if bf16 else "f32"
        xform_fwd = getattr(lib, "mrgcn_rel_transform_fwd_" + sfx)
        mix_fwd = getattr(lib, "mrgcn_basis_mix_fwd_" + sfx)
        gather_rows = getattr(lib, "mrgcn_gather_rows_" + sfx)
        s = _stream(dev)
        Xc = Wc = None
        with torch.cuda.device(dev):
            addend, ldA = 0, 0
            if X is not None:
                Xc = X.contiguous()
                Wc = W_F.contiguous()
                if weight_I is not None:
                    # feature term in plain compact order (sequential writes); the input-term
                    # pass below adds it while it emits the final rows in operand order
                    ldA = (F + 3) // 4 * 4
                    M2 = torch.empty((plan.ncols, ldA), dtype=torch.float32, device=dev)
                    out, ldo, order = M2, ldA, 0
                    addend = M2.data_ptr()
                else:
                    out, ldo, order = M, ld, 1
                fwd = xform_fwd if out is M else lib.mrgcn_rel_transform_fwd_f32  # M2 stays fp32
                L.check(fwd(plan.handle, Xc.data_ptr(), Xc.stride(0), Xc.shape[1], Wc.data_ptr(), F,
                            out.data_ptr(), ldo, order, s), "mrgcn_rel_transform_fwd_" + sfx)
            if weight_I is not None:
                wI = weight_I.contiguous()
                if comp_I is not None:
                    cI = comp_I.contiguous()
                    L.check(mix_fwd(plan.handle, wI.data_ptr(), cI.data_ptr(), cI.shape[1], F, addend, ldA,
                                    M.data_ptr(), ld, s), "mrgcn_basis_mix_fwd_" + sfx)
                else:
                    L.check(gather_rows(plan.handle, wI.data_ptr(), F, addend, ldA, M.data_ptr(), ld, s),
                            "mrgcn_gather_rows_" + sfx)
        Y = plan.spmm(L.VIEW_COMPACT, M, F=F, bias=bias, relu=relu)
        ctx.plan, ctx.F, ctx.ld, ctx.relu = plan, F, ld, relu
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
        if ctx.relu:
            dY = relu_bwd(dY, Y)
        dbias = dY.sum(0) if has_bias else None
        # dM = A'^T dY over touched columns only, plain compact order (its consumers are node-major)
        ld = (F + 3) // 4 * 4
        dM = torch.empty((plan.ncols, ld), dtype=torch.float32, device=dev)
        if _POISON_DEAD:
            dM.fill_(float("nan"))
        live = None
        gauge = _live_gauge(plan, F, ctx.relu, dev) if _LIVE_COLS else None
        if gauge is not None and gauge.sparse():
            # with few labelled nodes most rows of dY are zeros: gather the others only, and keep one
            # byte per compact column: does it carry any gradient?
            live = torch.empty((plan.ncols,), dtype=torch.uint8, device=dev)
            scratch = torch.empty((int(lib.mrgcn_spmm_transposed_live_scratch(plan.handle)),),
                                  dtype=torch.uint8, device=dev)
            # rows of dM without gradient are not even written when every consumer goes by the flags
            # (the no-bases scatter and the deferred update's second pass read dM itself)
            write_dead = int(has_I and (not has_comp or (_DEFER and weight_I.is_contiguous())))
            with torch.cuda.device(dev):
                L.check(lib.mrgcn_spmm_transposed_live_f32(
                    plan.handle, dY.data_ptr(), dY.stride(0), F, dM.data_ptr(), ld, scratch.data_ptr(),
                    live.data_ptr(), gauge.dev.data_ptr(), write_dead, s), "mrgcn_spmm_transposed_live_f32")
            gauge.publish()
        else:
            plan.spmm(L.VIEW_TRANSPOSED, dY, F=F, out=dM)
        d_wI = d_comp = dX = dW = None
        # The consumers of dM are independent of each other and bound by different resources
        # (dV: HBM writes, dcomp: vector-memory issue, dW/dX: matrix cores + gathers), so the
        # input-term and feature-term backward run on two HIP streams and overlap.
        overlap = has_I and has_X and _OVERLAP
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if overlap else main
        if overlap:
            side.wait_stream(main)  # dM is ready on `main`
        with torch.cuda.device(dev):
            if has_I:
                if has_comp:
                    defer = _DEFER and weight_I.is_contiguous()
                    if defer and weight_I.data_ptr() in _DEFERRED:
                        raise L.MrgcnError("deferred weight_I update: the layer ran twice in one step")
                    chunk_cur = 0
                    nm = _NODEMAJOR.get(weight_I.data_ptr())
                    if nm is not None:
                        nm["fresh"] = False
                    Bn = comp_I.shape[1]
                    if (_SPARSE_WGRAD and _NODE_MAJOR and not defer and live is not None and weight_I.is_contiguous()
                            and lib.mrgcn_nodemajor_supported(plan.handle, Bn, F) == 1):
                        N_ = plan.num_nodes
                        if nm is None or nm["numel"] != weight_I.numel() or nm["shape"] != (N_, Bn, F):
                            nm = dict(g=torch.empty((N_, Bn, F), dtype=torch.float32, device=dev),
                                      cur=torch.zeros(N_, dtype=torch.uint8, device=dev),
                                      ever=torch.zeros(N_, dtype=torch.uint8, device=dev),
                                      numel=weight_I.numel(), shape=(N_, Bn, F), fresh=False, sumsq=None)
                            if len(_NODEMAJOR) >= 4:  # gradient buffers of models long gone: keep the newest few
                                for k in list(_NODEMAJOR)[:len(_NODEMAJOR) - 3]:
                                    if not _NODEMAJOR[k]["fresh"]:
                                        del _NODEMAJOR[k]
                            _NODEMAJOR.pop(weight_I.data_ptr(), None)
                            _NODEMAJOR[weight_I.data_ptr()] = nm
                        d_comp = torch.empty_like(comp_I)
                        sq = torch.zeros((), dtype=torch.float64, device=dev)
                        L.check(lib.mrgcn_basis_mix_bwd_nodemajor_f32(
                            plan.handle, dM.data_ptr(), ld, live.data_ptr(), weight_I.data_ptr(), comp_I.data_ptr(),
                            Bn, F, nm["g"].data_ptr(), nm["cur"].data_ptr(), d_comp.data_ptr(), sq.data_ptr(), s),
                            "mrgcn_basis_mix_bwd_nodemajor_f32")
                        nm["ever"] |= nm["cur"]
                        nm["sumsq"], nm["fresh"] = sq, True
                        d_wI = None
                        has_comp_done = True
                    else:
                        has_comp_done = False
                    if not has_comp_done:
                        d_wI = None if defer else torch.empty_like(weight_I)
                        d_comp = torch.empty_like(comp_I)
                        sq = torch.zeros((), dtype=torch.float64, device=dev)
                        stale = _WCHUNKS.get(weight_I.data_ptr())
                        if stale is not None:
                            stale["fresh"] = False  # a mask of an earlier step says nothing about this gradient
                            if not _SPARSE_WGRAD or stale["dense"]:
                                # a step on the plain path may put moments where `ever` has never looked:
                                # the next masked step rebuilds `ever` from the optimizer state
                                stale["state_synced"] = False
                        if _SPARSE_WGRAD and not defer and live is not None and weight_I.is_contiguous():
                            ent = _WCHUNKS.get(weight_I.data_ptr())
                            nch = int(lib.mrgcn_weight_chunks(plan.handle, F))
                            if ent is None or ent["numel"] != weight_I.numel() or ent["cur"].numel() != nch:
                                ent = dict(cur=torch.zeros(nch, dtype=torch.uint8, device=dev),
                                           ever=torch.zeros(nch, dtype=torch.uint8, device=dev),
                                           n_ever=torch.zeros(1, dtype=torch.int32, device=dev),
                                           n_ever_host=torch.full((1,), -1, dtype=torch.int32).pin_memory(),
                                           numel=weight_I.numel(), slab=plan.num_nodes * F, B=comp_I.shape[1],
                                           fresh=False, dense=False)
                                _WCHUNKS[weight_I.data_ptr()] = ent
                            # with most chunks live the masks only cost (AM shape, nodes numbered at random:
                            # every chunk holds a node with gradient; + 0.16 ms): the count of the previous
                            # step decides, and once dense the parameter stays on the plain path (`ever` is
                            # not maintained there)
                            if not ent["dense"] and int(ent["n_ever_host"][0]) > _WCHUNK_DENSE * nch:
                                ent["dense"] = True
                            if not ent["dense"]:
                                L.check(lib.mrgcn_weight_chunks_live(plan.handle, live.data_ptr(), F,
                                                                     ent["cur"].data_ptr(), ent["ever"].data_ptr(), s),
                                        "mrgcn_weight_chunks_live")
                                torch.sum(ent["ever"], dim=(0,), keepdim=True, dtype=torch.int32, out=ent["n_ever"])
                                ent["n_ever_host"].copy_(ent["n_ever"], non_blocking=True)
                                ent["fresh"] = True
                                chunk_cur = ent["cur"].data_ptr()
                        L.check(lib.mrgcn_basis_mix_bwd_live_f32(
                            plan.handle, dM.data_ptr(), ld, live.data_ptr() if live is not None else 0, chunk_cur,
                            weight_I.data_ptr(), comp_I.data_ptr(),
                            comp_I.shape[1], F, 0 if defer else d_wI.data_ptr(), d_comp.data_ptr(),
                            sq.data_ptr(), s), "mrgcn_basis_mix_bwd_live_f32")
                        if defer:
                            # comp_I is cloned: the optimizer may update the parameter before pass 2
                            _DEFERRED[weight_I.data_ptr()] = dict(
                                numel=weight_I.numel(), plan=plan, dM=dM, ld=ld, comp=comp_I.detach().clone(),
                                B=comp_I.shape[1], F=F, sumsq=sq)
                        else:
                            register_grad_sumsq(d_wI, sq)  # ||dV||^2 came for free with the gradient
                else:
                    # dense (R*N) x F gradient: zero + scatter of the touched rows
                    d_wI = torch.zeros_like(weight_I)
                    d_wI.index_copy_(0, plan.ulcol_long(), dM[:, :F])
            if has_X:
                need_dX = ctx.needs_input_grad[4]
                need_dW = ctx.needs_input_grad[5]
                K = X.shape[1]
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
                        L.check(lib.mrgcn_rel_transform_bwd_live_f32(
                            plan.handle, dM.data_ptr(), ld, live.data_ptr() if live is not None else 0,
                            X.data_ptr(), X.stride(0), K, W_F.data_ptr(), F,
                            dX.data_ptr() if need_dX else 0, K, dW.data_ptr() if need_dW else 0,
                            ws.data_ptr() if ws is not None else 0, ws.numel() if ws is not None else 0,
                            side.cuda_stream), "mrgcn_rel_transform_bwd_live_f32")
                if overlap:
                    main.wait_stream(side)
                    for t in (dX, dW, ws, dM, live):  # allocated / used on `side`: keep the allocator honest
                        if t is not None:
                            t.record_stream(side if (t is dM or t is live) else main)
        return None, None, d_wI, d_comp, dX, dW, dbias, None, None


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
        Xin = X
        W_F = layer.weight_F
        if B > 0:  # graph.py:83-85: tiny (R x B) . (B x in*out) contraction -> library GEMM
            W_F = (layer.weight_F_comp @ W_F.reshape(B, -1)).view(layer.num_relations, layer.indim, F)
    bf16 = getattr(layer, "operand_dtype", "f32") == "bf16"
    bias = layer.b if (layer.bias and use_bias) else None
    if weight_I is not None and comp_I is None and Xin is None and not bf16:
        # featureless layer without bases: weight_I already *is* the literal operand
        return spmm_literal(plan, weight_I, bias=bias, relu=relu)
    return _RgcnLayer.apply(plan, F, weight_I, comp_I, Xin, W_F, bias, relu, bf16)
