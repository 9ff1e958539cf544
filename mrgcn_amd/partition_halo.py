"""Row partition with a halo exchange of OPERAND rows — the second node-partitioned engine ("row
partition + halo exchange of boundary rows ... pick by measured halo size"; north_star: "node-partitioned ... with halo
exchange").

Rank g owns the node range [g*S, (g+1)*S): those OUTPUT rows, those rows of the layer input and of `weight_I` (as in
mrgcn_amd.partition).  Per layer

    forward   M_g   = operand rows of the columns (r, j in range)        local: the mix / transform kernels on P_col
              halo  = the operand rows of REMOTE columns my rows read     one all-to-all (rows of 4 F bytes), started
                                                                          first and in flight under the next line
              Y_g   = A[rows_g, own] . M_g                                local: the product on P_own
              Y_g  += A[rows_g, remote] . halo                            local, after the wait: the product on P_halo
    backward  dM    = A[rows_g, :]^T dY_g                                 local
              the halo columns' gradient rows go back to their owners     one all-to-all (reverse)
              dV_g / dX_g / d(comp, W_F) from the summed dM_g             local; small gradients all-reduced

Against the column partition (one reduce-scatter of `Np x out` partial sums per layer) the exchange carries only the
distinct remote columns a rank's rows read: `tools/halo_probe.py` — less at 8 GPUs on the AM and synth10m shapes
(77.5 vs 122.5 MB, 591 vs 945 MB per layer pass), more at 2-4 GPUs and on FB15k-237.  `choose_partition` picks by
that measure.  This engine computes the backward on dense index spaces (no gradient support yet: every column of
P_own / P_halo gets its gradient row); both exchanges are started first and run under the product over the rank's own
columns (forward) and under that product's backward (reverse exchange); the arithmetic equals `mrgcn_amd.models.rgcn.RGCN` on one GPU (tests).
One process per GPU, torch.distributed (RCCL all-to-all over xGMI; gloo, CPU staged, in the tests)."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib as L
from . import functional as Fn
from .layers.graph import GraphConvolution
from .partition import NodePartition, _staged, all_reduce_sum_, partitioned_loss
from .plan import GraphPlan
from .stats import bump

import os
_COMPOSED = os.environ.get("MRGCN_HALO_COMPOSED", "0") == "1"   # the layer as a composition of autograd pieces (round 5)


# ---- index maps (host, pure numpy + one exchange of requests): testable without a GPU ---------------------------------
def halo_requests(part: NodePartition, rows, cols, num_relations: int):
    """What rank `part.rank` reads: the distinct literal columns (r*N + j, global) of the entries in its own rows, split
    into the ones whose source node it owns and, per owner rank, the remote ones (each list sorted).
    Returns (own_lit, {owner: lit array})."""
    rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
    mine = (rows >= part.j0) & (rows < part.j1)
    lit = np.unique(cols[mine])
    owner = (lit % part.N) // part.S
    own = lit[owner == part.rank]
    return own, {int(o): lit[owner == o] for o in np.unique(owner) if o != part.rank}


def exchange_requests(requests: dict, world: int, rank: int, group=None):
    """Every rank tells every owner which of its columns it needs.  `requests`: {owner: int64 literal ids}.
    Returns {requester: int64 literal ids} — what THIS rank must send, per peer, in the peer's order."""
    # (RCCL moves device tensors only; gloo host tensors)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    counts_out = torch.zeros(world, dtype=torch.int64)
    for o, ids in requests.items():
        counts_out[o] = len(ids)
    counts_in = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(counts_in, counts_out.to(dev), group=group)
    counts_in = counts_in.cpu()
    send = torch.from_numpy(np.concatenate([np.asarray(requests.get(o, np.zeros(0, np.int64)), dtype=np.int64)
                                            for o in range(world)] + [np.zeros(0, np.int64)]))
    recv = torch.empty(int(counts_in.sum()), dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv, send.to(dev), output_split_sizes=counts_in.tolist(),
                           input_split_sizes=counts_out.tolist(), group=group)
    recv = recv.cpu()
    out, at = {}, 0
    for r in range(world):
        n = int(counts_in[r])
        if n:
            out[r] = recv[at:at + n].numpy()
        at += n
    return out


# ---- collectives -------------------------------------------------------------------------------------------------------
def all_to_all_rows(send: torch.Tensor, in_splits, out_splits, group=None) -> torch.Tensor:
    """rows of `send` (grouped by destination rank: in_splits rows each) -> the rows this rank receives (grouped by
    source rank: out_splits rows each)."""
    n_out = int(sum(out_splits))
    if _staged(send, group):
        buf = torch.empty((n_out,) + tuple(send.shape[1:]), dtype=send.dtype)
        dist.all_to_all_single(buf, send.detach().cpu().contiguous(), output_split_sizes=list(out_splits),
                               input_split_sizes=list(in_splits), group=group)
        return buf.to(send.device)
    out = torch.empty((n_out,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    dist.all_to_all_single(out, send.contiguous(), output_split_sizes=list(out_splits), input_split_sizes=list(in_splits),
                           group=group)
    return out


class _AllToAllRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, send, in_splits, out_splits, group):
        ctx.in_splits, ctx.out_splits, ctx.group = in_splits, out_splits, group
        return all_to_all_rows(send, in_splits, out_splits, group)

    @staticmethod
    def backward(ctx, g):
        # the gradient of a received row goes back to the rank that sent it
        return all_to_all_rows(g.contiguous(), ctx.out_splits, ctx.in_splits, ctx.group), None, None, None


class _Pending:
    """a collective in flight: the receive buffer and what to wait on"""
    __slots__ = ("work", "buf", "host")

    def __init__(self):
        self.work = self.buf = self.host = None


def start_rows_exchange(send: torch.Tensor, in_splits, out_splits, group, pending: _Pending) -> torch.Tensor:
    """Starts the all-to-all of `send`'s rows (no autograd: a side effect) and returns the receive buffer at once.  Over
    RCCL the collective runs on the communicator's own stream; `pending.work` is what the consumer waits on.  gloo (the
    tests): staged through the host, blocking."""
    n_out = int(sum(out_splits))
    send = send.detach()
    if _staged(send, group) or dist.get_backend(group) != "nccl":
        pending.buf = all_to_all_rows(send, in_splits, out_splits, group)
        pending.work = None
        return pending.buf
    out = torch.empty((n_out,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    pending.work = dist.all_to_all_single(out, send.contiguous(), output_split_sizes=list(out_splits),
                                          input_split_sizes=list(in_splits), group=group, async_op=True)
    pending.buf = out
    return out


def wait_rows_exchange(pending: _Pending):
    if pending.work is not None:
        pending.work.wait()   # the current stream waits for the communicator's
        pending.work = None


class _SendRows(torch.autograd.Function):
    """send = M[pos] (the operand rows the other ranks read).  Backward: the gradient rows that came back from those
    ranks — `_ExchangedRows.backward` STARTED their exchange and handed the receive buffer on; here, after the backward
    of the product over this rank's own columns (created later in the forward: autograd runs it first), the stream waits
    for it and adds the rows into the operand's gradient."""

    @staticmethod
    def forward(ctx, M, pos, back: _Pending):
        ctx.back, ctx.n = back, M.shape[0]
        ctx.save_for_backward(pos)
        return M.index_select(0, pos)

    @staticmethod
    def backward(ctx, g):
        (pos,) = ctx.saved_tensors
        wait_rows_exchange(ctx.back)
        dM = torch.zeros((ctx.n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        dM.index_add_(0, pos, g)
        return dM, None, None


class _ExchangedRows(torch.autograd.Function):
    """The autograd node of the operand-row exchange.  Forward: the exchange itself was started earlier
    (`start_rows_exchange`, before the product over the rank's own columns); this node waits for it and returns the
    received rows.  It is created AFTER that product, so that in the backward it runs BEFORE the product's: the reverse
    exchange (gradient rows back to their senders) is started here and is in flight under the own product's backward;
    `_SendRows.backward` waits for it."""

    @staticmethod
    def forward(ctx, send, in_splits, out_splits, group, fwd: _Pending, back: _Pending):
        ctx.in_splits, ctx.out_splits, ctx.group, ctx.back = in_splits, out_splits, group, back
        wait_rows_exchange(fwd)
        return fwd.buf.view_as(fwd.buf)

    @staticmethod
    def backward(ctx, g):
        out = start_rows_exchange(g.contiguous(), ctx.out_splits, ctx.in_splits, ctx.group, ctx.back)
        return out, None, None, None, None, None


# ---- the two local halves of a layer as autograd functions -----------------------------------------------------------
def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


class _OperandFn(torch.autograd.Function):
    """M = the compact operand of graph.py:69-94 for the columns of `plan` (operand order, ld = functional._ld_for(F)):
    M[c] = comp_I[r_c] . V_I[j_c] (or weight_I[r_c*N + j_c]) + X[j_c] . W_F[r_c].  The backward takes the gradient of
    those rows and runs the plan-level kernels (no gradient-sparsity shortcut)."""

    @staticmethod
    def forward(ctx, plan: GraphPlan, F: int, weight_I, comp_I, X, W_F):
        lib = L.load()
        dev = plan.device
        ld = Fn._ld_for(F)
        M = torch.empty((plan.nop, ld), dtype=torch.float32, device=dev)
        s = _stream(dev)
        Xc = Wc = None
        with torch.cuda.device(dev):
            addend, ldA = 0, 0
            if X is not None:
                Xc = X if (X.dim() == 2 and X.stride(1) == 1) else X.contiguous()
                Wc = W_F.contiguous()
                if weight_I is not None:
                    ldA = (F + 3) // 4 * 4
                    M2 = torch.empty((plan.ncols, ldA), dtype=torch.float32, device=dev)
                    out, ldo, order = M2, ldA, 0
                    addend = M2.data_ptr()
                else:
                    out, ldo, order = M, ld, 1
                L.check(lib.mrgcn_rel_transform_fwd_f32(plan.handle, Xc.data_ptr(), Xc.stride(0), Xc.shape[1],
                                                        Wc.data_ptr(), F, out.data_ptr(), ldo, order, s),
                        "mrgcn_rel_transform_fwd_f32")
            if weight_I is not None:
                wI = weight_I.contiguous()
                if comp_I is not None:
                    cI = comp_I.contiguous()
                    L.check(lib.mrgcn_basis_mix_fwd_f32(plan.handle, wI.data_ptr(), cI.data_ptr(), cI.shape[1], F, addend,
                                                        ldA, M.data_ptr(), ld, s), "mrgcn_basis_mix_fwd_f32")
                else:
                    L.check(lib.mrgcn_gather_rows_f32(plan.handle, wI.data_ptr(), F, addend, ldA, M.data_ptr(), ld, s),
                            "mrgcn_gather_rows_f32")
        plan.replicate(M)
        ctx.plan, ctx.F = plan, F
        ctx.has = (weight_I is not None, comp_I is not None, X is not None)
        ctx.save_for_backward(weight_I, comp_I, Xc, Wc)
        return M

    @staticmethod
    def backward(ctx, dM_op):
        lib = L.load()
        plan, F = ctx.plan, ctx.F
        weight_I, comp_I, X, W_F = ctx.saved_tensors
        has_I, has_comp, has_X = ctx.has
        dev = plan.device
        s = _stream(dev)
        ld = (F + 3) // 4 * 4
        # the gradient rows in plain compact order, as the node-major consumers read them
        dM = torch.zeros((plan.ncols, ld), dtype=torch.float32, device=dev)
        dM[:, :F] = dM_op.index_select(0, _mpos_long(plan))[:, :F]
        d_wI = d_comp = dX = dW = None
        with torch.cuda.device(dev):
            if has_I and has_comp:
                wI = weight_I.contiguous()
                Bn = wI.shape[1]
                d_wI, d_comp = torch.empty_like(wI), torch.empty_like(comp_I)
                L.check(lib.mrgcn_basis_mix_bwd_f32(plan.handle, dM.data_ptr(), ld, 0, wI.data_ptr(),
                                                    comp_I.contiguous().data_ptr(), Bn, F, d_wI.data_ptr(), 0,
                                                    d_comp.data_ptr(), 0, s), "mrgcn_basis_mix_bwd_f32")
            elif has_I:
                d_wI = torch.zeros_like(weight_I)
                d_wI.index_copy_(0, plan.ulcol_long(), dM[:, :F])
            if has_X:
                need_dX, need_dW = ctx.needs_input_grad[4], ctx.needs_input_grad[5]
                K = X.shape[1]
                if need_dX or need_dW:
                    nws = int(lib.mrgcn_rel_transform_bwd_workspace(plan.handle, K, F, int(need_dX), int(need_dW)))
                    ws = torch.empty((max(nws, 1),), dtype=torch.float32, device=dev)
                    if need_dX:
                        dX = torch.empty((X.shape[0], K), dtype=torch.float32, device=dev)
                    if need_dW:
                        dW = torch.empty_like(W_F)
                    L.check(lib.mrgcn_rel_transform_bwd_masked_f32(
                        plan.handle, dM.data_ptr(), ld, 0, X.data_ptr(), X.stride(0), K, W_F.data_ptr(), F,
                        dX.data_ptr() if need_dX else 0, K, dW.data_ptr() if need_dW else 0, ws.data_ptr(), nws, 0, 0, 0,
                        s), "mrgcn_rel_transform_bwd_masked_f32")
        return None, None, d_wI, d_comp, dX, dW


def _mpos_long(plan: GraphPlan) -> torch.Tensor:
    t = plan.__dict__.get("_mpos_long")
    if t is None:
        t = plan.__dict__["_mpos_long"] = torch.from_numpy(plan.export(L.ARR_MPOS).astype(np.int64)).to(plan.device)
    return t


class _ProductFn(torch.autograd.Function):
    """Y = relu?( A' . M + b ) on the COMPACT view of `plan` (M in operand order); backward: A'^T dY by compact column,
    scattered back into operand order."""

    @staticmethod
    def forward(ctx, plan: GraphPlan, M, F: int, bias, relu: bool):
        Y = plan.spmm(L.VIEW_COMPACT, M.contiguous(), F=F, bias=bias, relu=relu)
        ctx.plan, ctx.F, ctx.relu, ctx.has_bias, ctx.shape = plan, F, relu, bias is not None, tuple(M.shape)
        ctx.save_for_backward(Y if relu else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        plan, F = ctx.plan, ctx.F
        (Y,) = ctx.saved_tensors
        dY = dY.contiguous()
        if ctx.relu:
            dY = Fn.relu_bwd(dY, Y)
        dbias = Fn._bias_grad(dY) if ctx.has_bias else None
        ld = (F + 3) // 4 * 4
        dMc = torch.empty((plan.ncols, ld), dtype=torch.float32, device=dY.device)
        plan.spmm(L.VIEW_TRANSPOSED, dY, F=F, out=dMc)
        dM = torch.zeros(ctx.shape, dtype=torch.float32, device=dY.device)
        dM[:, :F].index_copy_(0, _mpos_long(plan), dMc[:, :F])
        return None, dM, None, dbias, None


# ---- the whole layer as ONE autograd function: forward with the exchange under the own product, backward on gradient
# ---- supports (round 6) ------------------------------------------------------------------------------------------------
def _operand_forward(plan: GraphPlan, F: int, weight_I, comp_I, X, W_F):
    """_OperandFn.forward's kernels: the compact operand of `plan`'s columns in operand order.  Returns (M, Xc, Wc)."""
    lib = L.load()
    dev = plan.device
    ld = Fn._ld_for(F)
    M = torch.empty((plan.nop, ld), dtype=torch.float32, device=dev)
    s = _stream(dev)
    Xc = Wc = None
    with torch.cuda.device(dev):
        addend, ldA = 0, 0
        if X is not None:
            Xc = X if (X.dim() == 2 and X.stride(1) == 1) else X.contiguous()
            Wc = W_F.contiguous()
            if weight_I is not None:
                ldA = (F + 3) // 4 * 4
                M2 = torch.empty((plan.ncols, ldA), dtype=torch.float32, device=dev)
                out, ldo, order = M2, ldA, 0
                addend = M2.data_ptr()
            else:
                out, ldo, order = M, ld, 1
            L.check(lib.mrgcn_rel_transform_fwd_f32(plan.handle, Xc.data_ptr(), Xc.stride(0), Xc.shape[1],
                                                    Wc.data_ptr(), F, out.data_ptr(), ldo, order, s),
                    "mrgcn_rel_transform_fwd_f32")
        if weight_I is not None:
            wI = weight_I.contiguous()
            if comp_I is not None:
                cI = comp_I.contiguous()
                L.check(lib.mrgcn_basis_mix_fwd_f32(plan.handle, wI.data_ptr(), cI.data_ptr(), cI.shape[1], F, addend,
                                                    ldA, M.data_ptr(), ld, s), "mrgcn_basis_mix_fwd_f32")
            else:
                L.check(lib.mrgcn_gather_rows_f32(plan.handle, wI.data_ptr(), F, addend, ldA, M.data_ptr(), ld, s),
                        "mrgcn_gather_rows_f32")
    return M, Xc, Wc


class _Ctx:
    """what functional._support_backward_from_dM reads off an autograd context"""
    __slots__ = ("plan", "F", "saved_tensors", "has", "needs_input_grad", "owner", "x_is_relu_out", "Xb")


class _HaloLayerFn(torch.autograd.Function):
    """One `GraphConvolution` of the halo engine: Y_g = relu?( A[rows_g, own] . M_g + A[rows_g, remote] . halo + b ).

    forward: the operand rows of MY columns (P_col: mix / transform kernels), the rows the other ranks read sent off
    first, the product over my own columns (P_own) under the exchange, the product over the received rows (P_halo)
    behind the wait.  The operands of P_own / P_halo are persistent buffers filled through index maps (no zero fill,
    no autograd bookkeeping per piece).

    backward: when the rows of dY that can hold anything are known STRUCTURALLY (the loss's label flags, below that
    the node flags of the layer above — functional._grad_meta), everything runs on gradient supports, as on one GPU:
    `A[rows_g, own]^T dY` and `A[rows_g, remote]^T dY` over the LIVE columns only (mrgcn_support_spmm_t_f32), only the
    live halo columns' gradient rows go back to their owners, the owner adds them to its own live rows by live number
    and runs the mix / transform backward on P_col's support (functional._support_backward_from_dM) — row-sparse
    weight_I gradient and fused row Adam included.  Otherwise (a dense gradient: the link-prediction decoder reads every
    row) the general transposed products and the plan-level backward, in P_col's compact order."""

    @staticmethod
    def forward(ctx, hp, group, F: int, relu: bool, weight_I, comp_I, X, W_F, bias, owner):
        M_col, Xc, Wc = _operand_forward(hp.p_col, F, weight_I, comp_I, X, W_F)
        ld = M_col.shape[1]
        pend = _Pending()
        send = M_col.index_select(0, hp.send_pos)
        start_rows_exchange(send, hp.in_splits, hp.out_splits, group, pend)
        M_own = hp.buffer("own", hp.p_own, ld)
        M_own.index_copy_(0, hp.own_dst, M_col.index_select(0, hp.own_src))
        Y = hp.p_own.spmm(L.VIEW_COMPACT, M_own, F=F)
        wait_rows_exchange(pend)
        if hp.halo_columns > 0:
            M_halo = hp.buffer("halo", hp.p_halo, ld)
            M_halo.index_copy_(0, hp.halo_dst, pend.buf)
            Y += hp.p_halo.spmm(L.VIEW_COMPACT, M_halo, F=F)
        if bias is not None:
            Y += bias
        if relu:
            Y = torch.relu_(Y)
        ctx.hp, ctx.group, ctx.F, ctx.relu, ctx.owner = hp, group, F, relu, owner
        ctx.has = (weight_I is not None, comp_I is not None, X is not None, bias is not None)
        ctx.x_is_relu_out = X is not None and bool(getattr(X, "_mrgcn_relu_out", False))
        ctx.save_for_backward(weight_I, comp_I, Xc, Wc, Y if relu else None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = L.load()
        hp, group, F = ctx.hp, ctx.group, ctx.F
        weight_I, comp_I, X, W_F, Y = ctx.saved_tensors
        has_I, has_comp, has_X, has_bias = ctx.has
        dev = hp.device
        meta = Fn._grad_meta(dY)
        dY = dY.contiguous()
        if ctx.relu and not (meta is not None and meta.get("relu_applied")):
            dY = Fn.relu_bwd(dY, Y)
        flags = meta["row_live"] if (meta and meta.get("structural")) else None
        dbias = Fn._bias_grad(dY, flags) if has_bias else None
        ld = (F + 3) // 4 * 4
        s = _stream(dev)
        shim = _Ctx()
        shim.plan, shim.F, shim.has, shim.owner, shim.Xb = hp.p_col, F, ctx.has, ctx.owner, None
        shim.saved_tensors, shim.x_is_relu_out = ctx.saved_tensors, ctx.x_is_relu_out
        # (positions of _RgcnLayer.forward's arguments: [4] = X, [5] = W_F)
        shim.needs_input_grad = (False, False, ctx.needs_input_grad[4], ctx.needs_input_grad[5], ctx.needs_input_grad[6],
                                 ctx.needs_input_grad[7], False, False, False, False)
        use_sup = (flags is not None and F <= 16 and Fn._SUPPORT and Fn._LIVE_COLS and flags.dtype == torch.uint8
                   and flags.numel() == hp.p_own.num_rows and flags.device == dev)
        live = nws = None
        if use_sup:
            live = hp.live(flags)
            nws = Fn._support_backward_workspace(shim, live["sup_col"])
        if live is not None and nws is not None:
            bump("backward.support")
            bump("halo.backward.support")
            sup_col, sup_own, sup_halo = live["sup_col"], live["sup_own"], live["sup_halo"]
            dM_col = torch.zeros((max(sup_col.L, 1), ld), dtype=torch.float32, device=dev)
            back = _Pending()
            with torch.cuda.device(dev):
                if sup_halo is not None:   # the remote columns' part first: its rows travel under the own part
                    dM_h = torch.empty((max(sup_halo.L, 1), ld), dtype=torch.float32, device=dev)
                    L.check(lib.mrgcn_support_spmm_t_f32(sup_halo.handle, dY.data_ptr(), dY.stride(0), F, dM_h.data_ptr(),
                                                         ld, s), "mrgcn_support_spmm_t_f32")
                    send = dM_h.index_select(0, live["halo_perm"])
                else:
                    send = torch.empty((0, ld), dtype=torch.float32, device=dev)
                start_rows_exchange(send, live["back_in"], live["back_out"], group, back)
                dM_o = torch.empty((max(sup_own.L, 1), ld), dtype=torch.float32, device=dev)
                L.check(lib.mrgcn_support_spmm_t_f32(sup_own.handle, dY.data_ptr(), dY.stride(0), F, dM_o.data_ptr(), ld,
                                                     s), "mrgcn_support_spmm_t_f32")
            if sup_own.L > 0:
                dM_col.index_copy_(0, live["own_to_col"], dM_o[: sup_own.L])   # (one own row per live column at most)
            wait_rows_exchange(back)
            if back.buf.shape[0] > 0:
                dM_col.index_add_(0, live["recv_to_col"], back.buf)
            out = Fn._support_backward_from_dM(shim, sup_col, dM_col, ld, dbias, nws)
            _, _, d_wI, d_comp, dX, dW, dbias, _, _, _ = out
            return None, None, None, None, d_wI, d_comp, dX, dW, dbias, None
        # ---- dense gradient: general transposed products, plan-level backward in P_col's compact order ----------------
        bump("halo.backward.dense")
        p_col = hp.p_col
        dM = torch.zeros((p_col.ncols, ld), dtype=torch.float32, device=dev)
        back = _Pending()
        if hp.halo_columns > 0:
            dMh = torch.empty((hp.p_halo.ncols, ld), dtype=torch.float32, device=dev)
            hp.p_halo.spmm(L.VIEW_TRANSPOSED, dY, F=F, out=dMh[:, :F])
            send = dMh.index_select(0, hp.halo_cid)
        else:
            send = torch.empty((0, ld), dtype=torch.float32, device=dev)
        start_rows_exchange(send, hp.out_splits, hp.in_splits, group, back)
        dMo = torch.empty((hp.p_own.ncols, ld), dtype=torch.float32, device=dev)
        hp.p_own.spmm(L.VIEW_TRANSPOSED, dY, F=F, out=dMo[:, :F])
        dM[:, :F].index_copy_(0, hp.own_c2c, dMo[:, :F])
        wait_rows_exchange(back)
        if back.buf.shape[0] > 0:
            dM[:, :F].index_add_(0, hp.send_cid, back.buf[:, :F])
        d_wI = d_comp = dX = dW = None
        with torch.cuda.device(dev):
            if has_I and has_comp:
                wI = weight_I.contiguous()
                Bn = wI.shape[1]
                d_wI, d_comp = torch.empty_like(wI), torch.empty_like(comp_I)
                L.check(lib.mrgcn_basis_mix_bwd_f32(p_col.handle, dM.data_ptr(), ld, 0, wI.data_ptr(),
                                                    comp_I.contiguous().data_ptr(), Bn, F, d_wI.data_ptr(), 0,
                                                    d_comp.data_ptr(), 0, s), "mrgcn_basis_mix_bwd_f32")
            elif has_I:
                d_wI = torch.zeros_like(weight_I)
                d_wI.index_copy_(0, p_col.ulcol_long(), dM[:, :F])
            if has_X:
                need_dX, need_dW = ctx.needs_input_grad[6], ctx.needs_input_grad[7]
                K = X.shape[1]
                if need_dX or need_dW:
                    nw = int(lib.mrgcn_rel_transform_bwd_workspace(p_col.handle, K, F, int(need_dX), int(need_dW)))
                    ws = torch.empty((max(nw, 1),), dtype=torch.float32, device=dev)
                    if need_dX:
                        dX = torch.empty((X.shape[0], K), dtype=torch.float32, device=dev)
                    if need_dW:
                        dW = torch.empty_like(W_F)
                    L.check(lib.mrgcn_rel_transform_bwd_masked_f32(
                        p_col.handle, dM.data_ptr(), ld, 0, X.data_ptr(), X.stride(0), K, W_F.data_ptr(), F,
                        dX.data_ptr() if need_dX else 0, K, dW.data_ptr() if need_dW else 0, ws.data_ptr(), nw, 0, 0, 0,
                        s), "mrgcn_rel_transform_bwd_masked_f32")
        return None, None, None, None, d_wI, d_comp, dX, dW, dbias, None


# ---- the partition's index maps on the device ------------------------------------------------------------------------
class HaloPlans:
    """P_col (the columns whose source node this rank owns, over ALL rows: operand construction and its backward),
    P_own (this rank's rows over its OWN columns) and P_halo (this rank's rows over the REMOTE columns they read): the
    product of a layer is `P_own . M_own + P_halo . M_halo`, and the exchange that fills `M_halo` is in flight while the
    first product runs.  Plus the maps between those operands and the exchange buffers.  The plans are built WITHOUT
    operand replicas whatever the library's default says: the operands of `P_own` / `P_halo` are filled through index
    maps over their MPOS rows, a replica row would stay zero."""

    def __init__(self, part: NodePartition, rows, cols, vals, num_relations: int, device, operand_row_bytes, group=None):
        R, N, S, rank, world = num_relations, part.N, part.S, part.rank, part.world
        rows, cols, vals = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64), np.asarray(vals)
        # P_col: as mrgcn_amd.partition (rows global, columns r*S + local node)
        lr, lc, lv = part.local_coo(rows, cols, vals, R)
        A_col = torch.sparse_coo_tensor(torch.from_numpy(np.stack([lr, lc])), torch.from_numpy(lv),
                                        (part.Np, R * S)).to(device)
        self.p_col = GraphPlan(A_col, S, R, operand_row_bytes=operand_row_bytes, replicate=False)
        del A_col
        # my rows, split by who owns the source node
        mine = (rows >= part.j0) & (rows < part.j1)
        er, ec, ev = rows[mine] - part.j0, cols[mine], vals[mine]
        rel, node = ec // N, ec % N
        is_own = (node >= part.j0) & (node < part.j1)
        A_own = torch.sparse_coo_tensor(torch.from_numpy(np.stack([er[is_own], rel[is_own] * S + (node[is_own] - part.j0)])),
                                        torch.from_numpy(ev[is_own]), (S, R * S)).to(device)
        self.p_own = GraphPlan(A_own, S, R, operand_row_bytes=operand_row_bytes, replicate=False)
        del A_own
        remote = np.unique(node[~is_own])          # the remote source nodes my rows read (rising id)
        Nr = max(len(remote), 1)
        jr = np.searchsorted(remote, node[~is_own])
        A_halo = torch.sparse_coo_tensor(torch.from_numpy(np.stack([er[~is_own], rel[~is_own] * Nr + jr])),
                                         torch.from_numpy(ev[~is_own]), (S, R * Nr)).to(device)
        self.p_halo = GraphPlan(A_halo, Nr, R, operand_row_bytes=operand_row_bytes, replicate=False)
        del A_halo
        # P_col: literal id (global) of its compact columns -> operand position
        ulc = self.p_col.export(L.ARR_ULCOL).astype(np.int64)
        lit_col = (ulc // S) * N + (ulc % S) + part.j0
        mpos_col = self.p_col.export(L.ARR_MPOS).astype(np.int64)
        order = np.argsort(lit_col, kind="stable")
        lit_sorted = lit_col[order]

        def col_compact(lits):     # compact ids in P_col of the given global literal columns
            at = np.searchsorted(lit_sorted, lits)
            if len(lits) and (at.max() >= len(lit_sorted) or not np.array_equal(lit_sorted[at], lits)):
                raise L.MrgcnError("halo partition: a requested column is not a column of its owner")
            return order[at]

        def col_positions(lits):   # operand positions in M_col of the given global literal columns
            return mpos_col[col_compact(lits)]
        self._col_compact = col_compact
        # P_own's columns inside M_col
        ulo = self.p_own.export(L.ARR_ULCOL).astype(np.int64)
        lit_own = (ulo // S) * N + (ulo % S) + part.j0
        self.own_src = torch.from_numpy(col_positions(lit_own)).to(device)
        self.own_dst = torch.from_numpy(self.p_own.export(L.ARR_MPOS).astype(np.int64)).to(device)
        self._own_c2c = col_compact(lit_own)                       # P_own compact column -> P_col compact column (host)
        self.own_c2c = torch.from_numpy(self._own_c2c).to(device)
        # P_halo's columns: global literal id, owner; requests per owner in rising literal id
        ulh = self.p_halo.export(L.ARR_ULCOL).astype(np.int64)
        node_h = remote[ulh % Nr] if len(remote) else np.zeros(0, np.int64)
        lit_h = (ulh // Nr) * N + node_h
        mpos_h = self.p_halo.export(L.ARR_MPOS).astype(np.int64)
        own_of = node_h // S
        req, halo_dst, halo_cid = {}, [], []
        for o in range(world):
            sel = own_of == o
            if sel.any():
                ordr = np.argsort(lit_h[sel], kind="stable")
                req[o] = lit_h[sel][ordr]
                halo_dst.append(mpos_h[sel][ordr])
                halo_cid.append(np.flatnonzero(sel)[ordr])
        self.out_splits = [len(req.get(o, ())) for o in range(world)]
        self.halo_dst = torch.from_numpy(np.concatenate(halo_dst) if halo_dst else np.zeros(0, np.int64)).to(device)
        # P_halo compact column of every received row (request order) — the dense backward's send order
        self.halo_cid = torch.from_numpy(np.concatenate(halo_cid) if halo_cid else np.zeros(0, np.int64)).to(device)
        self._lit_h, self._own_of = lit_h, own_of
        asked = exchange_requests(req, world, rank, group)
        self.in_splits = [len(asked.get(r, ())) for r in range(world)]
        self.send_pos = torch.from_numpy(np.concatenate([col_positions(asked[r]) for r in range(world) if r in asked]
                                                        + [np.zeros(0, np.int64)])).to(device)
        # ... and the P_col compact column of every row this rank sends (the dense backward adds the returned rows there)
        self.send_cid = torch.from_numpy(np.concatenate([col_compact(asked[r]) for r in range(world) if r in asked]
                                                        + [np.zeros(0, np.int64)])).to(device)
        self.halo_columns = int(sum(self.out_splits))
        self.world, self.rank, self.group, self.device = world, rank, group, device
        self._buffers, self._live = {}, {}

    def buffer(self, which: str, plan: GraphPlan, ld: int) -> torch.Tensor:
        """The operand of P_own / P_halo: one zeroed buffer per row width, kept — the rows outside MPOS are never
        written, the rows inside are overwritten by every forward."""
        key = (which, ld)
        t = self._buffers.get(key)
        if t is None:
            t = self._buffers[key] = torch.zeros((plan.nop, ld), dtype=torch.float32, device=self.device)
        return t

    def live(self, flags: torch.Tensor):
        """Everything the backward on GRADIENT SUPPORTS needs for one structural row set (`flags`: this rank's rows that
        can hold gradient; the same tensor every epoch), built once — a collective: every rank reaches it in the same
        backward (partitioned_loss hands every rank structural flags).  The supports of the three plans, the live
        numbers in P_col's support of P_own's live columns, the live halo columns grouped by owner (the reverse
        exchange carries only those) and, on the owner's side, the live numbers the returned rows are added to."""
        from .partition import _gathered_flags
        ent = self._live.get(id(flags))
        if ent is not None and ent["flags"] is flags and ent["version"] == flags._version:
            return ent
        world, rank = self.world, self.rank
        gflags = _gathered_flags(flags, self.group) if world > 1 else flags
        sup_col = self.p_col.support_for(gflags)
        sup_own = self.p_own.support_for(flags)
        sup_halo = self.p_halo.support_for(flags) if self.halo_columns > 0 else None
        live_col = np.flatnonzero(sup_col.export(L.SUP_COL_FLAGS))          # compact ids of P_col, rising = live numbers
        live_own = np.flatnonzero(sup_own.export(L.SUP_COL_FLAGS))
        own_to_col = np.searchsorted(live_col, self._own_c2c[live_own])
        if len(live_own) and not np.array_equal(live_col[np.minimum(own_to_col, max(len(live_col) - 1, 0))],
                                                self._own_c2c[live_own]):
            raise L.MrgcnError("halo partition: a live column of P_own is not live in P_col")
        req, perm = {}, []
        if sup_halo is not None:
            live_h = np.flatnonzero(sup_halo.export(L.SUP_COL_FLAGS))       # compact ids of P_halo, rising = live numbers
            lit, own_of = self._lit_h[live_h], self._own_of[live_h]
            for o in range(world):
                sel = own_of == o
                if sel.any():
                    ordr = np.argsort(lit[sel], kind="stable")
                    req[o] = lit[sel][ordr]
                    perm.append(np.flatnonzero(sel)[ordr])                  # live numbers of P_halo's support, owner order
        back_in = [len(req.get(o, ())) for o in range(world)]                # rows I send back, per owner
        asked = exchange_requests(req, world, rank, self.group) if world > 1 else {}
        back_out = [len(asked.get(r, ())) for r in range(world)]             # rows that come back to me, per reader
        recv_c = np.concatenate([self._col_compact(asked[r]) for r in range(world) if r in asked] + [np.zeros(0, np.int64)])
        recv_to_col = np.searchsorted(live_col, recv_c)
        if len(recv_c) and not np.array_equal(live_col[np.minimum(recv_to_col, max(len(live_col) - 1, 0))], recv_c):
            raise L.MrgcnError("halo partition: a returned gradient row belongs to a column that is not live at its owner")
        dev = self.device
        ent = dict(flags=flags, version=flags._version, sup_col=sup_col, sup_own=sup_own, sup_halo=sup_halo,
                   own_to_col=torch.from_numpy(own_to_col.astype(np.int64)).to(dev),
                   halo_perm=torch.from_numpy((np.concatenate(perm) if perm else np.zeros(0, np.int64)).astype(np.int64)).to(dev),
                   recv_to_col=torch.from_numpy(recv_to_col.astype(np.int64)).to(dev),
                   back_in=back_in, back_out=back_out)
        self._live[id(flags)] = ent
        return ent

    def exchange_bytes(self, F: int) -> int:
        """bytes this rank RECEIVES per layer pass (forward; the backward returns as many)"""
        return self.halo_columns * Fn._ld_for(F) * 4


# ---- the model -----------------------------------------------------------------------------------------------------------
class HaloPartitionedRGCN(nn.Module):
    """`RGCN` (models/rgcn.py) with the row partition + operand-row halo exchange.  Same constructor and sharding of the
    parameters as `partition.PartitionedRGCN`."""

    def __init__(self, modules, num_relations, num_nodes, num_bases, featureless, bias, part: NodePartition, group=None,
                 link_prediction=False):
        super().__init__()
        if link_prediction:
            self.relations = nn.Parameter(torch.empty((num_relations, modules[-1][1])))
            nn.init.xavier_uniform_(self.relations)
        self.part, self.group = part, group
        self.num_nodes, self.num_relations, self.num_bases = num_nodes, num_relations, num_bases
        self.layers = nn.ModuleDict()
        self.relu = []
        for i, (indim, outdim, _t, act) in enumerate(modules):
            first = i == 0
            self.layers[f"layer_{i}"] = GraphConvolution(
                indim, outdim, num_relations, part.S, num_bases=num_bases, bias=bias, input_layer=first,
                featureless=featureless if first else False)
            self.relu.append(isinstance(act, nn.ReLU))
        self.num_layers = len(self.layers)
        self.plans = None

    # (the same sharding helpers as the column-partition model)
    def sharded_parameters(self):
        return [l.weight_I for l in self.layers.values() if l.weight_I is not None]

    def replicated_parameters(self):
        sh = {id(p) for p in self.sharded_parameters()}
        return [p for p in self.parameters() if id(p) not in sh]

    @torch.no_grad()
    def load_full_state(self, state: dict):
        if "relations" in state and hasattr(self, "relations"):
            self.relations.copy_(state["relations"].to(self.relations.device))
        for i, layer in enumerate(self.layers.values()):
            for name, p in layer.named_parameters():
                full = state[f"layers.layer_{i}.{name}"].to(p.device)
                if name == "weight_I":
                    S_b = self.num_bases if self.num_bases > 0 else self.num_relations
                    p.copy_(self.part.shard_weight_I(full, S_b, node_major=layer.weight_I_node_major))
                else:
                    p.copy_(full)

    def build_plan(self, rows, cols, vals, device):
        rb = sorted({l.operand_row_bytes() for l in self.layers.values()})
        self.plans = HaloPlans(self.part, rows, cols, vals, self.num_relations, device, rb, self.group)
        return self.plans

    def forward(self, X_local):
        if not _COMPOSED:
            return self._forward_fused(X_local)
        return self._forward_composed(X_local)

    def _forward_fused(self, X_local):
        """every layer one autograd function (`_HaloLayerFn`): backward on gradient supports"""
        hp = self.plans
        H = X_local
        for i, layer in enumerate(self.layers.values()):
            F, B = layer.outdim, layer.num_bases
            weight_I = comp_I = Xin = W_F = None
            if layer.input_layer:
                weight_I = layer.weight_I
                comp_I = layer.weight_I_comp if B > 0 else None
            if not (layer.input_layer and layer.featureless):
                Xin, W_F = H, layer.weight_F
                if B > 0:
                    W_F = Fn._BasisContract.apply(layer.weight_F_comp, W_F)
            H = _HaloLayerFn.apply(hp, self.group, F, bool(self.relu[i]), weight_I, comp_I, Xin, W_F,
                                   layer.b if layer.bias else None, layer)
            if self.relu[i]:
                H._mrgcn_relu_out = True
        return H

    def _forward_composed(self, X_local):
        """the round-5 form: the layer composed of autograd pieces (MRGCN_HALO_COMPOSED=1; dense backward)"""
        hp = self.plans
        H = X_local
        for i, layer in enumerate(self.layers.values()):
            F, B = layer.outdim, layer.num_bases
            weight_I = comp_I = Xin = W_F = None
            if layer.input_layer:
                weight_I = layer.weight_I
                comp_I = layer.weight_I_comp if B > 0 else None
            if not (layer.input_layer and layer.featureless):
                Xin, W_F = H, layer.weight_F
                if B > 0:
                    W_F = Fn._BasisContract.apply(layer.weight_F_comp, W_F)
            M_col = _OperandFn.apply(hp.p_col, F, weight_I, comp_I, Xin, W_F)           # my columns' operand rows
            # the rows the other ranks read leave first; the product over my own columns runs under the exchange — and
            # its backward under the reverse exchange (see _ExchangedRows)
            fwd, back = _Pending(), _Pending()
            send = _SendRows.apply(M_col, hp.send_pos, back)
            start_rows_exchange(send, hp.in_splits, hp.out_splits, self.group, fwd)
            M_own = torch.zeros((hp.p_own.nop, M_col.shape[1]), dtype=torch.float32, device=M_col.device)
            M_own = M_own.index_copy(0, hp.own_dst, M_col.index_select(0, hp.own_src))
            b = layer.b if layer.bias else None
            Y = _ProductFn.apply(hp.p_own, M_own, F, None, False)
            recv = _ExchangedRows.apply(send, hp.in_splits, hp.out_splits, self.group, fwd, back)
            if hp.halo_columns > 0:
                M_halo = torch.zeros((hp.p_halo.nop, M_col.shape[1]), dtype=torch.float32, device=M_col.device)
                M_halo = M_halo.index_copy(0, hp.halo_dst, recv)
                Y = Y + _ProductFn.apply(hp.p_halo, M_halo, F, None, False)
            else:   # (a rank that reads no remote column still takes part in the backward's exchange)
                Y = Y + recv.sum() * 0.0
            if b is not None:
                Y = Y + b
            H = torch.relu(Y) if self.relu[i] else Y
        return H

    @torch.no_grad()
    def sync_replicated(self, src: int = 0):
        for q in self.replicated_parameters():
            if _staged(q, self.group):
                buf = q.detach().cpu()
                dist.broadcast(buf, src, group=self.group)
                q.copy_(buf.to(q.device))
            else:
                dist.broadcast(q.data, src, group=self.group)

    def allreduce_replicated_grads(self):
        for p in self.replicated_parameters():
            if p.grad is not None:
                all_reduce_sum_(p.grad, self.group)


def halo_train_step(model: HaloPartitionedRGCN, X_local, idx_global, targets, optimizer, row_sparse=None):
    """One full-batch epoch on the halo engine: dense gradients (`.grad` of every parameter), the replicated ones
    all-reduced, ClipAdam with the shards' norms added (set_distributed)."""
    from .train import ClipAdam, _ROW_SPARSE_DEFAULT
    logits = model(X_local)
    local, total = partitioned_loss(logits, idx_global, targets, model.part, model.group)
    params = [p for g in optimizer.param_groups for p in g["params"]]
    Fn.clear_row_grads(params)
    optimizer.zero_grad(set_to_none=True)
    # (as partition._backward_and_step: row-sparse weight_I gradient + fused row Adam on this rank's node blocks when
    # the backward runs on gradient supports)
    sparse_ok = (row_sparse is not False and not _COMPOSED and _ROW_SPARSE_DEFAULT and isinstance(optimizer, ClipAdam)
                 and all(float(g["weight_decay"]) == 0.0 for g in optimizer.param_groups))
    prev = Fn.row_sparse_weight_grad(sparse_ok)
    try:
        local.backward()
    finally:
        Fn.row_sparse_weight_grad(prev)
    model.allreduce_replicated_grads()
    optimizer.step()
    return total


def halo_lp_step(model: HaloPartitionedRGCN, X_local, triples, labels, optimizer):
    """The link-prediction step of partition.partitioned_lp_step on the halo engine."""
    from .partition import _AllGatherRows
    from .tasks import link_prediction as lp
    part, group = model.part, model.group
    world, rank = part.world, part.rank
    E = _AllGatherRows.apply(model(X_local), group)[: part.N]
    n = triples.shape[0]
    mine = torch.arange(rank, n, world, device=triples.device)
    t = triples[mine]
    sc = lp.score_distmult_bc((t[:, 0], t[:, 1], t[:, 2]), E, model.relations)
    local = lp.binary_crossentropy(sc, labels[mine]) * (float(mine.numel()) / n) if mine.numel() else (E * 0.0).sum()
    total = local.detach().clone()
    all_reduce_sum_(total, group)
    optimizer.zero_grad(set_to_none=True)
    prev = Fn.row_sparse_weight_grad(False)
    try:
        local.backward()
    finally:
        Fn.row_sparse_weight_grad(prev)
    model.allreduce_replicated_grads()
    optimizer.step()
    return total


def choose_partition(rows, cols, num_nodes: int, world: int, layer_widths) -> dict:
    """Bytes one rank RECEIVES per forward of all layers under either engine (the probe of tools/halo_probe.py as a
    library call): {"column": bytes, "halo": bytes, "choice": "column" | "halo"}.  Host arithmetic on the COO arrays."""
    rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
    S = (num_nodes + world - 1) // world
    src = cols % num_nodes
    ro, so = rows // S, src // S
    halo_cols = [int(np.unique(cols[(ro == r) & (so != r)]).size) for r in range(world)]
    Np = S * world
    col_b = sum((world - 1) / world * Np * w * 4 for w in layer_widths)
    halo_b = sum(float(np.mean(halo_cols)) * w * 4 for w in layer_widths)
    return {"column": col_b, "halo": halo_b, "choice": "halo" if halo_b < col_b else "column",
            "halo_columns_per_rank": halo_cols}
