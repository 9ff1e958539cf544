"""Full-batch epoch driver: the MI355X counterpart of the train step of
mrgcn/tasks/node_classification.py:154-193 —

    Y_hat = model(batch); loss = CE(Y_hat[idx], targets)
    zero_grad; backward; clip_grad_norm_(params, 1.0); Adam.step

with the loss, the global gradient norm, the clip and Adam running as HIP kernels
(csrc/optim.hip).  The clip coefficient never leaves the device, so one epoch has no
host synchronisation."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from .functional import clear_row_grads, dense_from_rows, pop_row_grad, row_sparse_weight_grad
from .stats import bump

# MRGCN_MULTI=0: one launch per small tensor and phase (sum of squares, Adam) instead of the two multi-tensor launches
_MULTI = os.environ.get("MRGCN_MULTI", "1") != "0"
_MULTI_MAX_NUMEL = 1 << 20
# MRGCN_ROW_SPARSE=0 switches the row-sparse weight_I gradient off (A/B runs)
_ROW_SPARSE_DEFAULT = os.environ.get("MRGCN_ROW_SPARSE", "1") != "0"


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


_LABEL_FLAGS: dict = {}


def _label_flags(idx: torch.Tensor, num_rows: int):
    """(flags, unique): uint8 [num_rows] with a 1 at every labelled row, kept under the identity of `idx` — the same
    tensor object, unchanged, gives the same flags tensor every epoch, which is what keys the gradient support of
    the label set (plan.GraphPlan.support_for).  None while a stream capture is under way and the flags do not
    exist yet (their check for repeated rows synchronises)."""
    key = (idx.data_ptr(), idx._version, int(idx.numel()), int(num_rows), idx.device)
    ent = _LABEL_FLAGS.get(key)
    if ent is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        flags = torch.zeros((num_rows,), dtype=torch.uint8, device=idx.device)
        flags[idx] = 1
        unique = int(flags.sum(dtype=torch.int64)) == int(idx.numel())
        while len(_LABEL_FLAGS) >= 8:
            _LABEL_FLAGS.pop(next(iter(_LABEL_FLAGS)))
        ent = _LABEL_FLAGS[key] = (flags, unique, idx)  # (holds `idx`: its address cannot be handed to another tensor)
    return ent[0], ent[1]


class _SoftmaxXent(torch.autograd.Function):
    """nn.CrossEntropyLoss()(Y_hat[idx], targets) (node_classification.py:439-444).  The forward keeps the
    gradient of the labelled rows only (n x C); the backward forms the N x C gradient from it, scaled by the
    upstream gradient in the same pass, and notes which rows hold anything (functional._set_grad_meta: the last
    layer's backward then does not scan 73 MB of zeros for them).  `flags` (from _label_flags): the labelled rows
    as a persistent flags tensor — the note then names a structural row set; `sparse`: the rows outside it are not
    written at all (only for a consumer that reads the flagged rows: functional.rgcn_layer marks such outputs)."""

    @staticmethod
    def forward(ctx, logits, idx, targets, flags=None, sparse=False):
        ctx.flags, ctx.sparse = flags, bool(sparse and flags is not None)
        if logits.stride(1) != 1:        # (rows may be strided: a layer output in a buffer with padded rows)
            logits = logits.contiguous()
        N, C = logits.shape
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        drows = torch.empty((idx.numel(), C), dtype=torch.float32, device=logits.device)
        with torch.cuda.device(logits.device):
            L.check(L.load().mrgcn_softmax_xent_rows_f32(
                logits.data_ptr(), logits.stride(0), C, idx.data_ptr(), targets.data_ptr(), idx.numel(),
                loss.data_ptr(), drows.data_ptr(), _stream(logits.device)), "mrgcn_softmax_xent_rows_f32")
        ctx.save_for_backward(drows, idx)
        ctx.shape = (N, C)
        return loss

    @staticmethod
    def backward(ctx, g):
        from .functional import _set_grad_meta
        drows, idx = ctx.saved_tensors
        N, C = ctx.shape
        dev = drows.device
        g = g.to(torch.float32).contiguous()
        dlogits = torch.empty((N, C), dtype=torch.float32, device=dev)
        if ctx.sparse:  # the labelled rows only: no zero fill of the other N - n
            with torch.cuda.device(dev):
                L.check(L.load().mrgcn_softmax_xent_bwd_rows_f32(
                    drows.data_ptr(), idx.data_ptr(), idx.numel(), C, g.data_ptr(), dlogits.data_ptr(), C,
                    _stream(dev)), "mrgcn_softmax_xent_bwd_rows_f32")
            _set_grad_meta(dlogits, ctx.flags, False, structural=True, sparse_rows=True)
            return dlogits, None, None, None, None
        flags = ctx.flags if ctx.flags is not None else torch.empty((N,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            L.check(L.load().mrgcn_softmax_xent_bwd_f32(
                drows.data_ptr(), idx.data_ptr(), idx.numel(), C, g.data_ptr(), dlogits.data_ptr(), C, N,
                0 if ctx.flags is not None else flags.data_ptr(), _stream(dev)), "mrgcn_softmax_xent_bwd_f32")
        _set_grad_meta(dlogits, flags, False, structural=ctx.flags is not None)
        return dlogits, None, None, None, None


def categorical_crossentropy(Y_hat: torch.Tensor, idx: torch.Tensor, targets: torch.Tensor, sole_consumer: bool = False):
    """`idx`, `targets`: int64 device tensors (the `Y.nonzero()` pair of the reference).  `sole_consumer`: nothing
    but this loss reads `Y_hat` (train_step): when `Y_hat` comes straight out of a layer whose backward goes by the
    row flags, the gradient's unlabelled rows are then not even zero-filled."""
    assert idx.dtype == torch.int64 and targets.dtype == torch.int64
    from .functional import _SUPPORT
    flags = sparse = None
    if _SUPPORT and Y_hat.is_cuda and idx.is_contiguous():
        ent = _label_flags(idx, Y_hat.shape[0])
        if ent is not None:
            flags = ent[0]
            sparse = bool(sole_consumer and ent[1] and getattr(Y_hat, "_mrgcn_sparse_grad_ok", False)
                          and Y_hat.grad_fn is not None and type(Y_hat.grad_fn).__name__ == "_RgcnLayerBackward")
    bump("loss.sparse_rows" if sparse else "loss.flagged" if flags is not None else "loss.plain")
    return _SoftmaxXent.apply(Y_hat, idx.contiguous(), targets.contiguous(), flags, sparse)


def categorical_accuracy(Y_hat, idx, targets):
    """node_classification.py:432-437"""
    labels = Y_hat[idx].argmax(dim=1)
    return (labels == targets).float().mean(), labels, targets


def _to_reference_layout(t):
    """(N, B, F) node-major -> the reference's (B*N, F)."""
    N, B, F = t.shape
    return t.permute(1, 0, 2).reshape(B * N, F)


class ClipAdam(torch.optim.Optimizer):
    """clip_grad_norm_(all params, max_norm) followed by torch.optim.Adam, as two passes of
    HIP kernels: (1) sum of squares of every gradient into one device double, (2) Adam with
    the clip coefficient read from device memory.  Same hyper-parameter names / param-group
    layout as torch.optim.Adam so that `optimizer_params` groups (tasks/utils.py:8-45) work.

    `state_dict()` / `load_state_dict()` speak the reference's layout: the moments of a node-major
    `weight_I` (mrgcn_amd.layers.graph) are handed out and accepted as `(B*N, out)` tensors, so an
    optimizer checkpoint (run.py:230-236) is interchangeable with torch.optim.Adam over the reference
    model.  With `capturable=True` the step counter lives on the device (hipGraph replays advance it);
    `state_dict()` reads it back, `load_state_dict()` seeds it."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 max_norm=1.0, capturable=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.max_norm = max_norm
        # capturable: step counter and bias corrections on the device (one per distinct betas), so
        # that a hipGraph-captured step replays correctly (see GraphedTrainStep)
        self.capturable = capturable
        self._dev_step = {}
        self._scratch = {}
        self._dist = None  # (group, ids of parameters sharded across ranks)
        self._state_gen = 0  # bumped by load_state_dict: row flags built for the old moments are re-derived

    # -- checkpoints --------------------------------------------------------------------------------
    def _sync_host_steps(self):
        """Device step counters (capturable) -> the per-parameter `step` entries.  Synchronises."""
        if not self._dev_step:
            return
        for group in self.param_groups:
            ent = self._dev_step.get(tuple(float(b) for b in group["betas"]))
            if ent is None:
                continue
            t = int(ent[0].item())
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] = t

    def state_dict(self):
        self._sync_host_steps()
        sd = super().state_dict()
        params = [p for g in self.param_groups for p in g["params"]]
        state = {}
        for k, st in sd["state"].items():
            p = params[k]
            if getattr(p, "_mrgcn_node_major", False) and isinstance(st, dict) and "exp_avg" in st:
                st = dict(st)
                for key in ("exp_avg", "exp_avg_sq"):
                    if st[key].dim() == 3:
                        st[key] = _to_reference_layout(st[key])
            state[k] = st
        sd["state"] = state
        return sd

    def load_state_dict(self, state_dict):
        params = [p for g in self.param_groups for p in g["params"]]
        sd = dict(state_dict)
        sd["state"] = dict(sd["state"])
        for k, st in sd["state"].items():
            p = params[k]
            if "exp_avg" not in st:
                continue
            if getattr(p, "_mrgcn_node_major", False) and st["exp_avg"].dim() == 2:
                N, B, F = p.shape
                st = dict(st)
                for key in ("exp_avg", "exp_avg_sq"):
                    st[key] = st[key].view(B, N, F).permute(1, 0, 2).contiguous()
                sd["state"][k] = st
            elif tuple(st["exp_avg"].shape) != tuple(p.shape):
                # same element count in another layout (a reference-shaped moment for a parameter this optimizer
                # does not know to be node-major) would load silently permuted
                raise L.MrgcnError(f"optimizer state {k}: moments of shape {tuple(st['exp_avg'].shape)} for a "
                                   f"parameter of shape {tuple(p.shape)}")
        super().load_state_dict(sd)
        self._state_gen += 1
        self._dev_step = {}  # re-seeded from the loaded `step` entries at the next step

    def set_distributed(self, group, sharded_params):
        """Node-partitioned training (mrgcn_amd.partition): `sharded_params` hold disjoint shards
        per rank (their squared norms add up across ranks); every other parameter is replicated
        and already carries the all-reduced gradient (counted once)."""
        self._dist = (group, {id(p) for p in sharded_params})

    def _dev_scratch(self, device):
        s = self._scratch.get(device)
        if s is None:
            s = dict(accum=torch.zeros((), dtype=torch.float64, device=device),   # (self-cleaning: zero between steps)
                     ticket=torch.zeros((), dtype=torch.int32, device=device),
                     sumsq=torch.zeros((), dtype=torch.float64, device=device),
                     sumsq_sharded=torch.zeros((), dtype=torch.float64, device=device),
                     coef=torch.ones((), dtype=torch.float32, device=device),
                     norm=torch.zeros((), dtype=torch.float32, device=device))
            self._scratch[device] = s
        return s

    def _index_rows_ok(self, p, ent) -> bool:
        """A compact-rows gradient (kind "index") may skip the rows outside its index set only while those rows hold no
        moments: checked once per optimizer state (a loaded state, dense steps in between), with one host read."""
        owner = (id(self), self._state_gen)
        if ent.get("seeded_for") != owner:
            if p.is_cuda and torch.cuda.is_current_stream_capturing():
                raise L.MrgcnError("ClipAdam: the first step with a compact literal gradient looks at the moments "
                                   "(a host read): run one step before capturing")
            st = self.state.get(p)
            ok = p.dim() == 2 and p.is_contiguous() and p.shape[1] % 4 == 0
            if ok and st and int(st.get("step", 0)) > 0:
                outside = torch.ones(p.shape[0], dtype=torch.bool, device=p.device)
                outside[ent["index"]] = False
                ok = not bool(((st["exp_avg"][outside] != 0).any() | (st["exp_avg_sq"][outside] != 0).any()).item())
            ent["dense_only"] = not ok
            ent["seeded_for"] = owner
        return not ent["dense_only"]

    def _new_state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        lib = L.load()
        rowsparse = []  # gradient left on the parameter in row-sparse form (functional._ROW_SPARSE)
        indexed = []    # ... as compact rows of a literal operand (functional._SpmmLiteral: kind "index")
        for g in self.param_groups:
            for p in g["params"]:
                ent = pop_row_grad(p)
                if ent is None:
                    continue
                if ent.get("kind") == "index":
                    if (p.grad is None and float(g["weight_decay"]) == 0.0 and self._dist is None
                            and self._index_rows_ok(p, ent)):
                        indexed.append((g, p, ent))
                    else:
                        merge_row_grad(p, ent)
                elif p.grad is None and float(g["weight_decay"]) == 0.0:
                    rowsparse.append((g, p, ent))
                else:  # another term left a dense gradient on the same parameter (a regulariser): one dense step
                    merge_row_grad(p, ent)
        live = [(g, p) for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not live and not rowsparse and not indexed:
            return None
        device = (live[0][1] if live else (rowsparse or indexed)[0][1]).device
        if not all(p.device == device for _, p in live):
            raise L.MrgcnError("ClipAdam: all parameters must live on one GPU")
        sc = self._dev_scratch(device)
        s = _stream(device)
        use_clip = self.max_norm is not None and self.max_norm > 0
        # (contiguous and 16-byte aligned, as the vector kernels read them: a gradient that is a view into a flat
        # bucket at an odd offset is copied)
        grads = [p.grad if (p.grad.is_contiguous() and p.grad.data_ptr() % 16 == 0) else p.grad.contiguous().clone()
                 for _, p in live]
        # The dense parameters besides the node table are a handful of small tensors: their squared norms, the
        # row-sparse gradients' norms, the clip coefficient and the device step counter take ONE launch
        # (mrgcn_sumsq_clip_multi_f32) and their Adam updates another (mrgcn_adam_step_multi_f32) when every group
        # shares (beta1, beta2, eps) — the reference's groups do (tasks/utils.py:8-45 vary lr / weight_decay only).
        hyper = {(float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])) for g, _ in live} | \
                {(float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])) for g, _, _ in rowsparse + indexed}
        small = [i for i, g in enumerate(grads) if g.numel() <= _MULTI_MAX_NUMEL]
        # (16 tensors per launch: a model with more — an MRGCN with encoders has ~40 — takes a few launches, not 2 x 40)
        multi = (_MULTI and self._dist is None and len(hyper) == 1 and len(small) >= 1 and len(rowsparse) <= 16)
        with torch.cuda.device(device):
            bias = {}
            if self.capturable:
                for group in self.param_groups:  # one device counter per distinct (beta1, beta2)
                    key = tuple(float(b) for b in group["betas"])
                    if key in bias:
                        continue
                    ent = self._dev_step.get(key)
                    if ent is None:
                        # seeded with the steps already taken (a loaded checkpoint, eager steps before)
                        t0 = max([int(self.state[p].get("step", 0)) for g2 in self.param_groups
                                  if tuple(float(b) for b in g2["betas"]) == key for p in g2["params"]
                                  if self.state.get(p)] or [0])
                        ent = (torch.full((), t0, dtype=torch.int64, device=device),
                               torch.ones(2, dtype=torch.float32, device=device))
                        self._dev_step[key] = ent
                    bias[key] = ent[1]
            if multi:
                try:
                    b1m, b2m, _ = next(iter(hyper))
                    for i, g in enumerate(grads):
                        if i not in small:  # (a large dense gradient: its own streaming pass into the same accumulator)
                            L.check(lib.mrgcn_sumsq_accum_f32(g.data_ptr(), g.numel(), sc["accum"].data_ptr(), s),
                                    "mrgcn_sumsq_accum_f32")
                    for c0 in range(0, len(small) - 16, 16) if len(small) > 16 else ():
                        part = small[c0:c0 + 16]
                        L.check(lib.mrgcn_sumsq_accum_multi_f32(
                            len(part), (C.c_void_p * len(part))(*[grads[i].data_ptr() for i in part]),
                            (C.c_int64 * len(part))(*[grads[i].numel() for i in part]), sc["accum"].data_ptr(), s),
                            "mrgcn_sumsq_accum_multi_f32")
                    last = small[(len(small) - 1) // 16 * 16:]   # the launch that also closes the norm
                    closing = [grads[i] for i in last]
                    for _, _, ent in indexed:  # (a compact gradient: in the closing launch while it has room for it)
                        if len(closing) < 16 and ent["g"].numel() <= 4 * _MULTI_MAX_NUMEL:
                            closing.append(ent["g"])
                        else:
                            L.check(lib.mrgcn_sumsq_accum_f32(ent["g"].data_ptr(), ent["g"].numel(),
                                                              sc["accum"].data_ptr(), s), "mrgcn_sumsq_accum_f32")
                    gp = (C.c_void_p * len(closing))(*[g.data_ptr() for g in closing])
                    gn = (C.c_int64 * len(closing))(*[g.numel() for g in closing])
                    ex = (C.c_void_p * max(len(rowsparse), 1))(*[ent["sumsq"].data_ptr() for _, _, ent in rowsparse])
                    dstep = self._dev_step.get((b1m, b2m)) if self.capturable else None
                    L.check(lib.mrgcn_sumsq_clip_multi_f32(
                        len(closing), gp, gn, len(rowsparse), ex, sc["accum"].data_ptr(), sc["ticket"].data_ptr(),
                        float(self.max_norm) if use_clip else 0.0, sc["sumsq"].data_ptr(), sc["coef"].data_ptr(),
                        sc["norm"].data_ptr(), dstep[0].data_ptr() if dstep else 0, b1m, b2m,
                        dstep[1].data_ptr() if dstep else 0, s), "mrgcn_sumsq_clip_multi_f32")
                    for key, bc_t in bias.items():  # groups with other betas (no gradient this step): their counters too
                        if key != (b1m, b2m):
                            L.check(lib.mrgcn_adam_bias_f32(self._dev_step[key][0].data_ptr(), key[0], key[1],
                                                            bc_t.data_ptr(), s), "mrgcn_adam_bias_f32")
                except BaseException:
                    # the scratch words are self-cleaning only when the closing launch ran: a failure in between must not
                    # leak a partial sum into every later norm
                    sc["accum"].zero_()
                    sc["ticket"].zero_()
                    raise
            else:
                sc["sumsq"].zero_()
                sc["sumsq_sharded"].zero_()
                sharded = self._dist[1] if self._dist else ()
                for (_, p), g in zip(live, grads):
                    acc = sc["sumsq_sharded"] if id(p) in sharded else sc["sumsq"]
                    L.check(lib.mrgcn_sumsq_accum_f32(g.data_ptr(), g.numel(), acc.data_ptr(), s),
                            "mrgcn_sumsq_accum_f32")
                for _, p, ent in rowsparse:  # ||g||^2 came for free with the gradient
                    (sc["sumsq_sharded"] if id(p) in sharded else sc["sumsq"]).add_(ent["sumsq"])
                for _, _, ent in indexed:
                    L.check(lib.mrgcn_sumsq_accum_f32(ent["g"].data_ptr(), ent["g"].numel(), sc["sumsq"].data_ptr(), s),
                            "mrgcn_sumsq_accum_f32")
                if self._dist:
                    from .partition import all_reduce_sum_
                    all_reduce_sum_(sc["sumsq_sharded"], self._dist[0])
                sc["sumsq"] += sc["sumsq_sharded"]
                if use_clip:
                    L.check(lib.mrgcn_clip_coef_f32(sc["sumsq"].data_ptr(), float(self.max_norm),
                                                    sc["coef"].data_ptr(), sc["norm"].data_ptr(), s),
                            "mrgcn_clip_coef_f32")
                for key, bc_t in bias.items():
                    L.check(lib.mrgcn_adam_bias_f32(self._dev_step[key][0].data_ptr(), key[0], key[1], bc_t.data_ptr(), s),
                            "mrgcn_adam_bias_f32")
            coef_ptr = step_coef_ptr = sc["coef"].data_ptr() if use_clip else 0
            for group, p, ent in rowsparse:
                if float(group["weight_decay"]) != 0.0:
                    raise L.MrgcnError("row-sparse gradients need weight_decay = 0 (a decayed parameter "
                                       "moves without gradient)")
                st = self._new_state(p)
                owner = (id(self), self._state_gen)
                if ent.get("seeded_for") != owner:
                    # these flags have not seen this optimizer's moments yet (a loaded or dense-built state,
                    # a fresh gradient entry): every node that holds a non-zero moment counts as `ever`
                    ent["ever"].zero_()
                    ent["ever_in"] = None      # which row set the flags lie inside: None = none set yet
                    if st["step"] > 0:   # (a parameter that never took a step has zero moments)
                        nz = (st["exp_avg"] != 0).flatten(1).any(1) | (st["exp_avg_sq"] != 0).flatten(1).any(1)
                        ent["ever"] |= nz.to(torch.uint8)
                        ent["ever_in"] = "any"  # (moments from steps this entry has not seen)
                    ent["seeded_for"] = owner
                st["step"] += 1
                b1, b2 = group["betas"]
                bc = bias[(float(b1), float(b2))].data_ptr() if self.capturable else 0
                nrows = p.shape[0]
                fz = ent.get("fused")
                # the coefficient of a clip that ran between backward and step (mrgcn_amd.optim.clip_grad_norm_)
                pre = ent.pop("coef", None)
                coef_ptr = pre.data_ptr() if pre is not None else step_coef_ptr
                if fz is not None and fz.get("comp_version") is not None and fz["comp"]._version != fz["comp_version"]:
                    raise L.MrgcnError("row-sparse weight_I gradient: weight_I_comp was modified between backward and "
                                       "the node table's update (the fused update re-reads it)")
                if fz is not None and fz.get("sup") is not None:  # the same on the gradient support of the label set
                    # every step since the flags were zeroed ran on THIS support: no node outside it holds moments and
                    # the pass that looks for such nodes is not launched
                    inside = ent.get("ever_in", "any")
                    outside = 0 if (inside is None or inside is fz["sup"]) else 1
                    ent["ever_in"] = fz["sup"] if not outside else "any"
                    bump("adam.list")
                    L.check(lib.mrgcn_support_adam_rows_fused_f32(
                        fz["sup"].handle, fz["dM"].data_ptr(), fz["ld"], fz["comp"].data_ptr(), fz["B"], fz["F"],
                        p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), ent["ever"].data_ptr(),
                        float(group["lr"]), float(b1), float(b2), float(group["eps"]), int(st["step"]), bc, coef_ptr,
                        outside, s), "mrgcn_support_adam_rows_fused_f32")
                    continue
                ent["ever_in"] = "any"
                bump("adam.rows_fused" if fz is not None else "adam.rows")
                if fz is not None:  # no gradient tensor: the blocks are rebuilt from dM inside the Adam pass
                    L.check(lib.mrgcn_adam_step_rows_fused_f32(
                        fz["plan"].handle, fz["dM"].data_ptr(), fz["ld"], fz["live"].data_ptr(), fz["comp"].data_ptr(),
                        fz["B"], fz["F"], p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                        ent["cur"].data_ptr(), ent["ever"].data_ptr(), float(group["lr"]), float(b1), float(b2),
                        float(group["eps"]), int(st["step"]), bc, coef_ptr, s), "mrgcn_adam_step_rows_fused_f32")
                    continue
                L.check(lib.mrgcn_adam_step_rows_f32(
                    p.data_ptr(), ent["g"].data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                    nrows, p.numel() // max(nrows, 1), ent["cur"].data_ptr(), ent["ever"].data_ptr(),
                    float(group["lr"]), float(b1), float(b2), float(group["eps"]), int(st["step"]), bc, coef_ptr, s),
                    "mrgcn_adam_step_rows_f32")
            for group, p, ent in indexed:
                st = self._new_state(p)
                st["step"] += 1
                b1, b2 = group["betas"]
                bc = bias[(float(b1), float(b2))].data_ptr() if self.capturable else 0
                pre = ent.pop("coef", None)
                g = ent["g"]
                bump("adam.index_rows")
                L.check(lib.mrgcn_adam_step_index_rows_f32(
                    p.data_ptr(), g.data_ptr(), g.stride(0), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                    ent["index_ptr"], g.shape[0], p.numel() // max(p.shape[0], 1), float(group["lr"]), float(b1),
                    float(b2), float(group["eps"]), int(st["step"]), bc,
                    pre.data_ptr() if pre is not None else step_coef_ptr, s), "mrgcn_adam_step_index_rows_f32")
            coef_ptr = step_coef_ptr
            for (group, p) in live:
                st = self._new_state(p)
                st["step"] += 1
                rows = getattr(p, "_mrgcn_rows", None)
                if rows is not None and not rows.get("dense_only"):
                    rows["seeded_for"] = None  # a dense step may put moments where the row flags never looked
            # (host-side bias corrections are per step count: the one launch needs the tensors to share it)
            adam_multi = multi and (self.capturable or len({int(self.state[live[i][1]]["step"]) for i in small}) == 1)
            if adam_multi:
                b1m, b2m, epsm = next(iter(hyper))
                for c0 in range(0, len(small), 16):
                    sel = [(live[i][0], live[i][1], grads[i]) for i in small[c0:c0 + 16]]
                    n = len(sel)
                    arr = lambda ptrs: (C.c_void_p * n)(*ptrs)  # noqa: E731
                    L.check(lib.mrgcn_adam_step_multi_f32(
                        n, arr([p.data_ptr() for _, p, _ in sel]), arr([g.data_ptr() for _, _, g in sel]),
                        arr([self.state[p]["exp_avg"].data_ptr() for _, p, _ in sel]),
                        arr([self.state[p]["exp_avg_sq"].data_ptr() for _, p, _ in sel]),
                        (C.c_int64 * n)(*[p.numel() for _, p, _ in sel]),
                        (C.c_float * n)(*[float(g["lr"]) for g, _, _ in sel]),
                        (C.c_float * n)(*[float(g["weight_decay"]) for g, _, _ in sel]), b1m, b2m, epsm,
                        int(self.state[sel[0][1]]["step"]), bias[(b1m, b2m)].data_ptr() if self.capturable else 0,
                        coef_ptr, s), "mrgcn_adam_step_multi_f32")
            for i, ((group, p), g) in enumerate(zip(live, grads)):
                if adam_multi and i in small:
                    continue
                st = self.state[p]
                b1, b2 = group["betas"]
                if self.capturable:
                    L.check(lib.mrgcn_adam_step_dev_f32(
                        p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                        p.numel(), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                        float(group["weight_decay"]), bias[(float(b1), float(b2))].data_ptr(), coef_ptr, s),
                        "mrgcn_adam_step_dev_f32")
                    continue
                L.check(lib.mrgcn_adam_step_f32(
                    p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                    p.numel(), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                    float(group["weight_decay"]), int(st["step"]), coef_ptr, s), "mrgcn_adam_step_f32")
        # the kernels wrote through raw pointers: tell autograd (and every cache keyed by a tensor's version — the gate
        # decisions of models.mrgcn) that these parameters changed, as an in-place torch update would
        torch.autograd.graph.increment_version([p for _, p in live] + [p for _, p, _ in rowsparse + indexed])
        return None

    def last_grad_norm(self) -> float:
        """Total gradient norm of the last step (synchronises)."""
        dev = next(iter(self._scratch))
        return float(self._scratch[dev]["norm"].item())


def merge_row_grad(p, ent):
    """Adds the gradient a row-sparse entry stands for (scaled by the clip coefficient it may carry) to `p.grad`."""
    g = dense_from_rows(p, ent)
    pre = ent.pop("coef", None)
    if pre is not None:
        g = g * pre
    p.grad = g if p.grad is None else p.grad.add_(g)


def weight_regularisation(model, l1_lambda: float = 0.0, l2_lambda: float = 0.0):
    """l1 * sum|p| + l2 * sum p^2 over the parameters whose NAME contains 'weight'
    (node_classification.py:172-188).  Zero in every shipped config; plain tensor ops."""
    reg = None
    for name, p in model.named_parameters():
        if "weight" not in name:
            continue
        term = None
        if l1_lambda > 0:
            term = l1_lambda * p.abs().sum()
        if l2_lambda > 0:
            t2 = l2_lambda * (p * p).sum()
            term = t2 if term is None else term + t2
        if term is not None:
            reg = term if reg is None else reg + term
    return reg


_ONES: dict = {}


def _ones_like_loss(loss):
    """The seed gradient of `loss.backward()`: one cached scalar per (device, dtype) instead of a fill per epoch."""
    key = (loss.device, loss.dtype)
    t = _ONES.get(key)
    if t is None:
        t = _ONES[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    return t


def train_step(model, forward_fn, idx, targets, optimizer, l1_lambda: float = 0.0, l2_lambda: float = 0.0,
               row_sparse=None):
    """One full-batch epoch.  `forward_fn()` returns the logits (e.g. `lambda: model(batch)`).
    Returns the loss as a device scalar (no host sync).  `row_sparse`: None = skip the rows of a node-major
    weight_I's gradient / Adam update that carry no gradient whenever that is exact (ClipAdam, no weight
    decay, no regulariser — `weight_I.grad` stays None for such a step, the gradient travels on the
    parameter); False = always the dense gradient in `.grad` and the dense Adam kernel."""
    params = [p for g in optimizer.param_groups for p in g["params"]]
    clear_row_grads(params)
    logits = forward_fn()
    loss = categorical_crossentropy(logits, idx, targets, sole_consumer=True)
    reg = l1_lambda > 0 or l2_lambda > 0
    if reg:
        loss = loss + weight_regularisation(model, l1_lambda, l2_lambda)
    optimizer.zero_grad(set_to_none=True)
    # weight_I's gradient may stay unwritten where no node has any when the optimizer is the one that
    # knows how to read it and nothing but the loss feeds that gradient (a regulariser adds its own term)
    sparse_ok = (row_sparse is not False and _ROW_SPARSE_DEFAULT and not reg and isinstance(optimizer, ClipAdam)
                 and all(float(g["weight_decay"]) == 0.0 for g in optimizer.param_groups))
    prev = row_sparse_weight_grad(sparse_ok)
    try:
        loss.backward(gradient=_ones_like_loss(loss))
    finally:
        row_sparse_weight_grad(prev)
    optimizer.step()
    return loss.detach()


class GraphedStep:
    """Any allocation-stable, synchronisation-free step `fn()` (returning a device scalar) captured into a hipGraph and
    replayed — e.g. one full-batch link-prediction epoch: negatives drawn on the device from torch's default generator
    (whose state torch advances per replay), encoder, DistMult scores, BCE, backward, `ClipAdam(capturable=True)`.
    `warmup` real calls run first, on the stream the capture then uses (plans and their per-stream scratch exist, lazily
    built caches are filled)."""

    def __init__(self, fn, warmup: int = 3):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            self.out = fn()
        self.warmup_steps = max(warmup, 1)

    def __call__(self):
        self.graph.replay()
        return self.out


class GraphedTrainStep:
    """One full-batch epoch captured into a hipGraph (torch.cuda.CUDAGraph) and replayed: the ~40
    kernel launches of a step become one graph launch, which is what bounds the small shapes
    (AIFB / MUTAG epochs are launch-latency territory).  Everything in the step is stream-ordered
    and allocation-free at the C ABI, the optimizer keeps its step counter on the device
    (`ClipAdam(capturable=True)`), so the captured sequence is exactly the eager one.

        step = GraphedTrainStep(model, lambda: model(X, A), idx, targets, optimizer)
        loss = step()          # device scalar, no host sync

    The graph plans must exist before capture (the warm-up steps build them); shapes are static."""

    def __init__(self, model, forward_fn, idx, targets, optimizer, warmup: int = 3,
                 l1_lambda: float = 0.0, l2_lambda: float = 0.0, row_sparse=None):
        if not getattr(optimizer, "capturable", False):
            raise L.MrgcnError("GraphedTrainStep needs ClipAdam(..., capturable=True)")
        args = (model, forward_fn, idx, targets, optimizer, l1_lambda, l2_lambda, row_sparse)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                train_step(*args)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: calls other threads make while this one captures (e.g. the process group's
        # watchdog in a multi-rank job) do not invalidate the capture.  Captured on the stream the warm-up steps ran
        # on: whatever keeps scratch per stream (a plan's product scratch) has met this stream already.
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            self.loss = train_step(*args)
        # capturing executes nothing on the device: `warmup` optimizer steps have been taken so far (the
        # optimizer's device counter says the same; ClipAdam.state_dict() reads it back)
        self.warmup_steps = max(warmup, 1)
        self._optimizer, self._state_gen = optimizer, optimizer._state_gen

    def __call__(self):
        if self._optimizer._state_gen != self._state_gen:
            # load_state_dict replaced the moment tensors and the device step counter the graph was captured on
            raise L.MrgcnError("GraphedTrainStep: the optimizer's state was loaded after the capture; build a new "
                               "GraphedTrainStep (the captured graph still updates the old moment buffers)")
        self.graph.replay()
        return self.loss
