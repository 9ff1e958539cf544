"""Full-batch epoch driver: the MI355X counterpart of the train step of
mrgcn/tasks/node_classification.py:154-193 —

    Y_hat = model(batch); loss = CE(Y_hat[idx], targets)
    zero_grad; backward; clip_grad_norm_(params, 1.0); Adam.step

with the loss, the global gradient norm, the clip and Adam running as HIP kernels
(csrc/optim.hip).  The clip coefficient never leaves the device, so one epoch has no
host synchronisation."""
from __future__ import annotations

import torch

from . import _lib as L
from .functional import (clear_grad_sumsq, defer_input_grad, pop_deferred, pop_grad_sumsq, pop_nodemajor,
                         pop_weight_chunks, sparse_weight_grad)


import os

# MRGCN_SPARSE_WGRAD=0 switches the chunk-sparse weight_I gradient off (A/B runs)
_SPARSE_WGRAD_DEFAULT = os.environ.get("MRGCN_SPARSE_WGRAD", "1") != "0"


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class _SoftmaxXent(torch.autograd.Function):
    """nn.CrossEntropyLoss()(Y_hat[idx], targets) (node_classification.py:439-444) with the
    gradient produced in the same pass."""

    @staticmethod
    def forward(ctx, logits, idx, targets):
        logits = logits.contiguous()
        N, C = logits.shape
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits)
        with torch.cuda.device(logits.device):
            L.check(L.load().mrgcn_softmax_xent_f32(
                logits.data_ptr(), logits.stride(0), C, idx.data_ptr(), targets.data_ptr(),
                idx.numel(), loss.data_ptr(), dlogits.data_ptr(), dlogits.stride(0), N,
                _stream(logits.device)), "mrgcn_softmax_xent_f32")
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None, None


def categorical_crossentropy(Y_hat: torch.Tensor, idx: torch.Tensor, targets: torch.Tensor):
    """`idx`, `targets`: int64 device tensors (the `Y.nonzero()` pair of the reference)."""
    assert idx.dtype == torch.int64 and targets.dtype == torch.int64
    return _SoftmaxXent.apply(Y_hat, idx.contiguous(), targets.contiguous())


def categorical_accuracy(Y_hat, idx, targets):
    """node_classification.py:432-437"""
    labels = Y_hat[idx].argmax(dim=1)
    return (labels == targets).float().mean(), labels, targets


class ClipAdam(torch.optim.Optimizer):
    """clip_grad_norm_(all params, max_norm) followed by torch.optim.Adam, as two passes of
    HIP kernels: (1) sum of squares of every gradient into one device double, (2) Adam with
    the clip coefficient read from device memory.  Same hyper-parameter names / param-group
    layout as torch.optim.Adam so that `optimizer_params` groups (tasks/utils.py:8-45) work."""

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
        self._state_gen = 0  # bumped by load_state_dict: masks built for the old moments are re-derived

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._state_gen = getattr(self, "_state_gen", 0) + 1

    def state_dict(self):
        """torch.optim.Adam's format: moments kept node-major internally are handed out in the
        parameter's own layout."""
        sd = super().state_dict()
        params = [p for g in self.param_groups for p in g["params"]]
        state = {}
        for k, st in sd["state"].items():
            if isinstance(st, dict) and st.get("node_major"):
                p = params[k]
                st = {kk: vv for kk, vv in st.items() if kk != "node_major"}
                for key in ("exp_avg", "exp_avg_sq"):
                    st[key] = st[key].permute(1, 0, 2).contiguous().view_as(p)
            state[k] = st
        sd["state"] = state
        return sd

    def set_distributed(self, group, sharded_params):
        """Node-partitioned training (mrgcn_amd.partition): `sharded_params` hold disjoint shards
        per rank (their squared norms add up across ranks); every other parameter is replicated
        and already carries the all-reduced gradient (counted once)."""
        self._dist = (group, {id(p) for p in sharded_params})

    def _dev_scratch(self, device):
        s = self._scratch.get(device)
        if s is None:
            s = dict(sumsq=torch.zeros((), dtype=torch.float64, device=device),
                     sumsq_sharded=torch.zeros((), dtype=torch.float64, device=device),
                     coef=torch.ones((), dtype=torch.float32, device=device),
                     norm=torch.zeros((), dtype=torch.float32, device=device))
            self._scratch[device] = s
        return s

    @torch.no_grad()
    def step(self, closure=None):
        lib = L.load()
        live = [(g, p) for g in self.param_groups for p in g["params"] if p.grad is not None]
        # parameters whose gradient was left in recomputable form (functional.defer_input_grad)
        deferred = []
        nodemajor = []  # gradient in node-major form (functional._NODEMAJOR), moments kept the same way
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    ent = pop_deferred(p)
                    if ent is not None:
                        deferred.append((g, p, ent))
                        continue
                    ent = pop_nodemajor(p)
                    if ent is not None:
                        nodemajor.append((g, p, ent))
        if not live and not deferred and not nodemajor:
            return None
        device = (live[0][1] if live else (deferred or nodemajor)[0][1]).device
        if not all(p.device == device for _, p in live):
            raise L.MrgcnError("ClipAdam: all parameters must live on one GPU")
        sc = self._dev_scratch(device)
        s = _stream(device)
        with torch.cuda.device(device):
            sc["sumsq"].zero_()
            sc["sumsq_sharded"].zero_()
            sharded = self._dist[1] if self._dist else ()
            grads = []
            for _, p in live:
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                grads.append(g)
                acc = sc["sumsq_sharded"] if id(p) in sharded else sc["sumsq"]
                pre = pop_grad_sumsq(g)  # already accumulated by the kernel that produced g?
                if pre is not None:
                    acc += pre
                else:
                    L.check(lib.mrgcn_sumsq_accum_f32(g.data_ptr(), g.numel(), acc.data_ptr(), s),
                            "mrgcn_sumsq_accum_f32")
            for _, p, ent in deferred + nodemajor:
                (sc["sumsq_sharded"] if id(p) in sharded else sc["sumsq"]).add_(ent["sumsq"])
            clear_grad_sumsq()
            if self._dist:
                from .partition import all_reduce_sum_
                all_reduce_sum_(sc["sumsq_sharded"], self._dist[0])
            sc["sumsq"] += sc["sumsq_sharded"]
            use_clip = self.max_norm is not None and self.max_norm > 0
            if use_clip:
                L.check(lib.mrgcn_clip_coef_f32(sc["sumsq"].data_ptr(), float(self.max_norm),
                                                sc["coef"].data_ptr(), sc["norm"].data_ptr(), s),
                        "mrgcn_clip_coef_f32")
            # deferred parameters first: their pass 2 needs the other parameters' pre-step values
            for group, p, ent in deferred:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                b1, b2 = group["betas"]
                L.check(lib.mrgcn_basis_mix_bwd_adam_f32(
                    ent["plan"].handle, ent["dM"].data_ptr(), ent["ld"], ent["comp"].data_ptr(), ent["B"],
                    ent["F"], p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                    float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                    float(group["weight_decay"]), int(st["step"]),
                    sc["coef"].data_ptr() if use_clip else 0, s), "mrgcn_basis_mix_bwd_adam_f32")
            bias = {}
            if self.capturable:
                if deferred:
                    raise L.MrgcnError("ClipAdam(capturable=True) does not take deferred gradients")
                for group in self.param_groups:  # one device counter per distinct (beta1, beta2)
                    key = tuple(float(b) for b in group["betas"])
                    if key in bias:
                        continue
                    ent = self._dev_step.get(key)
                    if ent is None:
                        ent = (torch.zeros((), dtype=torch.int64, device=device),
                               torch.ones(2, dtype=torch.float32, device=device))
                        self._dev_step[key] = ent
                    L.check(lib.mrgcn_adam_bias_f32(ent[0].data_ptr(), key[0], key[1], ent[1].data_ptr(), s),
                            "mrgcn_adam_bias_f32")
                    bias[key] = ent[1]
            for group, p, ent in nodemajor:
                if float(group["weight_decay"]) != 0.0:
                    raise L.MrgcnError("node-major gradients need weight_decay = 0 (a decayed parameter "
                                       "moves without gradient)")
                N_, Bn, Fn_ = ent["shape"]
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros((N_, Bn, Fn_), dtype=torch.float32, device=p.device)
                    st["exp_avg_sq"] = torch.zeros((N_, Bn, Fn_), dtype=torch.float32, device=p.device)
                    st["node_major"] = (N_, Bn, Fn_)
                elif not st.get("node_major"):
                    # moments built in the parameter's layout (plain steps, a loaded checkpoint): transpose
                    # once, and every node that holds a non-zero moment counts as `ever`
                    for key in ("exp_avg", "exp_avg_sq"):
                        st[key] = st[key].reshape(Bn, N_, Fn_).permute(1, 0, 2).contiguous()
                    st["node_major"] = (N_, Bn, Fn_)
                    ent["seeded_for"] = None
                owner = (id(self), getattr(self, "_state_gen", 0))
                if ent.get("seeded_for") != owner:
                    # these flags have not seen this optimizer's moments yet (a converted or re-loaded
                    # state, a rebuilt gradient entry): every node with a non-zero moment counts as `ever`
                    if st["step"] > 0:
                        nz = (st["exp_avg"] != 0).flatten(1).any(1) | (st["exp_avg_sq"] != 0).flatten(1).any(1)
                        ent["ever"] |= nz.to(torch.uint8)
                    ent["seeded_for"] = owner
                st["step"] += 1
                b1, b2 = group["betas"]
                bc = bias[(float(b1), float(b2))].data_ptr() if self.capturable else 0
                L.check(lib.mrgcn_adam_step_nodemajor_f32(
                    p.data_ptr(), ent["g"].data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                    N_, Bn, Fn_, ent["cur"].data_ptr(), ent["ever"].data_ptr(), float(group["lr"]), float(b1),
                    float(b2), float(group["eps"]), int(st["step"]), bc,
                    sc["coef"].data_ptr() if use_clip else 0, s), "mrgcn_adam_step_nodemajor_f32")
            for (group, p), g in zip(live, grads):
                st = self.state[p]
                had_state = bool(st)
                if st.get("node_major"):  # back on the plain path: moments return to the parameter's layout
                    for key in ("exp_avg", "exp_avg_sq"):
                        st[key] = st[key].permute(1, 0, 2).contiguous().view_as(p)
                    del st["node_major"]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                b1, b2 = group["betas"]
                chunks = pop_weight_chunks(p)
                if chunks is not None:
                    # gradient written only where nodes with gradient live (functional.sparse_weight_grad)
                    if float(group["weight_decay"]) != 0.0:
                        raise L.MrgcnError("chunk-sparse gradients need weight_decay = 0 (a decayed "
                                           "parameter moves without gradient)")
                    owner = (id(self), getattr(self, "_state_gen", 0))  # this optimizer, this (possibly re-loaded) state
                    if chunks.get("synced_for") != owner:
                        chunks["state_synced"] = False
                        chunks["synced_for"] = owner
                    if had_state and not chunks.get("state_synced"):
                        # moments that were not built under these masks (a loaded checkpoint, steps taken
                        # on the plain path): every chunk that holds a non-zero moment counts as `ever`
                        nzm = (st["exp_avg"].view(chunks["B"], -1) != 0).any(0)
                        nzm |= (st["exp_avg_sq"].view(chunks["B"], -1) != 0).any(0)
                        pad = chunks["ever"].numel() * 1024 - nzm.numel()
                        nzm = torch.nn.functional.pad(nzm, (0, pad)).view(-1, 1024).any(1)
                        chunks["ever"] |= nzm.to(torch.uint8)
                    chunks["state_synced"] = True
                    bc = bias[(float(b1), float(b2))].data_ptr() if self.capturable else 0
                    L.check(lib.mrgcn_adam_step_chunked_f32(
                        p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                        chunks["slab"], chunks["B"], chunks["cur"].data_ptr(), chunks["ever"].data_ptr(),
                        float(group["lr"]), float(b1), float(b2), float(group["eps"]), int(st["step"]), bc,
                        sc["coef"].data_ptr() if use_clip else 0, s), "mrgcn_adam_step_chunked_f32")
                    continue
                if self.capturable:
                    L.check(lib.mrgcn_adam_step_dev_f32(
                        p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                        p.numel(), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                        float(group["weight_decay"]), bias[(float(b1), float(b2))].data_ptr(),
                        sc["coef"].data_ptr() if use_clip else 0, s), "mrgcn_adam_step_dev_f32")
                    continue
                L.check(lib.mrgcn_adam_step_f32(
                    p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                    p.numel(), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                    float(group["weight_decay"]), int(st["step"]),
                    sc["coef"].data_ptr() if use_clip else 0, s), "mrgcn_adam_step_f32")
        return None

    def last_grad_norm(self) -> float:
        """Total gradient norm of the last step (synchronises)."""
        dev = next(iter(self._scratch))
        return float(self._scratch[dev]["norm"].item())


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


def train_step(model, forward_fn, idx, targets, optimizer, l1_lambda: float = 0.0, l2_lambda: float = 0.0,
               row_sparse=None):
    """One full-batch epoch.  `forward_fn()` returns the logits (e.g. `lambda: model(batch)`).
    Returns the loss as a device scalar (no host sync).  `row_sparse`: None = skip the rows of
    weight_I's gradient / Adam update that carry no gradient whenever that is exact (ClipAdam, no
    weight decay, no regulariser); False = always the dense gradient and the dense Adam kernel."""
    clear_grad_sumsq()
    logits = forward_fn()
    loss = categorical_crossentropy(logits, idx, targets)
    reg = l1_lambda > 0 or l2_lambda > 0
    if reg:
        loss = loss + weight_regularisation(model, l1_lambda, l2_lambda)
    optimizer.zero_grad(set_to_none=True)
    # a regulariser adds its own term to weight_I's gradient: the deferred (recomputed) form
    # cannot represent that, so it is off for such steps
    prev = defer_input_grad(False) if reg else None
    # weight_I's gradient may stay unwritten where no node has any (chunk-sparse) when the optimizer
    # is the one that knows how to read it and nothing but the loss feeds that gradient
    sparse_ok = (row_sparse is not False and _SPARSE_WGRAD_DEFAULT and not reg and isinstance(optimizer, ClipAdam)
                 and all(float(g["weight_decay"]) == 0.0 for g in optimizer.param_groups))
    prev_sparse = sparse_weight_grad(sparse_ok)
    try:
        loss.backward()
    finally:
        sparse_weight_grad(prev_sparse)
        if reg:
            defer_input_grad(prev)
    optimizer.step()
    return loss.detach()


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
        # watchdog in a multi-rank job) do not invalidate the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = train_step(*args)
        self.warmup_steps = max(warmup, 1) + 1  # optimizer steps already taken (capture runs one)

    def __call__(self):
        self.graph.replay()
        return self.loss
