"""The optimizer half of the reference's own training loop on the fast path.

The reference drives a step as (mrgcn/tasks/node_classification.py:35-37, :190-193; link_prediction.py:325)

    optimizer = optim.Adam(optimizer_params(model, ...), lr=..., weight_decay=...)
    ...
    optimizer.zero_grad(); batch_loss.backward()
    nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    optimizer.step()

With `torch.optim.Adam` / `torch.nn.utils.clip_grad_norm_` that loop works on this package's models as it is
(dense gradients).  The optimizer CHECKPOINT of run.py:232-235 / node_classification.py:73-80 needs one thing more:
the moments of the node-major `weight_I` must travel in the reference's `(B*N, out)` shape.  `install_as_mrgcn()`
therefore always binds `optim.Adam` inside the reference's two task modules to `ReferenceLayoutAdam` (torch's own
Adam with the layout translated in `state_dict()` / `load_state_dict()`), or to `RowSparseAdam` with
`patch_optimizer=True`; an optimizer built elsewhere gets the same through `speak_reference_layout(optimizer)`
(`reference_state_dict` / `load_reference_state_dict` are the one-shot forms).  Beyond the checkpoint, the node table `weight_I` then costs a dense gradient write, a norm pass, a
scaling pass and a dense Adam pass over memory that mostly holds zeros.  `Adam` and `clip_grad_norm_` here are
drop-ins for the two names with the same call signatures: the backward leaves the gradient of a node-major
`weight_I` in row-sparse form (mrgcn_amd.functional), `clip_grad_norm_` folds its squared norm — a by-product of
the backward — into the total norm and hands the coefficient on, `Adam.step()` touches only the node blocks that
have (or ever had) gradient.  Same arithmetic as the dense loop (tests/test_gpu_reference_loop.py: golden vectors
of the reference's own loop).  `mrgcn_amd.install_as_mrgcn(patch_optimizer=True)` puts them in place of `optim.Adam`
and `nn.utils.clip_grad_norm_` inside the reference's task modules.
"""
from __future__ import annotations

import weakref

import torch

from . import _lib as L
from .functional import clear_row_grads
from .train import ClipAdam, _stream, _to_reference_layout, merge_row_grad


class Adam(ClipAdam):
    """`torch.optim.Adam(params, lr, betas, eps, weight_decay)` on HIP kernels; no clipping of its own (the
    reference clips with `nn.utils.clip_grad_norm_` between backward and step).  `state_dict()` /
    `load_state_dict()` speak the reference's layout (ClipAdam).

    `row_sparse` (default False: every gradient dense, `.grad` of every parameter as torch leaves it — any clip,
    scaler or inspection code sees all of it).  True: the optimizer announces itself on the node-major `weight_I`
    parameters it owns and a plain `loss.backward()` then leaves their gradient in ROW-SPARSE form (`weight_I.grad`
    stays None; flags, `dM` and the squared norm travel on the parameter).  Only this module's `clip_grad_norm_`
    knows that form — torch's would skip the node table, i.e. leave its norm out of the total and step it unclipped —
    so the two go together: `RowSparseAdam` + `clip_grad_norm_`, which is what `install_as_mrgcn(patch_optimizer=True)`
    binds inside the reference's task modules."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, *,
                 foreach=None, maximize=False, capturable=False, differentiable=False, fused=None,
                 row_sparse=False):
        if amsgrad or maximize or differentiable:
            raise L.MrgcnError("mrgcn_amd.optim.Adam: amsgrad / maximize / differentiable are not implemented")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_norm=None,
                         capturable=capturable)
        if row_sparse:
            me = weakref.ref(self)
            for g in self.param_groups:
                if float(g["weight_decay"]) != 0.0:
                    continue  # a decayed parameter moves without gradient: its rows cannot be skipped
                for p in g["params"]:
                    if getattr(p, "_mrgcn_node_major", False):
                        p._mrgcn_row_consumer = me

    def zero_grad(self, set_to_none: bool = True):
        clear_row_grads([p for g in self.param_groups for p in g["params"]])
        super().zero_grad(set_to_none=set_to_none)


class RowSparseAdam(Adam):
    """`Adam(row_sparse=True)` under the constructor signature of `torch.optim.Adam`: what the reference's loop gets
    for `optim.Adam` next to this module's `clip_grad_norm_` (mrgcn_amd.patch_task_optimizer)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, **kw):
        kw.setdefault("row_sparse", True)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)


def clip_grad_norm_(parameters, max_norm, norm_type=2.0, error_if_nonfinite=False, foreach=None):
    """`torch.nn.utils.clip_grad_norm_` that also sees gradients in row-sparse form.  The total norm, the
    coefficient `max_norm / (norm + 1e-6)` (clamped to 1) and the scaling stay on the device; the returned norm
    is a 0-dim device tensor like torch's.  Anything this path does not cover (other norm types, no row-sparse
    gradient among the parameters, gradients on the CPU / another GPU / of another dtype) goes to torch's
    implementation, row-sparse entries densified first."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    params = list(parameters)
    rows = [(p, getattr(p, "_mrgcn_rows", None)) for p in params]
    rows = [(p, e) for p, e in rows if e is not None and e["fresh"]]
    if not rows or float(norm_type) != 2.0:
        for p, e in rows:  # (other norm types: densify, then torch)
            e["fresh"] = False
            merge_row_grad(p, e)
        return torch.nn.utils.clip_grad_norm_(params, max_norm, norm_type=norm_type,
                                              error_if_nonfinite=error_if_nonfinite, foreach=foreach)
    for p, e in rows:
        if p.grad is not None:  # a second, dense term on the same parameter: one dense gradient
            e["fresh"] = False
            merge_row_grad(p, e)
    rows = [(p, e) for p, e in rows if e["fresh"]]
    dev = (rows[0][0] if rows else params[0]).device
    dense = [p.grad for p in params if p.grad is not None]
    if dev.type != "cuda" or any(g.device != dev or g.dtype != torch.float32 or g.is_sparse for g in dense) \
            or any(p.device != dev for p, _ in rows):
        # gradients on another device (the reference spreads modules over model.devices) or of another type: the
        # kernels below would read them as float32 pointers of `dev` — densify and let torch do it
        for p, e in rows:
            e["fresh"] = False
            merge_row_grad(p, e)
        return torch.nn.utils.clip_grad_norm_(params, max_norm, norm_type=norm_type,
                                              error_if_nonfinite=error_if_nonfinite, foreach=foreach)
    lib = L.load()
    sumsq = torch.zeros((), dtype=torch.float64, device=dev)
    coef = torch.ones((), dtype=torch.float32, device=dev)
    norm = torch.zeros((), dtype=torch.float32, device=dev)
    s = _stream(dev)
    with torch.cuda.device(dev):
        for g in dense:
            gc = g if g.is_contiguous() else g.contiguous()
            L.check(lib.mrgcn_sumsq_accum_f32(gc.data_ptr(), gc.numel(), sumsq.data_ptr(), s), "mrgcn_sumsq_accum_f32")
        for _, e in rows:
            if e.get("kind") == "index":   # compact rows of a literal operand: the norm of the compact gradient
                L.check(lib.mrgcn_sumsq_accum_f32(e["g"].data_ptr(), e["g"].numel(), sumsq.data_ptr(), s),
                        "mrgcn_sumsq_accum_f32")
            else:
                sumsq.add_(e["sumsq"])
        L.check(lib.mrgcn_clip_coef_f32(sumsq.data_ptr(), float(max_norm), coef.data_ptr(), norm.data_ptr(), s),
                "mrgcn_clip_coef_f32")
    if error_if_nonfinite and not bool(torch.isfinite(norm)):
        raise RuntimeError("The total norm for gradients from `parameters` is non-finite, so it cannot be clipped")
    for g in dense:
        g.mul_(coef)
    for _, e in rows:
        e["coef"] = coef  # applied inside the row-sparse Adam pass
    return norm


# ---- torch.optim.Adam over this package's models: checkpoint layout ---------------------------------------------
def _ref_layout_post_hook(optimizer, state_dict):
    """state_dict post-hook: moments of node-major parameters leave in the reference's `(B*N, out)` shape."""
    idx, _ = _node_major_indices(optimizer)
    if not idx:
        return None
    state = {}
    for k, st in state_dict["state"].items():
        if k in idx:
            st = {key: (_to_reference_layout(v) if torch.is_tensor(v) and v.dim() == 3 else v) for key, v in st.items()}
        state[k] = st
    return dict(state_dict, state=state)


def _ref_layout_load_pre_hook(optimizer, state_dict):
    """load_state_dict pre-hook: reference-shaped moments of node-major parameters are transposed on the way in."""
    idx, params = _node_major_indices(optimizer)
    if not idx:
        return None
    state = {}
    for k, st in state_dict["state"].items():
        if k in idx:
            N, B, F = params[k].shape
            st = {key: (v.view(B, N, F).permute(1, 0, 2).contiguous()
                        if torch.is_tensor(v) and v.dim() == 2 and tuple(v.shape) == (B * N, F) else v)
                  for key, v in st.items()}
        state[k] = st
    return dict(state_dict, state=state)


def speak_reference_layout(optimizer):
    """Makes ANY torch optimizer whose state tensors have the parameter's shape (Adam, AdamW, SGD with momentum, ...)
    save and load its checkpoint in the reference's layout: `optimizer.state_dict()` hands the moments of a node-major
    `weight_I` out as `(B*N, out)` (what `torch.save(optimizer.state_dict())` of run.py:232-235 holds for the reference
    model) and `optimizer.load_state_dict()` accepts them in that shape (node_classification.py:73-80) — instance
    hooks, nothing global.  Idempotent; returns the optimizer.  `ClipAdam` and its subclasses speak it natively."""
    if isinstance(optimizer, ClipAdam) or optimizer.__dict__.get("_mrgcn_reference_layout"):
        return optimizer
    optimizer.register_state_dict_post_hook(_ref_layout_post_hook)
    optimizer.register_load_state_dict_pre_hook(_ref_layout_load_pre_hook)
    optimizer.__dict__["_mrgcn_reference_layout"] = True
    return optimizer


class ReferenceLayoutAdam(torch.optim.Adam):
    """`torch.optim.Adam` itself — torch's arithmetic, dense gradients, every keyword — whose checkpoints are in the
    reference's layout (`speak_reference_layout`).  What `install_as_mrgcn()` binds for `optim.Adam` inside the
    reference's task modules by default, so that `optimizer.load_state_dict(checkpoint['optimizer_state_dict'])`
    (node_classification.py:73-80) takes a checkpoint written by the reference and `optimizer.state_dict()`
    (run.py:230-236) writes one the reference can load."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        speak_reference_layout(self)

    def __setstate__(self, state):   # (unpickling / deepcopy drop instance hooks)
        super().__setstate__(state)
        # load_state_dict ends in __setstate__ too, with the hook dicts intact: register again only when they are gone
        if not self.__dict__.get("_optimizer_state_dict_post_hooks"):
            self.__dict__.pop("_mrgcn_reference_layout", None)
        speak_reference_layout(self)


def _node_major_indices(optimizer):
    params = [p for g in optimizer.param_groups for p in g["params"]]
    return {i for i, p in enumerate(params) if getattr(p, "_mrgcn_node_major", False)}, params


def reference_state_dict(optimizer) -> dict:
    """`optimizer.state_dict()` with the moments of node-major `weight_I` parameters in the reference's `(B*N, out)`
    layout — what `torch.save(optimizer.state_dict())` of run.py:232-235 holds for the reference model.  For
    any optimizer whose state tensors have the parameter's shape (torch.optim.Adam, AdamW, ...)."""
    sd = optimizer.state_dict()
    if isinstance(optimizer, ClipAdam) or optimizer.__dict__.get("_mrgcn_reference_layout"):
        return sd  # already speaks the reference's layout
    idx, _ = _node_major_indices(optimizer)
    state = {}
    for k, st in sd["state"].items():
        if k in idx:
            st = {key: (_to_reference_layout(v) if torch.is_tensor(v) and v.dim() == 3 else v) for key, v in st.items()}
        state[k] = st
    return dict(sd, state=state)


def load_reference_state_dict(optimizer, state_dict) -> None:
    """The inverse: loads an optimizer checkpoint written for the reference model (or by `reference_state_dict`)."""
    if isinstance(optimizer, ClipAdam) or optimizer.__dict__.get("_mrgcn_reference_layout"):
        optimizer.load_state_dict(state_dict)
        return
    idx, params = _node_major_indices(optimizer)
    state = {}
    for k, st in state_dict["state"].items():
        if k in idx:
            N, B, F = params[k].shape
            st = {key: (v.view(B, N, F).permute(1, 0, 2).contiguous()
                        if torch.is_tensor(v) and v.dim() == 2 and tuple(v.shape) == (B * N, F) else v)
                  for key, v in st.items()}
        state[k] = st
    optimizer.load_state_dict(dict(state_dict, state=state))
