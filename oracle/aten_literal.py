"""CPU BASELINE PORT — TEST / BENCH INFRASTRUCTURE ONLY.  Not part of the product path.

The reference's Python cannot travel to the GPU box, so the CPU number reported beside the
MI355X number comes from this *literal port*: the identical ATen op sequence the reference
executes per full-batch epoch, on PyTorch CPU —

  layer    mrgcn/layers/graph.py:62-102   einsum('rb,bij->rij'), A.float(),
                                          torch.mm(sparse_coo, dense), einsum('ij,bjk->bik')
  network  mrgcn/models/rgcn.py:69-89     layer loop + ReLU
  epoch    mrgcn/tasks/node_classification.py:166-193
                                          CrossEntropyLoss on labelled rows, zero_grad,
                                          backward, clip_grad_norm_(…, 1.0), Adam.step

Parity status: PINNED — `tests/test_oracle_golden.py::test_aten_literal_*` checks logits,
loss, gradients and post-Adam parameters against the golden vectors captured from the
reference itself.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.
"""
from __future__ import annotations

import time

import numpy as np
import torch


def make_params(dims, R, N, B, bias, featureless, seed=0, relations=False):
    """Parameter dict with the reference's names/shapes (graph.py:33-57) and Glorot init."""
    g = torch.Generator().manual_seed(seed)
    S = B if B > 0 else R
    params = {}

    def xavier(shape):
        t = torch.empty(shape)
        if len(shape) == 2:
            fan_in, fan_out = shape[1], shape[0]
        else:  # torch's rule for >2-D: receptive field = prod(shape[2:])
            rf = int(np.prod(shape[2:]))
            fan_in, fan_out = shape[1] * rf, shape[0] * rf
        a = float(np.sqrt(6.0 / (fan_in + fan_out)))
        return t.uniform_(-a, a, generator=g).requires_grad_(True)

    for li, (i, o) in enumerate(dims):
        pre = f"layers.layer_{li}."
        input_layer = li == 0
        fl = featureless and input_layer
        if B > 0 and input_layer:
            params[pre + "weight_I_comp"] = xavier((R, B))
        if B > 0 and not fl:
            params[pre + "weight_F_comp"] = xavier((R, B))
        if input_layer:
            params[pre + "weight_I"] = xavier((S * N, o))
        if not fl:
            params[pre + "weight_F"] = xavier((S, i, o))
        if bias:
            params[pre + "b"] = torch.zeros(o, requires_grad=True)
    return params


def layer_forward(p: dict, pre: str, X, A, R, N, B, input_layer, featureless):
    """graph.py:62-102, op for op."""
    has_b = (pre + "b") in p
    AIW_I = 0.0
    if input_layer:
        W_I = p[pre + "weight_I"]
        out = W_I.shape[1]
        if B > 0:
            W_I = W_I.view(B, N, out)
            W_I = torch.einsum("rb,bij->rij", p[pre + "weight_I_comp"], W_I)
            W_I = W_I.view(R * N, out)
        AIW_I = torch.mm(A.float(), W_I)
        if featureless:
            return torch.add(AIW_I, p[pre + "b"]) if has_b else AIW_I
    W_F = p[pre + "weight_F"]
    out = W_F.shape[2]
    if B > 0:
        W_F = torch.einsum("rb,bij->rij", p[pre + "weight_F_comp"], W_F)
    FW_F = torch.einsum("ij,bjk->bik", X, W_F)
    FW_F = torch.reshape(FW_F, (R * N, out))
    AFW_F = torch.mm(A.float(), FW_F)
    AXW = torch.add(AIW_I, AFW_F) if input_layer else AFW_F
    return torch.add(AXW, p[pre + "b"]) if has_b else AXW


def forward(p: dict, n_layers, X, A, R, N, B, featureless, relu_last=False):
    """rgcn.py:69-89 with p_dropout = 0."""
    H = X
    for li in range(n_layers):
        H = layer_forward(p, f"layers.layer_{li}.", H, A, R, N, B, li == 0,
                          featureless and li == 0)
        if li < n_layers - 1 or relu_last:
            H = torch.relu(H)
    return H


class Epoch:
    """Hand-driven epoch of node_classification.py:166-193."""

    def __init__(self, p: dict, n_layers, R, N, B, featureless, lr=0.01, weight_decay=0.0,
                 relu_last=False, l1_lambda=0.0, l2_lambda=0.0):
        self.l1_lambda, self.l2_lambda = l1_lambda, l2_lambda
        self.p, self.n_layers, self.R, self.N, self.B = p, n_layers, R, N, B
        self.featureless, self.relu_last = featureless, relu_last
        self.criterion = torch.nn.CrossEntropyLoss()
        self.opt = torch.optim.Adam(list(p.values()), lr=lr, weight_decay=weight_decay)

    def step(self, X, A, idx, targets):
        Y_hat = forward(self.p, self.n_layers, X, A, self.R, self.N, self.B, self.featureless,
                        self.relu_last)
        loss = self.criterion(Y_hat[idx], targets)
        if self.l1_lambda > 0:  # node_classification.py:172-179 (names containing 'weight')
            loss = loss + self.l1_lambda * sum(v.abs().sum() for k, v in self.p.items() if "weight" in k)
        if self.l2_lambda > 0:  # node_classification.py:181-188
            loss = loss + self.l2_lambda * sum((v ** 2).sum() for k, v in self.p.items() if "weight" in k)
        self.opt.zero_grad()
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(list(self.p.values()), 1.0)
        self.opt.step()
        return Y_hat, loss, norm


def coo_tensor(rows, cols, vals, shape):
    """The uncoalesced COO the reference builds (data/utils.py:165-170); `vals` int8 or f32."""
    idx = torch.from_numpy(np.stack([rows, cols]).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(np.asarray(vals)), shape)


def time_epochs(dims, R, N, B, rows, cols, vals, X, idx, targets, featureless, warmup=1, steps=2,
                threads=None, seed=0, per_epoch=False):
    """Times `steps` epochs (after `warmup`) of the literal ATen path; returns
    (ms per epoch, threads used) — with `per_epoch` the list of every timed epoch's ms instead of their mean
    (BASELINE.md's protocol reports median and min)."""
    if threads:
        torch.set_num_threads(threads)
    A = coo_tensor(rows, cols, vals, (N, R * N))
    p = make_params(dims, R, N, B, False, featureless, seed)
    ep = Epoch(p, len(dims), R, N, B, featureless)
    Xt = None if X is None else torch.from_numpy(X)
    it = torch.from_numpy(idx)
    tt = torch.from_numpy(targets)
    for _ in range(warmup):
        ep.step(Xt, A, it, tt)
    if per_epoch:
        each = []
        for _ in range(steps):
            t0 = time.perf_counter()
            ep.step(Xt, A, it, tt)
            each.append((time.perf_counter() - t0) * 1e3)
        return each, torch.get_num_threads()
    t0 = time.perf_counter()
    for _ in range(steps):
        ep.step(Xt, A, it, tt)
    dt = (time.perf_counter() - t0) / steps
    return dt * 1e3, torch.get_num_threads()
