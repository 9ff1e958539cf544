"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A numpy/scipy restatement of the reference's full-batch R-GCN path, written from
the reference's behaviour (file:line relative to /root/reference).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package; `mrgcn_amd/` never does (the product path fails loudly when the HIP
library is missing).

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here
against the `.npz` fixtures under `tests/golden/`, which were produced by importing
the reference itself in the authoring container (`tests/golden/make_goldens.py`).

All arithmetic is done in float64 unless `dtype=np.float32` is requested, so the
oracle is the *more* accurate side of every comparison.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


# ---------------------------------------------------------------------------
# a-1  layout contract of A          mrgcn/encodings/graph_structure.py:13-38,
#                                    :70-108, :162-169 ; tarball.py:151-157
# ---------------------------------------------------------------------------
def normalize_adjacency(adj: sp.csr_matrix) -> sp.csr_matrix:
    """Row normalisation D^-1 A (graph_structure.py:162-169)."""
    d = np.asarray(adj.sum(1)).flatten().astype(np.float64)
    with np.errstate(divide="ignore"):
        d_inv = 1.0 / d
    d_inv[np.isinf(d_inv)] = 0.0
    return sp.diags(d_inv).dot(adj).tocsr()


def build_stacked_adjacency(triples: np.ndarray, num_nodes: int, num_pred: int,
                            include_inverse: bool = True) -> sp.csr_matrix:
    """Integer triples (s, p, o) -> N x (R*N) CSR, float32 values.

    Block order [p0, p0^-1, p1, p1^-1, ..., identity] (graph_structure.py:78-106,
    :33-38); column of relation block r, source node j is r*N + j."""
    shape = (num_nodes, num_nodes)
    blocks = []
    for p in range(num_pred):
        e = triples[triples[:, 1] == p]
        row, col = e[:, 0], e[:, 2]
        data = np.ones(len(row), dtype=np.int8)
        blocks.append(normalize_adjacency(
            sp.csr_matrix((data, (row, col)), shape=shape, dtype=np.int8)))
        if include_inverse:
            blocks.append(normalize_adjacency(
                sp.csr_matrix((data, (col, row)), shape=shape, dtype=np.int8)))
    blocks.append(normalize_adjacency(sp.identity(num_nodes).tocsr()))
    A = sp.hstack(blocks, format="csr")
    return sp.csr_matrix((A.data.astype(np.float32), A.indices, A.indptr),
                         shape=A.shape, dtype=np.float32)


# ---------------------------------------------------------------------------
# a-2  CSR -> COO int8               mrgcn/data/utils.py:165-170,
#                                    mrgcn/data/batch.py:144-149
# ---------------------------------------------------------------------------
def csr_to_coo(A: sp.csr_matrix, value_mode: str = "ref_int8"):
    """Returns (indices int64 2 x nnz, values).  `ref_int8` reproduces the cast of
    the row-normalised floats to int8 (truncation toward zero: only 1.0 survives)."""
    indices = np.array(A.nonzero()).astype(np.int64)
    if value_mode == "ref_int8":
        values = A.data.astype(np.float32).astype(np.int8)
    elif value_mode == "norm_f32":
        values = A.data.astype(np.float32)
    else:
        raise ValueError(value_mode)
    if indices.shape[1] != values.shape[0]:
        raise ValueError("explicit zeros stored in A (SURVEY Appendix A-2)")
    return indices, values


def coo_to_csr(indices: np.ndarray, values: np.ndarray, shape, dtype=np.float64):
    return sp.csr_matrix((values.astype(dtype), (indices[0], indices[1])), shape=shape)


# ---------------------------------------------------------------------------
# a-4 .. a-8  GraphConvolution.forward    mrgcn/layers/graph.py:62-102
# ---------------------------------------------------------------------------
class LayerCfg:
    def __init__(self, indim, outdim, num_relations, num_nodes, num_bases=-1,
                 bias=False, input_layer=False, featureless=False):
        self.indim, self.outdim = indim, outdim
        self.R, self.N, self.B = num_relations, num_nodes, num_bases
        self.bias, self.input_layer, self.featureless = bias, input_layer, featureless


def layer_forward(cfg: LayerCfg, p: dict, X, A: sp.csr_matrix, dtype=np.float64):
    """Returns (Y, cache).  `p` holds weight_I / weight_F / weight_I_comp /
    weight_F_comp / b as numpy arrays with the reference's shapes (graph.py:33-57)."""
    R, N, B, out = cfg.R, cfg.N, cfg.B, cfg.outdim
    cache = {}
    Y = 0.0
    if cfg.input_layer:
        W_I = p["weight_I"].astype(dtype)
        if B > 0:  # graph.py:69-72  einsum('rb,bij->rij')
            V = W_I.reshape(B, N, out)
            W_I = np.einsum("rb,bij->rij", p["weight_I_comp"].astype(dtype), V
                            ).reshape(R * N, out)
        Y = A @ W_I  # graph.py:75
        if cfg.featureless:
            if cfg.bias:
                Y = Y + p["b"].astype(dtype)
            return Y, cache
    W_F = p["weight_F"].astype(dtype)
    if B > 0:  # graph.py:83-85
        W_F = np.einsum("rb,bij->rij", p["weight_F_comp"].astype(dtype), W_F)
    Xd = X.astype(dtype)
    FW = np.einsum("ij,bjk->bik", Xd, W_F).reshape(R * Xd.shape[0], out)  # graph.py:93-94
    AFW = A @ FW  # graph.py:95
    Y = Y + AFW if cfg.input_layer else AFW
    if cfg.bias:
        Y = Y + p["b"].astype(dtype)
    cache["W_F"] = W_F
    return Y, cache


def layer_backward(cfg: LayerCfg, p: dict, X, A: sp.csr_matrix, dY, cache,
                   dtype=np.float64):
    """Analytic gradients of `layer_forward` (what autograd does for graph.py:62-102):
    returns (grads dict, dX or None)."""
    R, N, B, out = cfg.R, cfg.N, cfg.B, cfg.outdim
    g = {}
    dD = (A.T @ dY).reshape(R, -1, out)  # dense (R, N, out)
    if cfg.bias:
        g["b"] = dY.sum(0)
    if cfg.input_layer:
        if B > 0:
            V = p["weight_I"].astype(dtype).reshape(B, N, out)
            comp = p["weight_I_comp"].astype(dtype)
            g["weight_I"] = np.einsum("rb,rij->bij", comp, dD).reshape(B * N, out)
            g["weight_I_comp"] = np.einsum("rij,bij->rb", dD, V)
        else:
            g["weight_I"] = dD.reshape(R * N, out)
        if cfg.featureless:
            return g, None
    W_F = cache["W_F"]
    Xd = X.astype(dtype)
    dX = np.einsum("rjo,rio->ji", dD, W_F)
    dW = np.einsum("ji,rjo->rio", Xd, dD)
    if B > 0:
        g["weight_F"] = np.einsum("rb,rio->bio", p["weight_F_comp"].astype(dtype), dW)
        g["weight_F_comp"] = np.einsum("rio,bio->rb", dW, p["weight_F"].astype(dtype))
    else:
        g["weight_F"] = dW
    return g, dX


# ---------------------------------------------------------------------------
# a-10  RGCN._forward_full_batch          mrgcn/models/rgcn.py:69-89
# ---------------------------------------------------------------------------
def rgcn_cfgs(dims, R, N, B, bias, featureless):
    """Layer 0 is always the input layer (rgcn.py:30-37); later layers never are
    (rgcn.py:41-51)."""
    cfgs = []
    for li, (i, o) in enumerate(dims):
        cfgs.append(LayerCfg(i, o, R, N, B, bias, input_layer=(li == 0),
                             featureless=(featureless if li == 0 else False)))
    return cfgs


def split_params(state: dict, num_layers: int, prefix="layers.layer_"):
    out = []
    for li in range(num_layers):
        pre = f"{prefix}{li}."
        out.append({k[len(pre):]: v for k, v in state.items() if k.startswith(pre)})
    return out


def rgcn_forward(cfgs, params, X, A, relu_last=False, dtype=np.float64):
    """p_dropout = 0 (every shipped config); ReLU on all but the last layer
    (node_classification.py:399-419), on every layer for link prediction
    (link_prediction.py:449-464)."""
    H = X
    tape = []
    for li, (cfg, p) in enumerate(zip(cfgs, params)):
        pre, cache = layer_forward(cfg, p, H, A, dtype)
        act = (li < len(cfgs) - 1) or relu_last
        post = np.maximum(pre, 0.0) if act else pre
        tape.append((H, pre, cache, act))
        H = post
    return H, tape


def cross_entropy(logits, idx, targets):
    """nn.CrossEntropyLoss (mean) over labelled rows
    (node_classification.py:439-444).  Returns (loss, dlogits)."""
    Z = logits[idx]
    Z = Z - Z.max(1, keepdims=True)
    lse = np.log(np.exp(Z).sum(1, keepdims=True))
    logp = Z - lse
    n = len(idx)
    loss = -logp[np.arange(n), targets].mean()
    dZ = np.exp(logp)
    dZ[np.arange(n), targets] -= 1.0
    dZ /= n
    dlogits = np.zeros_like(logits)
    np.add.at(dlogits, idx, dZ)
    return loss, dlogits


def rgcn_backward(cfgs, params, A, tape, dOut, dtype=np.float64):
    grads = [None] * len(cfgs)
    dH = dOut
    for li in reversed(range(len(cfgs))):
        H_in, pre, cache, act = tape[li]
        dPre = dH * (pre > 0) if act else dH
        g, dX = layer_backward(cfgs[li], params[li], H_in, A, dPre, cache, dtype)
        grads[li] = g
        dH = dX
    return grads, dH


def accuracy(logits, idx, targets):
    """node_classification.py:432-437"""
    return float((logits[idx].argmax(1) == targets).mean())


# ---------------------------------------------------------------------------
# a-12  clip_grad_norm_(…, 1.0) + Adam      node_classification.py:190-193, :35-37
# ---------------------------------------------------------------------------
def clip_grad_norm(grads_flat: list, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_: total L2 norm over all grads, scale by
    min(1, max_norm / (norm + 1e-6)).  Returns (total_norm, coef)."""
    total = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads_flat))
    coef = min(1.0, max_norm / (total + 1e-6))
    return total, coef


class Adam:
    """torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, amsgrad False;
    L2 weight decay added to the gradient."""

    def __init__(self, lr=0.01, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, params: dict, grads: dict):
        self.t += 1
        b1, b2 = self.betas
        for k, g in grads.items():
            p = params[k].astype(np.float64)
            g = g.astype(np.float64)
            if self.wd:
                g = g + self.wd * p
            m = self.m.get(k, 0.0) * b1 + (1 - b1) * g
            v = self.v.get(k, 0.0) * b2 + (1 - b2) * g * g
            self.m[k], self.v[k] = m, v
            mhat = m / (1 - b1 ** self.t)
            denom = np.sqrt(v) / np.sqrt(1 - b2 ** self.t) + self.eps
            params[k] = p - self.lr * mhat / denom


def train_steps(dims, R, N, B, bias, featureless, state, X, A, idx, targets,
                n_steps, relu_last=False, lr=0.01, weight_decay=0.0):
    """Hand-driven epoch loop of node_classification.py:166-193 (zero_grad,
    backward, clip 1.0, Adam) over an `RGCN` state dict with keys
    `layers.layer_<i>.<name>`.  Returns a list of per-step records."""
    cfgs = rgcn_cfgs(dims, R, N, B, bias, featureless)
    state = {k: v.astype(np.float64) for k, v in state.items()}
    opt = Adam(lr=lr, weight_decay=weight_decay)
    records = []
    for _ in range(n_steps):
        params = split_params(state, len(cfgs))
        logits, tape = rgcn_forward(cfgs, params, X, A, relu_last)
        loss, dlogits = cross_entropy(logits, idx, targets)
        grads, dX = rgcn_backward(cfgs, params, A, tape, dlogits)
        flat = {f"layers.layer_{li}.{k}": v for li, g in enumerate(grads) for k, v in g.items()}
        total, coef = clip_grad_norm(list(flat.values()), 1.0)
        clipped = {k: v * coef for k, v in flat.items()}
        rec = dict(logits=logits, loss=loss, grads=flat, dX=dX, grad_norm=total, tape=tape)
        opt.step(state, clipped)
        rec["state"] = {k: v.copy() for k, v in state.items()}
        records.append(rec)
    return records


# ---------------------------------------------------------------------------
# stand-alone SpMM (K3 / K6 / K9)            graph.py:75, :95 and its autograd
# ---------------------------------------------------------------------------
def spmm(indptr, indices, values, D, num_rows, dtype=np.float64):
    A = sp.csr_matrix((values.astype(dtype), indices, indptr),
                      shape=(num_rows, D.shape[0]))
    return A @ D.astype(dtype)


def spmm_t(indptr, indices, values, dY, num_cols, dtype=np.float64):
    A = sp.csr_matrix((values.astype(dtype), indices, indptr),
                      shape=(dY.shape[0], num_cols))
    return A.T @ dY.astype(dtype)


# ---------------------------------------------------------------------------
# the same forward, evaluated for a handful of output rows only
# (graph.py:62-102 + rgcn.py:69-89 restricted to the rows' receptive field)
# ---------------------------------------------------------------------------
def rgcn_forward_at_rows(cfgs, params, X, A: sp.csr_matrix, rows, relu_last=False,
                         chunk=65536, dtype=np.float64):
    """Activations of the LAST layer at `rows` (sorted unique node ids), float64, without
    ever forming an (R*N) x out operand: layer l at a set of rows needs layer l-1 at the
    source nodes of those rows only, so a full-size graph (AM: 1.67 M nodes) is evaluated
    for a few hundred rows in seconds.  Pinned against `rgcn_forward` by
    tests/test_oracle_golden.py.  `params[l]` hold the reference's shapes."""
    rows = np.asarray(rows, dtype=np.int64)
    A = A.tocsr()

    def layer_at(li, rws):
        cfg, p = cfgs[li], params[li]
        R, N, B, out = cfg.R, cfg.N, cfg.B, cfg.outdim
        sub = A[rws, :].tocoo()
        ucol, inv = np.unique(sub.col.astype(np.int64), return_inverse=True)
        r_u, j_u = ucol // N, ucol % N
        D = np.zeros((len(ucol), out), dtype=dtype)
        if cfg.input_layer:
            W_I = p["weight_I"]
            if B > 0:  # graph.py:69-72 for the touched columns only
                V = W_I.reshape(B, N, out)
                comp = p["weight_I_comp"].astype(dtype)
                for s in range(0, len(ucol), chunk):
                    e = slice(s, s + chunk)
                    D[e] = np.einsum("nb,bnf->nf", comp[r_u[e]], V[:, j_u[e], :].astype(dtype))
            else:
                D += W_I[ucol].astype(dtype)
        if not (cfg.input_layer and cfg.featureless):
            W_F = p["weight_F"].astype(dtype)
            if B > 0:  # graph.py:83-85
                W_F = np.einsum("rb,bij->rij", p["weight_F_comp"].astype(dtype), W_F)
            src = np.unique(j_u)
            H_src = X[src].astype(dtype) if li == 0 else layer_at(li - 1, src)
            pos = np.searchsorted(src, j_u)
            order = np.argsort(r_u, kind="stable")
            bounds = np.flatnonzero(np.diff(r_u[order])) + 1
            for grp in np.split(order, bounds):  # graph.py:93-94, one relation at a time
                if len(grp):
                    D[grp] += H_src[pos[grp]] @ W_F[r_u[grp[0]]]
        sub_c = sp.csr_matrix((sub.data.astype(dtype), (sub.row, inv)), shape=(len(rws), len(ucol)))
        Y = sub_c @ D  # graph.py:75, :95
        if cfg.bias:
            Y = Y + p["b"].astype(dtype)
        act = (li < len(cfgs) - 1) or relu_last
        return np.maximum(Y, 0.0) if act else Y

    return layer_at(len(cfgs) - 1, rows)


def input_term_comp_grad_at_rows(cfg: LayerCfg, p: dict, A: sp.csr_matrix, rows, dY_rows, dtype=np.float64):
    """d(loss)/d(weight_I_comp) of an input layer with bases when only `rows` of the layer's output carry gradient
    (`dY_rows`: [len(rows), out], the gradient at the layer's PRE-activation output) — the backward of
    graph.py:69-75 restricted to the columns those rows touch: dW_I[c] = sum_i A[i, c] dY[i];
    dcomp[r, b] = sum_{c = (r, j)} <dW_I[c], V[b, j]>.  Never forms the (R*N) x out gradient, so it runs at the
    FB15k-237 / AM shapes.  Pinned against `layer_backward` by tests/test_oracle_golden.py."""
    rows = np.asarray(rows, dtype=np.int64)
    R, N, B, out = cfg.R, cfg.N, cfg.B, cfg.outdim
    assert cfg.input_layer and B > 0
    sub = A.tocsr()[rows, :].tocoo()
    ucol, inv = np.unique(sub.col.astype(np.int64), return_inverse=True)
    r_u, j_u = ucol // N, ucol % N
    sub_c = sp.csr_matrix((sub.data.astype(dtype), (sub.row, inv)), shape=(len(rows), len(ucol)))
    dD = sub_c.T @ np.asarray(dY_rows, dtype=dtype)                      # [touched columns, out]
    V = p["weight_I"].reshape(B, N, out)
    dots = np.einsum("cf,bcf->cb", dD, V[:, j_u, :].astype(dtype))       # [touched columns, B]
    dcomp = np.zeros((R, B), dtype=dtype)
    np.add.at(dcomp, r_u, dots)
    return dcomp


# ---------------------------------------------------------------------------
# the whole train step, evaluated on the receptive field of the labelled rows
# (node_classification.py:166-193 over graph.py:62-102 / rgcn.py:69-89)
# ---------------------------------------------------------------------------
def _node_groups(j_u):
    """The touched columns grouped by source node: (order, live, groups) with `order` the columns sorted by node,
    `live` the distinct nodes (rising) and `groups` a list of (k, node positions in `live` [n], column ids [n, k]) — the
    nodes with exactly k columns, so that per-node products run as batched matrix products."""
    order = np.argsort(j_u, kind="stable")
    jo = j_u[order]
    starts = np.flatnonzero(np.concatenate([[True], jo[1:] != jo[:-1]])) if len(jo) else np.zeros(0, np.int64)
    counts = np.diff(np.concatenate([starts, [len(jo)]]))
    live = jo[starts]
    groups = []
    for k in np.unique(counts):
        at = np.flatnonzero(counts == k)
        groups.append((int(k), at, order[starts[at][:, None] + np.arange(int(k))[None, :]]))
    return order, live, groups


def _levels_forward(cfgs, params, X, A: sp.csr_matrix, rows, relu_last, chunk, dtype):
    """Forward on the receptive field of `rows`: level l holds what layer l needs to produce its output at the
    row set S_l (S_last = rows; S_{l-1} = the source nodes of the columns the rows of S_l read).  Returns the
    per-layer records, bottom layer first."""
    A = A.tocsr()
    nl = len(cfgs)
    subs = [None] * nl
    rws = np.unique(np.asarray(rows, dtype=np.int64))
    for li in reversed(range(nl)):                       # the row sets, top down
        sub = A[rws, :].tocoo()
        ucol, inv = np.unique(sub.col.astype(np.int64), return_inverse=True)
        N = cfgs[li].N
        r_u, j_u = ucol // N, ucol % N
        src = np.unique(j_u)
        sub_c = sp.csr_matrix((sub.data.astype(dtype), (sub.row, inv)), shape=(len(rws), len(ucol)))
        subs[li] = dict(rows=rws, ucol=ucol, r_u=r_u, j_u=j_u, src=src, pos=np.searchsorted(src, j_u), sub_c=sub_c)
        rws = src
    H = None
    for li in range(nl):                                 # the activations, bottom up
        cfg, p, lv = cfgs[li], params[li], subs[li]
        R, N, B, out = cfg.R, cfg.N, cfg.B, cfg.outdim
        r_u, j_u = lv["r_u"], lv["j_u"]
        D = np.zeros((len(lv["ucol"]), out), dtype=dtype)
        if cfg.input_layer:
            W_I = p["weight_I"]
            if B > 0:   # graph.py:69-72 for the touched columns only: D[c] = comp[r_c] . V[:, j_c, :]
                order, live, groups = _node_groups(j_u)
                # the live nodes' blocks, node-major [live, B, out] (kept in the parameter's dtype; widened per chunk)
                Vl = np.ascontiguousarray(np.transpose(W_I.reshape(B, N, out)[:, live, :], (1, 0, 2)))
                comp = p["weight_I_comp"].astype(dtype)
                per = max(chunk // max(B, 1), 1)
                for k, at, cols in groups:
                    for s in range(0, len(at), per):
                        e = slice(s, s + per)
                        D[cols[e]] = np.matmul(comp[r_u[cols[e]]], Vl[at[e]].astype(dtype))   # [n,k,B] @ [n,B,out]
                lv.update(live=live, groups=groups, Vl=Vl)
            else:
                D += W_I[lv["ucol"]].astype(dtype)
        feat = not (cfg.input_layer and cfg.featureless)
        if feat:
            W_F = p["weight_F"].astype(dtype)
            if B > 0:   # graph.py:83-85
                W_F = np.einsum("rb,bij->rij", p["weight_F_comp"].astype(dtype), W_F)
            # the layer's input at the source nodes: X itself for the bottom layer, else the layer below at S_{l-1} = src
            H_src = X[lv["src"]].astype(dtype) if li == 0 else H
            order = np.argsort(r_u, kind="stable")
            bounds = np.flatnonzero(np.diff(r_u[order])) + 1
            rgroups = [g for g in np.split(order, bounds) if len(g)]
            for grp in rgroups:   # graph.py:93-94, one relation at a time
                D[grp] += H_src[lv["pos"][grp]] @ W_F[r_u[grp[0]]]
            lv.update(W_F=W_F, H_src=H_src, rgroups=rgroups)
        pre = lv["sub_c"] @ D    # graph.py:75, :95
        if cfg.bias:
            pre = pre + p["b"].astype(dtype)
        act = (li < nl - 1) or relu_last
        lv.update(pre=pre, act=act, feat=feat, kinks=_kinks(lv["sub_c"], D, pre, lv) if act else None)
        H = np.maximum(pre, 0.0) if act else pre
    return subs, H


def _kinks(sub_c, D, pre, lv, k=64):
    """The k pre-activations of a ReLU layer that lie closest to zero RELATIVE to the size of the sums they come from
    (`mag` = sum |a| |d| over the row's entries: what a float32 evaluation's rounding error scales with).  Where
    |pre| <~ 1e-6 mag a float32 forward may land on the other side of the kink: the unit's mask, and with it one whole
    term of the gradient of every source node of that row, is then decided by rounding — tests widen their intervals
    there instead of calling it a mismatch.  dict(row [global row ids], col, pre, mag, nodes [per entry: the source
    nodes j of the row's entries])."""
    if pre.size == 0:
        return dict(row=np.zeros(0, np.int64), col=np.zeros(0, np.int64), pre=np.zeros(0), mag=np.zeros(0), nodes=[])
    k = min(k, pre.size)
    absd = np.abs(D)
    rowmag = abs(sub_c) @ absd.max(1)                      # a cheap upper bound per row, to rank candidates
    score = np.abs(pre) / np.maximum(rowmag[:, None], 1e-300)
    flat = np.argpartition(score.ravel(), k - 1)[:k]
    ri, ci = np.unravel_index(flat, pre.shape)
    mags, nodes = [], []
    for i, f in zip(ri, ci):
        a = sub_c[i]
        mags.append(float(np.abs(a.data) @ absd[a.indices, f]))
        nodes.append(np.unique(lv["j_u"][a.indices]))
    return dict(row=lv["rows"][ri], col=ci.astype(np.int64), pre=pre[ri, ci], mag=np.asarray(mags), nodes=nodes)


def rgcn_train_step_at_rows(cfgs, params, X, A: sp.csr_matrix, idx, targets, sample_nodes=None, moments=None, t=1,
                            lr=0.01, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0, relu_last=False, chunk=65536,
                            dtype=np.float64, loss_fn=None, extra_params=None, moments_extra=None):
    """One epoch of node_classification.py:166-193 (forward, CE on the labelled rows, backward, clip_grad_norm_ 1.0,
    Adam) in float64 WITHOUT any (R*N) x out array: everything is evaluated on the receptive field of the labelled
    rows, so the AM shape (1.67 M nodes) and the 10 M-node stress shape run on a host in tens of seconds.  The mirror of
    `rgcn_forward_at_rows` for the backward; pinned against `train_steps` / the reference's own gradients and
    post-Adam parameters by tests/test_oracle_golden.py.

    `params[l]`: the reference's shapes (weight_I `(B*N, out)` or `(R*N, out)`).  `idx` must not repeat a row.
    `sample_nodes` (distinct): node ids whose `weight_I` blocks are returned (gradient, updated parameter, updated
    moments); the squared norm over ALL blocks enters the clip.  `moments`: None (step 1 from zeros) or, per layer, a
    dict name -> (exp_avg, exp_avg_sq) in the reference's shapes for the small parameters and, under "weight_I",
    `(m, v)` of the SAMPLED blocks `[len(sample_nodes), B, out]` (bases; without bases the whole `(R*N, out)` arrays) —
    the state the step starts from.

    `loss_fn(rows, H) -> (loss, dH, extra_grads)`: another loss on the top layer's output `H` at the sorted distinct
    `rows` of `idx` (`targets` unused) — the link-prediction decoder (oracle.lp_oracle: DistMult + BCE over the
    embeddings, link_prediction.py:266-275); `extra_grads` / `extra_params` {name: array}: parameters outside the
    layers (the decoder's `relations`): they enter the clip norm and get their Adam step under `new_extra`
    (from `moments_extra` {name: (exp_avg, exp_avg_sq)} when t > 1).

    Returns dict(loss, logits [len(idx), C], grads [per layer {name: array}] for every parameter but weight_I,
    wI [per layer: rows / grad — bases: the sampled node ids and their `[n, B, out]` blocks; no bases: the touched
    literal rows `r*N+j` and their gradient rows — and live_nodes, the ids with any weight_I gradient], grad_norm, coef,
    new [per layer {name: (param, exp_avg, exp_avg_sq)}] (weight_I: the sampled blocks / touched rows only))."""
    idx = np.asarray(idx, dtype=np.int64)
    assert len(np.unique(idx)) == len(idx), "repeated labelled rows"
    nl = len(cfgs)
    levels, H_top = _levels_forward(cfgs, params, X, A, idx, relu_last, chunk, dtype)
    top_rows = levels[-1]["rows"]
    at_lab = np.searchsorted(top_rows, idx)
    extra_grads = {}
    if loss_fn is None:
        loss, d_sorted = cross_entropy(H_top, at_lab, np.asarray(targets))   # node_classification.py:439-444
    else:
        loss, d_sorted, extra_grads = loss_fn(top_rows, H_top)
    logits = H_top[at_lab]
    sample_nodes = None if sample_nodes is None else np.asarray(sample_nodes, dtype=np.int64)
    if sample_nodes is not None:
        assert len(np.unique(sample_nodes)) == len(sample_nodes), "repeated sample nodes"

    grads = [dict() for _ in range(nl)]
    wI = [None] * nl
    sumsq = 0.0
    dH = d_sorted                                         # gradient at the rows of the top level
    for li in reversed(range(nl)):
        cfg, p, lv = cfgs[li], params[li], levels[li]
        R, N, B, out = cfg.R, cfg.N, cfg.B, cfg.outdim
        r_u, j_u = lv["r_u"], lv["j_u"]
        dPre = dH * (lv["pre"] > 0) if lv["act"] else dH
        g = grads[li]
        if cfg.bias:
            g["b"] = dPre.sum(0)
        dD = lv["sub_c"].T @ dPre                         # [touched columns, out]: autograd of graph.py:75,:95
        if cfg.input_layer:
            if B > 0:
                comp = p["weight_I_comp"].astype(dtype)
                dcomp = np.zeros((R, B), dtype=dtype)
                live, Vl = lv["live"], lv["Vl"]
                keep = None
                if sample_nodes is not None:
                    keep = np.zeros((len(sample_nodes), B, out), dtype=dtype)
                    s_order = np.argsort(sample_nodes, kind="stable")
                    s_sorted = sample_nodes[s_order]
                per = max(chunk // max(B, 1), 1)
                for k, at, cols in lv["groups"]:          # the nodes with k touched columns, `per` of them at a time
                    for s in range(0, len(at), per):
                        cs, ns = cols[s:s + per], at[s:s + per]
                        dDc = dD[cs]                                                  # [n, k, out]
                        cr = comp[r_u[cs]]                                            # [n, k, B]
                        # a node's block: sum_c comp[r_c]^T (x) dD[c]   (einsum 'rb,rij->bij' restricted to the node)
                        blocks = np.matmul(np.transpose(cr, (0, 2, 1)), dDc)          # [n, B, out]
                        sumsq += float((blocks ** 2).sum())
                        # dcomp[r, b] += <dD[c], V[b, j_c]>
                        dots = np.matmul(dDc, np.transpose(Vl[ns].astype(dtype), (0, 2, 1)))   # [n, k, B]
                        np.add.at(dcomp, r_u[cs].ravel(), dots.reshape(-1, B))
                        if keep is not None and len(s_sorted):
                            nd = live[ns]
                            at_s = np.minimum(np.searchsorted(s_sorted, nd), len(s_sorted) - 1)
                            hit = s_sorted[at_s] == nd
                            keep[s_order[at_s[hit]]] = blocks[hit]
                g["weight_I_comp"] = dcomp
                wI[li] = dict(rows=sample_nodes, grad=keep, live_nodes=live)
            else:   # the literal (R*N, out) table: its gradient IS dD at the touched rows, zero elsewhere
                sumsq += float((dD ** 2).sum())
                wI[li] = dict(rows=lv["ucol"], grad=dD, live_nodes=np.unique(j_u))
        dH = None
        if lv["feat"]:
            W_F, H_src, pos = lv["W_F"], lv["H_src"], lv["pos"]
            K = H_src.shape[1]
            dW = np.zeros((R, K, out), dtype=dtype)
            dH = np.zeros((len(lv["src"]), K), dtype=dtype)
            for grp in lv["rgroups"]:
                r = int(r_u[grp[0]])
                dW[r] = H_src[pos[grp]].T @ dD[grp]
                dH[pos[grp]] += dD[grp] @ W_F[r].T             # (a node appears once per relation: no repeats)
            if B > 0:
                g["weight_F"] = np.einsum("rb,rio->bio", p["weight_F_comp"].astype(dtype), dW)
                g["weight_F_comp"] = np.einsum("rio,bio->rb", dW, p["weight_F"].astype(dtype))
            else:
                g["weight_F"] = dW
    for g in grads + [extra_grads]:
        for v in g.values():
            sumsq += float((v.astype(np.float64) ** 2).sum())
    total = np.sqrt(sumsq)
    coef = min(1.0, max_norm / (total + 1e-6)) if max_norm else 1.0   # clip_grad_norm_

    b1, b2 = betas

    def adam(pv, gv, mv, vv):
        gv = gv * coef
        m = b1 * mv + (1 - b1) * gv
        v = b2 * vv + (1 - b2) * gv * gv
        return pv - lr * (m / (1 - b1 ** t)) / (np.sqrt(v) / np.sqrt(1 - b2 ** t) + eps), m, v

    new = [dict() for _ in range(nl)]
    for li in range(nl):
        p, mom = params[li], (moments[li] if moments is not None else {})
        for name, gv in grads[li].items():
            m0, v0 = mom.get(name, (0.0, 0.0))
            new[li][name] = adam(p[name].astype(dtype), gv, np.asarray(m0, dtype=dtype), np.asarray(v0, dtype=dtype))
        w = wI[li]
        if w is None or w["grad"] is None:
            continue
        cfg = cfgs[li]
        if cfg.B > 0:
            V = p["weight_I"].reshape(cfg.B, cfg.N, cfg.outdim)
            pv = np.transpose(V[:, w["rows"], :], (1, 0, 2)).astype(dtype)      # node-major blocks
        else:
            pv = p["weight_I"][w["rows"]].astype(dtype)
        m0, v0 = mom.get("weight_I", (0.0, 0.0))
        if cfg.B <= 0 and np.ndim(m0) == 2:     # (no bases: the whole (R*N, out) moments, read at the touched rows)
            m0, v0 = m0[w["rows"]], v0[w["rows"]]
        new[li]["weight_I"] = adam(pv, w["grad"], np.asarray(m0, dtype=dtype), np.asarray(v0, dtype=dtype))
    mx = moments_extra or {}
    new_extra = {k: adam(np.asarray(extra_params[k], dtype=dtype), gv, *[np.asarray(a, dtype=dtype) for a in mx.get(k, (0.0, 0.0))])
                 for k, gv in extra_grads.items()}
    return dict(loss=loss, logits=logits, grads=grads, wI=wI, grad_norm=total, coef=coef, new=new,
                extra_grads=extra_grads, new_extra=new_extra,
                levels=[dict(rows=lv["rows"], src=lv["src"], ncols=len(lv["ucol"]), kinks=lv.get("kinks")) for lv in levels])
