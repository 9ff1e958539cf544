"""DistMult link-prediction decoder on the R-GCN encoder's output — the numeric core of
`mrgcn/tasks/link_prediction.py` (scores :645-665, loss :550-554, negative sampling :247-263,
ranks :593-643, metrics :373-420), same function names and argument meaning, computed by the
HIP kernels of `csrc/distmult.hip` through the C ABI.  The reference's run loop, logging, TSV
writers and mini-batch machinery are out of scope."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from .. import _lib


_SORTED_BWD_MIN = 4096  # below this the scatter kernel's atomics do not collide enough to matter


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _f32_rows(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.MrgcnError(f"{what} must live on the GPU (mrgcn_amd has no CPU decoder)")
    if t.dtype != torch.float32 or t.dim() != 2:
        raise TypeError(f"{what} must be a 2-D float32 tensor")
    return t if t.stride(1) == 1 else t.contiguous()


def _triples(data, device) -> torch.Tensor:
    """(s, p, o) 1-D index tensors, or an [n, 3] array -> contiguous int64 [n, 3] on `device`."""
    if isinstance(data, (tuple, list)):
        si, pi, oi = (torch.as_tensor(x) for x in data)
        if not (si.dim() == pi.dim() == oi.dim() == 1 and len(si) == len(pi) == len(oi)):
            raise NotImplementedError("score_distmult_bc: only equally long 1-D index tensors (the "
                                      "train_model call); ranking goes through compute_ranks_fast")
        t = torch.stack([si.long(), pi.long(), oi.long()], 1)
    else:
        t = torch.as_tensor(data).long()
        if t.dim() != 2 or t.shape[1] != 3:
            raise ValueError("facts must be [n, 3]")
    return t.to(device).contiguous()


class SortedTriples:
    """The orders of a FIXED triple set by subject / predicate / object (what the sorted decoder backward walks), built
    once: a full-batch run scores the same training facts every epoch (tasks/link_prediction.py:231-263 — only the 20 %
    corrupted copies are drawn anew), so three per-epoch sorts of 326 k keys (0.16 ms of a 1.7 ms epoch at the
    FB15k-237 shape) shrink to nothing.  Pass it to `score_distmult_bc(..., static=...)` when the first
    `len(static)` rows of the scored triples ARE these facts, in this order (checked by identity / version)."""

    def __init__(self, triples: torch.Tensor, num_nodes: int, num_relations: int):
        t = triples.contiguous()
        if not t.is_cuda or t.dtype != torch.int64 or t.dim() != 2 or t.shape[1] != 3:
            raise TypeError("SortedTriples: int64 [n, 3] triples on the GPU")
        self.triples, self.version, self.n = t, t._version, int(t.shape[0])
        # built once, so the sorts may be anything: stable argsorts with a secondary key — inside a run of equal
        # predicate the facts follow their subject (the subject's embedding row repeats for consecutive facts instead of
        # being gathered anew), inside a run of equal subject / object the other end rises
        s_, p_, o_ = t[:, 0], t[:, 1], t[:, 2]
        nn_ = int(num_nodes)
        self.order = [torch.argsort(s_ * nn_ + o_, stable=True), torch.argsort(p_ * nn_ + s_, stable=True),
                      torch.argsort(o_ * nn_ + s_, stable=True)]
        self._tail = None

    def tail_orders(self, nt: int, num_nodes: int, num_relations: int):
        """Buffers for the orders of the `nt` triples behind the fixed facts (+ the counting sort's workspace), kept."""
        if self._tail is None or self._tail[0] != (nt, num_nodes, num_relations):
            dev = self.triples.device
            ws = torch.empty(int(_lib.load().mrgcn_distmult_orders_counting_workspace(num_nodes, num_relations)),
                             dtype=torch.uint8, device=dev)
            self._tail = ((nt, num_nodes, num_relations),
                          [torch.empty(nt, dtype=torch.int64, device=dev) for _ in range(3)] + [ws])
        return self._tail[1]

    def __len__(self):
        return self.n

    def covers(self, triples: torch.Tensor) -> bool:
        """May the stored orders serve `triples`?  The facts are unchanged since the orders were built and `triples`
        starts with them (compared once, outside stream captures; afterwards the caller's contract)."""
        if self.triples._version != self.version or triples.shape[0] < self.n:
            return False
        if not getattr(self, "_checked", False) and not torch.cuda.is_current_stream_capturing():
            if not torch.equal(triples[: self.n], self.triples):
                return False
            self._checked = True
        return True


class _DistMultScore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E, Rel, triples, static=None):
        ctx.static = static
        lib = _lib.load()
        n, H = triples.shape[0], E.shape[1]
        scores = torch.empty(n, dtype=torch.float32, device=E.device)
        _lib.check(lib.mrgcn_distmult_score_f32(_ptr(E), E.stride(0), _ptr(Rel), Rel.stride(0), H,
                                                _ptr(triples), n, _ptr(scores), _stream()), "distmult_score")
        ctx.save_for_backward(E, Rel, triples)
        return scores

    @staticmethod
    def backward(ctx, g):
        E, Rel, triples = ctx.saved_tensors
        lib = _lib.load()
        g = g.contiguous().float()
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            # (both gradients out of ONE zeroed buffer: one fill launch per step instead of two)
            off = (E.numel() + 3) // 4 * 4   # (dR keeps the buffer's 16-byte alignment)
            buf = torch.zeros(off + Rel.numel(), dtype=torch.float32, device=E.device)
            dE, dR = buf[:E.numel()].view(E.shape), buf[off:].view(Rel.shape)
        else:
            dE = torch.zeros_like(E, memory_format=torch.contiguous_format) if ctx.needs_input_grad[0] else None
            dR = torch.zeros_like(Rel, memory_format=torch.contiguous_format) if ctx.needs_input_grad[1] else None
        n = triples.shape[0]
        st = ctx.static
        if st is not None and st.covers(triples) and os.environ.get("MRGCN_LP_SORTED_BWD", "1") != "0":
            # the fixed facts through their stored orders (runs of equal targets summed in registers), the few
            # freshly drawn ones behind them through the scatter kernel: no sort in the epoch
            ns = len(st)
            _lib.check(lib.mrgcn_distmult_score_bwd_sorted_f32(
                _ptr(E), E.stride(0), _ptr(Rel), Rel.stride(0), E.shape[1], _ptr(triples), ns, _ptr(g),
                _ptr(st.order[0]), _ptr(st.order[1]), _ptr(st.order[2]), _ptr(dE), dE.stride(0) if dE is not None else 0,
                _ptr(dR), dR.stride(0) if dR is not None else 0, _stream()), "distmult_score_bwd_sorted")
            if n > ns:
                tail, gtail, nt = C.c_void_p(triples.data_ptr() + 24 * ns), C.c_void_p(g.data_ptr() + 4 * ns), n - ns
                if (nt >= 1024 and E.shape[0] <= (1 << 22) and Rel.shape[0] <= (1 << 22)
                        and os.environ.get("MRGCN_LP_COUNTING", "1") != "0"):
                    # enough of them to collide in the scatter kernel's atomics (54 k corrupted facts on 237 relation
                    # rows: 310 us at the FB15k-237 shape): a counting sort (four launches) and the sorted passes
                    o3 = st.tail_orders(nt, E.shape[0], Rel.shape[0])
                    _lib.check(lib.mrgcn_distmult_orders_counting(
                        tail, nt, E.shape[0], Rel.shape[0], _ptr(o3[0]), _ptr(o3[1]), _ptr(o3[2]), _ptr(o3[3]),
                        o3[3].numel(), _stream()), "distmult_orders_counting")
                    _lib.check(lib.mrgcn_distmult_score_bwd_sorted_f32(
                        _ptr(E), E.stride(0), _ptr(Rel), Rel.stride(0), E.shape[1], tail, nt, gtail, _ptr(o3[0]),
                        _ptr(o3[1]), _ptr(o3[2]), _ptr(dE), dE.stride(0) if dE is not None else 0, _ptr(dR),
                        dR.stride(0) if dR is not None else 0, _stream()), "distmult_score_bwd_sorted")
                else:
                    _lib.check(lib.mrgcn_distmult_score_bwd_f32(
                        _ptr(E), E.stride(0), _ptr(Rel), Rel.stride(0), E.shape[1], tail, nt, gtail, _ptr(dE),
                        dE.stride(0) if dE is not None else 0, _ptr(dR), dR.stride(0) if dR is not None else 0,
                        _stream()), "distmult_score_bwd")
            return dE, dR, None, None
        if n >= _SORTED_BWD_MIN and os.environ.get("MRGCN_LP_SORTED_BWD", "1") != "0":
            # runs of equal subject / predicate / object are summed in registers (three passes over
            # sorted triples) instead of one float atomic per triple and feature
            order = [torch.empty(n, dtype=torch.int64, device=E.device) for _ in range(3)]
            ws = torch.empty(int(lib.mrgcn_distmult_orders_workspace(n)), dtype=torch.uint8, device=E.device)
            _lib.check(lib.mrgcn_distmult_orders(_ptr(triples), n, E.shape[0], Rel.shape[0], _ptr(order[0]),
                                                 _ptr(order[1]), _ptr(order[2]), _ptr(ws), ws.numel(), _stream()),
                       "distmult_orders")
            _lib.check(lib.mrgcn_distmult_score_bwd_sorted_f32(
                _ptr(E), E.stride(0), _ptr(Rel), Rel.stride(0), E.shape[1], _ptr(triples), n, _ptr(g),
                _ptr(order[0]), _ptr(order[1]), _ptr(order[2]), _ptr(dE), dE.stride(0) if dE is not None else 0,
                _ptr(dR), dR.stride(0) if dR is not None else 0, _stream()), "distmult_score_bwd_sorted")
        else:
            _lib.check(lib.mrgcn_distmult_score_bwd_f32(
                _ptr(E), E.stride(0), _ptr(Rel), Rel.stride(0), E.shape[1], _ptr(triples), n,
                _ptr(g), _ptr(dE), dE.stride(0) if dE is not None else 0, _ptr(dR),
                dR.stride(0) if dR is not None else 0, _stream()), "distmult_score_bwd")
        return dE, dR, None, None


def score_distmult_bc(data, node_embeddings, edge_embeddings, static: "SortedTriples | None" = None):
    """link_prediction.py:645-665 for the 1-D (s, p, o) index tensors train_model passes (or an int64 [n, 3] tensor).
    `static`: the stored orders of the facts the triples START with (SortedTriples): the backward then sorts nothing."""
    E = _f32_rows(node_embeddings, "node_embeddings")
    Rel = _f32_rows(edge_embeddings, "edge_embeddings")
    return _DistMultScore.apply(E, Rel, _triples(data, E.device), static)


class _BceLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        lib = _lib.load()
        x = x.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        _lib.check(lib.mrgcn_bce_logits_f32(_ptr(x), _ptr(y), x.numel(), _ptr(loss), _ptr(dx), _stream()),
                   "bce_logits")
        ctx.save_for_backward(dx)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None


def binary_crossentropy(Y_hat, Y, criterion=None):
    """link_prediction.py:550-554 with criterion = nn.BCEWithLogitsLoss() (:57); `criterion` is
    accepted for signature parity and must be that loss (or None)."""
    if criterion is not None and not isinstance(criterion, torch.nn.BCEWithLogitsLoss):
        raise NotImplementedError("only nn.BCEWithLogitsLoss is implemented on the device")
    if not Y_hat.is_cuda:
        raise _lib.MrgcnError("binary_crossentropy: scores must live on the GPU")
    return _BceLogits.apply(Y_hat.float(), Y.to(Y_hat.device).float().contiguous())


def sample_negatives(batch_data: np.ndarray, rng=np.random):
    """train_model's within-batch corruption (link_prediction.py:239-263): 20 % of the positives
    are copied, half get a random in-batch head, half a random in-batch tail.  Returns
    (corrupted [ncorrupt, 3], labels float32 [n + ncorrupt]).  `rng`: np.random or a RandomState
    (the reference uses the global np.random)."""
    n = batch_data.shape[0]
    batch_nodes = np.union1d(batch_data[:, 0], batch_data[:, 2])
    ncorrupt = n // 5
    neg_idx = rng.choice(np.arange(n), ncorrupt, replace=False)
    nhead = ncorrupt // 2
    ntail = ncorrupt - nhead
    corrupted = np.empty((ncorrupt, 3), dtype=int)
    corrupted[:] = batch_data[neg_idx]
    corrupted[:nhead, 0] = rng.choice(batch_nodes, nhead)
    if ntail:
        corrupted[-ntail:, 2] = rng.choice(batch_nodes, ntail)
    Y = np.ones(n + ncorrupt, dtype=np.float32)
    if ncorrupt:
        Y[-ncorrupt:] = 0
    return corrupted, Y


def sample_negatives_device(batch_data: torch.Tensor, generator=None):
    """The same corruption scheme drawn on the device (a torch generator instead of np.random, so
    not the reference's random stream): `batch_data` int64 [n, 3] on the GPU -> (corrupted
    [n // 5, 3], labels float32 [n + n // 5]), no host round trip."""
    n = batch_data.shape[0]
    dev = batch_data.device
    # the nodes of the batch: a sort-based unique (half of this function's device time) — kept for the tensor it was
    # computed from while that tensor is unchanged (a full-batch run passes the same training facts every epoch)
    cached = getattr(batch_data, "_mrgcn_nodes", None)
    if cached is not None and cached[0] == batch_data._version:
        nodes = cached[1]
    else:
        nodes = torch.unique(torch.cat([batch_data[:, 0], batch_data[:, 2]]))
        try:
            batch_data._mrgcn_nodes = (batch_data._version, nodes)
        except AttributeError:
            pass
    ncorrupt = n // 5
    # ncorrupt distinct facts: a keyed bijection of [0, n) evaluated at 0..ncorrupt-1 (one launch; a randperm sorts n keys)
    seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=dev, generator=generator)
    neg_idx = torch.empty(ncorrupt, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mrgcn_random_subset_i64(n, ncorrupt, _ptr(seed), _ptr(neg_idx), _stream()), "random_subset")
    nhead = ncorrupt // 2
    ntail = ncorrupt - nhead
    corrupted = batch_data[neg_idx].clone()
    pick = lambda k: nodes[torch.randint(0, nodes.numel(), (k,), device=dev, generator=generator)]  # noqa: E731
    corrupted[:nhead, 0] = pick(nhead)
    if ntail:
        corrupted[ncorrupt - ntail:, 2] = pick(ntail)
    Y = torch.ones(n + ncorrupt, dtype=torch.float32, device=dev)
    Y[n:] = 0
    return corrupted, Y


class DeviceNegativeSampler:
    """`sample_negatives_device` for a FIXED fact set, without its per-epoch torch traffic: the facts sit at the head of
    one [n + n // 5, 3] buffer, a single launch (mrgcn_corrupt_triples_i64) writes the corrupted copies behind them, the
    labels are built once.  `triples, labels = sampler()` — the same tensors every call (their contents change): what a
    captured epoch wants.  The draws come from a 64-bit seed taken from torch's generator per call."""

    def __init__(self, facts: torch.Tensor, generator=None):
        n = int(facts.shape[0])
        dev = facts.device
        self.n, self.ncorrupt = n, n // 5
        self.nhead = self.ncorrupt // 2
        self.generator = generator
        self.buf = torch.empty((n + self.ncorrupt, 3), dtype=torch.int64, device=dev)
        self.buf[:n] = facts
        self.facts = self.buf[:n]
        self.nodes = torch.unique(torch.cat([facts[:, 0], facts[:, 2]]))
        self.labels = torch.ones(n + self.ncorrupt, dtype=torch.float32, device=dev)
        self.labels[n:] = 0

    def __call__(self):
        dev = self.buf.device
        seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=dev, generator=self.generator)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mrgcn_corrupt_triples_i64(
                _ptr(self.buf), self.n, _ptr(self.nodes), int(self.nodes.numel()), _ptr(seed), self.ncorrupt, self.nhead,
                C.c_void_p(self.buf.data_ptr() + 24 * self.n), _stream()), "corrupt_triples")
        return self.buf, self.labels


def filter_lists(data: np.ndarray):
    """The filter of compute_ranks_fast(filtered=True) (truedicts :568-591 + filter_scores_
    :667-689) as CSR lists: per fact, the sorted nodes that are other true objects of (s, p)
    [tail corruption] / other true subjects of (p, o) [head corruption] among `data`."""
    data = np.asarray(data, dtype=np.int64)
    nf = len(data)

    def build(key_a, key_b, ans):
        # group facts by (key_a, key_b); members of a group = unique answers
        order = np.lexsort((ans, key_b, key_a))
        ka, kb, an = key_a[order], key_b[order], ans[order]
        new_grp = np.ones(nf, bool)
        new_grp[1:] = (ka[1:] != ka[:-1]) | (kb[1:] != kb[:-1])
        uniq = new_grp.copy()
        uniq[1:] |= an[1:] != an[:-1]
        gid_sorted = np.cumsum(new_grp) - 1                # group id per sorted fact
        ngrp = int(gid_sorted[-1]) + 1 if nf else 0
        gptr = np.zeros(ngrp + 1, np.int64)
        np.add.at(gptr, gid_sorted[uniq] + 1, 1)
        gptr = np.cumsum(gptr)
        members = an[uniq]                                  # sorted within each group
        gid = np.empty(nf, np.int64)
        gid[order] = gid_sorted
        cnt = gptr[gid + 1] - gptr[gid] - 1                 # minus the fact's own answer
        ptr = np.zeros(nf + 1, np.int64)
        np.cumsum(cnt, out=ptr[1:])
        idx = np.empty(int(ptr[-1]), np.int32)
        # expand: for fact f, members of its group except ans[f]
        rep = np.repeat(np.arange(nf), cnt + 1)
        off = np.arange(len(rep)) - np.repeat(np.cumsum(cnt + 1) - (cnt + 1), cnt + 1)
        cand = members[gptr[gid[rep]] + off]
        keep = cand != ans[rep]
        idx[:] = cand[keep]
        return ptr, idx

    if nf == 0:
        z = np.zeros(1, np.int64)
        return z, np.zeros(0, np.int32), z.copy(), np.zeros(0, np.int32)
    tp, ti = build(data[:, 0], data[:, 1], data[:, 2])
    hp, hi = build(data[:, 1], data[:, 2], data[:, 0])
    return tp, ti, hp, hi


def compute_ranks_fast(data, node_embeddings, edge_embeddings, batch_size=100, filtered=False):
    """link_prediction.py:593-643.  Returns the int64 [2 * num_facts] ranks (tail corruption
    then head corruption) on the embeddings' device.  `batch_size` (the reference's
    mrr_batchsize, a memory knob for its [facts, nodes] score matrix) is accepted and unused:
    scores are never materialised here."""
    E = _f32_rows(node_embeddings.detach(), "node_embeddings")
    Rel = _f32_rows(edge_embeddings.detach(), "edge_embeddings")
    lib = _lib.load()
    dev = E.device
    facts_np = data.cpu().numpy() if torch.is_tensor(data) else np.asarray(data)
    tr = _triples(facts_np, dev)
    nf, N, H = tr.shape[0], E.shape[0], E.shape[1]
    ranks = torch.empty(2 * nf, dtype=torch.int64, device=dev)
    if nf == 0:
        return ranks
    ws_bytes = lib.mrgcn_distmult_ranks_workspace(N, H, nf)
    ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.int32, device=dev)
    lists = [None] * 4
    if filtered:
        lists = [torch.from_numpy(a).to(dev) for a in filter_lists(facts_np)]
        lists = [a if a.numel() else torch.zeros(1, dtype=a.dtype, device=dev) for a in lists]
    _lib.check(lib.mrgcn_distmult_ranks(_ptr(E), E.stride(0), N, _ptr(Rel), Rel.stride(0), H, _ptr(tr), nf,
                                        _ptr(lists[0]), _ptr(lists[1]), _ptr(lists[2]), _ptr(lists[3]),
                                        _ptr(ws), ws_bytes, _ptr(ranks), _stream()), "distmult_ranks")
    return ranks


def mrr_hits(ranks, K=(1, 3, 10)):
    """One batch's metrics as test_model computes them (link_prediction.py:403-407)."""
    r = ranks.float()
    return torch.mean(1.0 / r).item(), [float(torch.mean((ranks <= k).float())) for k in K]
