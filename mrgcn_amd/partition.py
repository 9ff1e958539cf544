"""Node-partitioned full-batch R-GCN over the GPUs of one node.

Rank g of G owns the contiguous source-node range [g*S, (g+1)*S) (S = ceil(N/G)): the rows of
the layer input H for those nodes, their rows of `weight_I` (the dominant parameter and its
Adam state are therefore never replicated nor communicated) and the *columns* (r, j in range)
of the stacked adjacency A.  Per layer

    forward   Y^_g = A[:, cols_g] . M_g          local, the fused kernels on the local plan
              Y_g  = reduce-scatter_rows(Y^_g)    one collective: every rank gets its own rows
    backward  dY^  = all-gather_rows(dY_g)        one collective
              dM_g = A[:, cols_g]^T dY^           local; then dV_g / dX_g local,
              d(comp, W_F, b) all-reduced          (KBs)

One process per GPU, torch.distributed (RCCL over xGMI on GPUs; gloo in the tests).  The
maths equals `mrgcn_amd.models.rgcn.RGCN` on one GPU (tests: logits and gradients agree)."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

import os

from . import functional as Fn
from .layers.graph import GraphConvolution
from .plan import GraphPlan

# MRGCN_PART_COMPACT_GATHER=0: the backward all-gather moves every row of the output gradient (A/B)
_COMPACT_GATHER = os.environ.get("MRGCN_PART_COMPACT_GATHER", "1") != "0"


class NodePartition:
    def __init__(self, num_nodes: int, world: int, rank: int):
        assert 0 <= rank < world
        self.N, self.world, self.rank = int(num_nodes), int(world), int(rank)
        self.S = (self.N + world - 1) // world      # nodes per rank (last ranks may be short)
        self.Np = self.S * world                    # padded node count (reduce-scatter needs equal parts)
        self.j0 = min(rank * self.S, self.N)
        self.j1 = min(self.j0 + self.S, self.N)

    @property
    def n_local(self) -> int:
        return self.j1 - self.j0

    def owner(self, nodes: np.ndarray) -> np.ndarray:
        return np.asarray(nodes) // self.S

    def local_coo(self, rows, cols, vals, num_relations: int):
        """Entries of A whose column's source node lies in this rank's range, re-indexed to the
        local column space r*S + (j - j0); rows stay global (padded to Np by the plan)."""
        rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
        r, j = cols // self.N, cols % self.N
        keep = (j >= self.j0) & (j < self.j1)
        return rows[keep], r[keep] * self.S + (j[keep] - self.j0), np.asarray(vals)[keep]

    def shard_weight_I(self, w: torch.Tensor, S_b: int, node_major: bool = False) -> torch.Tensor:
        """The reference's (S_b*N, out) -> this rank's rows (zero rows for padding nodes): (S_b*S, out),
        or the local layer's node-major (S, S_b, out) block — one contiguous slab of the full table."""
        out = w.shape[1]
        full = w.view(S_b, self.N, out)
        loc = torch.zeros((S_b, self.S, out), dtype=w.dtype, device=w.device)
        loc[:, : self.n_local] = full[:, self.j0:self.j1]
        if node_major:
            return loc.permute(1, 0, 2).contiguous()
        return loc.reshape(S_b * self.S, out)

    def shard_rows(self, X: torch.Tensor) -> torch.Tensor:
        loc = torch.zeros((self.S,) + tuple(X.shape[1:]), dtype=X.dtype, device=X.device)
        loc[: self.n_local] = X[self.j0:self.j1]
        return loc


# ---- collectives (RCCL on GPUs; gloo — CPU staged — in the tests) -----------------------------
def _staged(t: torch.Tensor, group):
    return dist.get_backend(group) == "gloo" and t.is_cuda


def reduce_scatter_rows(x: torch.Tensor, group=None) -> torch.Tensor:
    """x: [world*S, F] partial sums -> this rank's [S, F] rows of the total."""
    world = dist.get_world_size(group)
    S = x.shape[0] // world
    if dist.get_backend(group) == "gloo":
        buf = x.detach().cpu() if x.is_cuda else x.detach().clone()
        dist.all_reduce(buf, group=group)
        r = dist.get_rank(group)
        return buf[r * S:(r + 1) * S].to(x.device).contiguous()
    out = torch.empty((S, x.shape[1]), dtype=x.dtype, device=x.device)
    dist.reduce_scatter_tensor(out, x.contiguous(), group=group)
    return out


def all_gather_rows(x: torch.Tensor, group=None) -> torch.Tensor:
    world = dist.get_world_size(group)
    if _staged(x, group):
        parts = [torch.empty(x.shape, dtype=x.dtype) for _ in range(world)]
        dist.all_gather(parts, x.detach().cpu().contiguous(), group=group)
        return torch.cat(parts, 0).to(x.device)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous(), group=group)
        return torch.cat(parts, 0)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


def all_reduce_sum_(x: torch.Tensor, group=None) -> torch.Tensor:
    if _staged(x, group):
        buf = x.detach().cpu()
        dist.all_reduce(buf, group=group)
        x.copy_(buf.to(x.device))
    else:
        dist.all_reduce(x, group=group)
    return x


def _gathered_flags(flags: torch.Tensor, group) -> torch.Tensor:
    """The row flags of every rank, gathered once and kept on the local flags tensor (a structural row set: the same
    tensor every epoch).  A collective: every rank reaches it in the same backward (partitioned_loss gives every rank
    structural flags, all-zero ones where a rank has no labelled node)."""
    full = getattr(flags, "_mrgcn_gathered", None)
    if full is None or full[0] != flags._version:
        full = (flags._version, all_gather_rows(flags.contiguous(), group).contiguous())
        try:
            flags._mrgcn_gathered = full
        except AttributeError:
            pass
    return full[1]


def _live_row_exchange(flags: torch.Tensor, group):
    """What the backward all-gather of a gradient with structurally known live rows needs, built once per flags tensor
    (collectives: every rank reaches it in the same backward): this rank's live row numbers, the largest count over the
    ranks (all-gather wants equal parts) and, for the gathered [world * n_max] rows, which are real and where they go in
    the [world * S] row space."""
    ent = getattr(flags, "_mrgcn_exchange", None)
    if ent is not None and ent[0] == flags._version:
        return ent[1]
    world, S = dist.get_world_size(group), int(flags.numel())
    dev = flags.device
    mine = torch.nonzero(flags, as_tuple=False).flatten()
    counts = all_gather_rows(torch.tensor([int(mine.numel())], dtype=torch.int64, device=dev), group).cpu().tolist()
    n_max = max(max(counts), 1)
    send_idx = torch.zeros(n_max, dtype=torch.int64, device=dev)
    send_idx[: mine.numel()] = mine
    all_idx = all_gather_rows(send_idx, group).view(world, n_max)           # every rank's (padded) local row numbers
    valid = (torch.arange(n_max, device=dev)[None, :] < torch.tensor(counts, device=dev)[:, None])
    dst = (all_idx + torch.arange(world, device=dev)[:, None] * S)[valid]      # global row of every real gathered row
    src = torch.nonzero(valid.flatten(), as_tuple=False).flatten()             # its position among the gathered rows
    info = dict(send_idx=send_idx, n_max=n_max, src=src, dst=dst, total=int(sum(counts)))
    try:
        flags._mrgcn_exchange = (flags._version, info)
    except AttributeError:
        pass
    return info


class _ReduceScatterRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return reduce_scatter_rows(x, group)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        meta = Fn._grad_meta(g)
        structural = (meta is not None and meta.get("structural") and not meta.get("sparse_rows")
                      and meta["row_live"] is not None)
        if structural and g.is_cuda and g.shape[1] <= 16 and Fn._SUPPORT and Fn._LIVE_COLS and _COMPACT_GATHER:
            # only the rows that can hold anything travel (AM shape, layer 0: 5 565 of 1.67 M rows): the ranks exchange
            # their live rows, the rest of the gathered gradient stays unwritten — the local layer's backward reads the
            # flagged rows only (functional._backward_on_support)
            flags = meta["row_live"]
            full_flags = _gathered_flags(flags, ctx.group)
            ex = _live_row_exchange(flags, ctx.group)
            world = dist.get_world_size(ctx.group)
            send = g.index_select(0, ex["send_idx"])
            recv = all_gather_rows(send, ctx.group)
            out = torch.empty((world * g.shape[0], g.shape[1]), dtype=g.dtype, device=g.device)
            out.index_copy_(0, ex["dst"], recv.index_select(0, ex["src"]))
            Fn._set_grad_meta(out, full_flags, meta["relu_applied"], structural=True, sparse_rows=True)
            return out, None
        out = all_gather_rows(g, ctx.group)
        if structural:
            # the rows that can hold anything are known on every rank: the gathered gradient carries the gathered
            # flags, and the local layer's backward runs on the gradient support of that row set (functional.py)
            Fn._set_grad_meta(out, _gathered_flags(meta["row_live"], ctx.group), meta["relu_applied"], structural=True)
        return out, None


class _ReluRows(torch.autograd.Function):
    """relu(Y) whose backward hands the note on its gradient (functional._grad_meta: which rows can hold anything)
    on to the masked gradient — torch.relu's backward returns a fresh tensor without it."""

    @staticmethod
    def forward(ctx, Y):
        H = torch.relu(Y)
        ctx.save_for_backward(H)
        return H

    @staticmethod
    def backward(ctx, g):
        (H,) = ctx.saved_tensors
        meta = Fn._grad_meta(g)
        if meta is not None and meta.get("relu_applied"):
            return g
        out = Fn.relu_bwd(g.contiguous(), H) if g.is_cuda else g * (H > 0)
        if meta is not None and meta.get("structural") and not meta.get("sparse_rows"):
            Fn._set_grad_meta(out, meta["row_live"], True, structural=True)
        return out


class _ZeroLoss(torch.autograd.Function):
    """The loss share of a rank without a labelled node: 0, with a zero gradient that says so structurally (all-zero
    row flags), so that this rank takes the same path through the backward — and the same collectives — as the others."""

    @staticmethod
    def forward(ctx, logits, flags):
        ctx.flags, ctx.shape = flags, tuple(logits.shape)
        return torch.zeros((), dtype=torch.float32, device=logits.device)

    @staticmethod
    def backward(ctx, g):
        d = torch.zeros(ctx.shape, dtype=torch.float32, device=ctx.flags.device)
        Fn._set_grad_meta(d, ctx.flags, False, structural=True)
        return d, None


class _AllGatherRows(torch.autograd.Function):
    """[S, F] own rows -> [world*S, F] all rows; backward: every rank's gradient w.r.t. all rows is
    summed and each rank keeps its own rows (reduce-scatter)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return all_gather_rows(x.contiguous(), group)

    @staticmethod
    def backward(ctx, g):
        return reduce_scatter_rows(g.contiguous(), ctx.group), None


# ---- the partitioned model ----------------------------------------------------------------------
class PartitionedRGCN(nn.Module):
    """`RGCN` (models/rgcn.py) with node-partitioned layers.  `modules` as for RGCN.  Every
    rank constructs it with the same seed; `load_full_state` shards a reference-shaped state."""

    def __init__(self, modules, num_relations, num_nodes, num_bases, featureless, bias, part: NodePartition,
                 group=None, link_prediction=False):
        super().__init__()
        if link_prediction:  # DistMult relation embeddings (rgcn.py:55-61): small, replicated
            self.relations = nn.Parameter(torch.empty((num_relations, modules[-1][1])))
            nn.init.xavier_uniform_(self.relations)
        self.part, self.group = part, group
        self.num_nodes, self.num_relations, self.num_bases = num_nodes, num_relations, num_bases
        self.layers = nn.ModuleDict()
        self.relu = []
        for i, (indim, outdim, _t, act) in enumerate(modules):
            first = i == 0
            # local layer: S source nodes per relation block (rows of weight_I for my nodes only)
            self.layers[f"layer_{i}"] = GraphConvolution(
                indim, outdim, num_relations, part.S, num_bases=num_bases, bias=bias, input_layer=first,
                featureless=featureless if first else False)
            self.relu.append(isinstance(act, nn.ReLU))
        self.num_layers = len(self.layers)
        self.plan = None

    def sharded_parameters(self):
        return [l.weight_I for l in self.layers.values() if l.weight_I is not None]

    def replicated_parameters(self):
        sh = {id(p) for p in self.sharded_parameters()}
        return [p for p in self.parameters() if id(p) not in sh]

    @torch.no_grad()
    def load_full_state(self, state: dict):
        """`state`: an `RGCN.state_dict()` in the reference's shapes (keys layers.layer_<i>.<name>)."""
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
        lr, lc, lv = self.part.local_coo(rows, cols, vals, self.num_relations)
        A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([lr, lc])), torch.from_numpy(lv),
                                    (self.part.Np, self.num_relations * self.part.S)).to(device)
        self.plan = GraphPlan(A, self.part.S, self.num_relations,
                              operand_row_bytes=sorted({l.operand_row_bytes() for l in self.layers.values()}))
        return self.plan

    def forward(self, X_local):
        """X_local: this rank's [S, K] input rows (None when featureless).  Returns its [S, C]
        rows of the logits."""
        H = X_local
        for i, layer in enumerate(self.layers.values()):
            Yp = Fn.rgcn_layer(self.plan, _NoBias(layer), H, relu=False)   # [Np, out] partial sums
            Y = _ReduceScatterRows.apply(Yp, self.group)                   # [S, out] own rows
            if layer.bias:
                Y = Y + layer.b
            H = _ReluRows.apply(Y) if self.relu[i] else Y
        return H

    @torch.no_grad()
    def sync_replicated(self, src: int = 0):
        """Replicated parameters must start identical on every rank."""
        for q in self.replicated_parameters():
            if _staged(q, self.group):
                buf = q.detach().cpu()
                dist.broadcast(buf, src, group=self.group)
                q.copy_(buf.to(q.device))
            else:
                dist.broadcast(q.data, src, group=self.group)

    def allreduce_replicated_grads(self):
        """Gradients of the small replicated parameters are sums over all columns of A."""
        for p in self.replicated_parameters():
            if p.grad is not None:
                all_reduce_sum_(p.grad, self.group)

    # -- overlapped form: every replicated gradient is reduced as soon as autograd has accumulated it, on the
    #    communication stream, while the backward of the layers below it still computes (the dominant local work
    #    of a step — dM = A^T dY, the node table's dV, the transforms' dX / dW — needs none of these sums)
    def begin_overlapped_grad_reduce(self):
        """Arms post-accumulate hooks on the replicated parameters; returns a `finish()` that waits for the
        reductions (call it before the optimizer reads the gradients).  RCCL: asynchronous all-reduce per
        parameter on the process group's own stream; gloo (tests): the staged blocking form inside the hook —
        same arithmetic, no overlap."""
        pending, hooks = [], []
        nccl = dist.get_backend(self.group) == "nccl"

        def hook(p):
            if p.grad is None:
                return
            if nccl:
                pending.append(dist.all_reduce(p.grad, group=self.group, async_op=True))
            else:
                all_reduce_sum_(p.grad, self.group)

        for q in self.replicated_parameters():
            hooks.append(q.register_post_accumulate_grad_hook(hook))

        def finish():
            for h in hooks:
                h.remove()
            for w in pending:
                w.wait()   # orders the current stream after the reduction (no host block with RCCL)

        return finish


class _NoBias:
    """View of a layer without its bias (added after the reduction, once)."""

    def __init__(self, layer):
        self._l = layer

    def __getattr__(self, k):
        if k == "bias":
            return False
        return getattr(self._l, k)


def _label_shard(idx_global, targets, part: NodePartition, dev):
    """This rank's labelled rows (local row index, target) as device tensors + its share of the mean, built ONCE per
    label set and kept on the partition object: a step then uploads nothing and builds no host masks
    (the label set of a run is fixed: node_classification.py:378-381)."""
    cache = part.__dict__.setdefault("_label_shards", {})
    key = (id(idx_global), id(targets), str(dev))
    ent = cache.get(key)
    if ent is not None and ent[0] is idx_global and ent[1] is targets:
        return ent[2]
    idx_np = idx_global.detach().cpu().numpy() if torch.is_tensor(idx_global) else np.asarray(idx_global)
    tgt_np = targets.detach().cpu().numpy() if torch.is_tensor(targets) else np.asarray(targets)
    mine = (idx_np >= part.j0) & (idx_np < part.j1)
    if mine.any():
        shard = (torch.from_numpy(idx_np[mine] - part.j0).to(dev), torch.from_numpy(tgt_np[mine]).to(dev),
                 float(mine.sum()) / len(idx_np))
    else:
        shard = (None, None, 0.0)
    if len(cache) > 16:
        cache.clear()
    cache[key] = (idx_global, targets, shard)
    return shard


def partitioned_loss(logits_local, idx_global, targets, part: NodePartition, group=None):
    """Mean cross-entropy over ALL labelled nodes; each rank differentiates its own rows."""
    from .train import categorical_crossentropy
    li, lt, share = _label_shard(idx_global, targets, part, logits_local.device)
    if li is not None:
        local = categorical_crossentropy(logits_local, li, lt) * share
    elif logits_local.is_cuda and Fn._SUPPORT:
        # (the predicate under which train.categorical_crossentropy gives the labelled ranks a structural gradient:
        # every rank must reach `_ReduceScatterRows.backward` with the same kind of note, or the ranks' collective
        # sequences differ — with MRGCN_SUPPORT=0 nobody has one and this rank takes the plain zero loss below)
        zf = part.__dict__.get("_zero_flags")
        if zf is None or zf.numel() != logits_local.shape[0] or zf.device != logits_local.device:
            zf = part.__dict__["_zero_flags"] = torch.zeros((logits_local.shape[0],), dtype=torch.uint8,
                                                            device=logits_local.device)
        local = _ZeroLoss.apply(logits_local, zf)
    else:
        local = (logits_local * 0.0).sum()
    total = local.detach().clone()
    all_reduce_sum_(total, group)
    return local, total


def _backward_and_step(model, local, optimizer, row_sparse):
    """backward + all-reduce of the replicated gradients + optimizer step, with the same gradient-sparsity
    machinery as the single-GPU `train_step` (live-column backward, row-sparse weight_I gradient and Adam on
    this rank's shard of the node table)."""
    from .train import ClipAdam, _ROW_SPARSE_DEFAULT
    params = [p for g in optimizer.param_groups for p in g["params"]]
    Fn.clear_row_grads(params)
    optimizer.zero_grad(set_to_none=True)
    sparse_ok = (row_sparse is not False and _ROW_SPARSE_DEFAULT and isinstance(optimizer, ClipAdam)
                 and all(float(g["weight_decay"]) == 0.0 for g in optimizer.param_groups))
    prev = Fn.row_sparse_weight_grad(sparse_ok)
    finish = model.begin_overlapped_grad_reduce()
    try:
        local.backward()
    finally:
        Fn.row_sparse_weight_grad(prev)
        finish()
    optimizer.step()


def partitioned_train_step(model: PartitionedRGCN, X_local, idx_global, targets, optimizer, row_sparse=None):
    logits = model(X_local)
    local, total = partitioned_loss(logits, idx_global, targets, model.part, model.group)
    _backward_and_step(model, local, optimizer, row_sparse)
    return total


class GraphedPartitionedStep:
    """`partitioned_train_step` captured into a hipGraph — local kernels AND the RCCL collectives between them — and
    replayed: at 8 ranks every rank's share of an AM-sized graph is launch-latency territory (AM/8 on one rank:
    1.85 ms eager, 1.58 ms replayed).  Needs the nccl backend (gloo's CPU-staged collectives cannot be captured),
    `ClipAdam(capturable=True)` and static shapes; the warm-up steps are real optimizer steps, as in
    `train.GraphedTrainStep`.  Every rank must construct it (the capture runs the collectives' bookkeeping on all
    ranks alike).  Exercised with one rank over RCCL on the test box; never run on more than one GPU so far."""

    def __init__(self, model: PartitionedRGCN, X_local, idx_global, targets, optimizer, warmup: int = 3, row_sparse=None):
        if not getattr(optimizer, "capturable", False):
            raise RuntimeError("GraphedPartitionedStep needs ClipAdam(..., capturable=True)")
        if dist.is_initialized() and dist.get_backend(model.group) != "nccl":
            raise RuntimeError("GraphedPartitionedStep needs the nccl (RCCL) backend")
        args = (model, X_local, idx_global, targets, optimizer, row_sparse)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                partitioned_train_step(*args)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            self.loss = partitioned_train_step(*args)
        self.warmup_steps = max(warmup, 1)

    def __call__(self):
        self.graph.replay()
        return self.loss


def partitioned_lp_step(model: PartitionedRGCN, X_local, triples, labels, optimizer):
    """One link-prediction step on the partitioned encoder (BASELINE config 4 over several GPUs):
    every rank computes the embeddings of its node range, the (small) embedding table is
    all-gathered, each rank scores triples `rank::world`, and the decoder's gradient w.r.t. the
    table returns to the owners by reduce-scatter.  `triples` [n, 3] / `labels` [n] are the full
    batch (positives + negatives), identical on every rank.  Returns the mean BCE over all triples."""
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
    _backward_and_step(model, local, optimizer, None)
    return total
