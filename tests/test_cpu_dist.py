"""World-size-2 gloo test of the N > 1 glue used by bench.py (replicas: barrier + max over ranks)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mrgcn_amd import dist as D
    w, r, lr = D.init(backend="gloo")
    assert (w, r, lr) == (world, rank, rank)
    D.barrier()
    local = 10.0 + 5.0 * rank  # rank 1 is the slow one
    out[rank] = D.replica_value(local)
    D.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_max_over_ranks_gloo_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: 15.0, 1: 15.0}


def test_world1_is_a_noop():
    from mrgcn_amd import dist as D
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    assert D.env_world() == (1, 0, 0)
    assert D.max_over_ranks(3.5) == 3.5
    D.barrier()


# ---- node partition: host logic on CPU, collectives over gloo ----------------------
def test_node_partition_arithmetic_and_local_coo():
    import numpy as np
    from mrgcn_amd import synth
    from mrgcn_amd.partition import NodePartition
    g = synth.make_graph("aifb", seed=2, scale=0.2)
    N, R = g.num_nodes, g.num_relations
    world = 3
    parts = [NodePartition(N, world, r) for r in range(world)]
    assert parts[0].Np == parts[0].S * world >= N and sum(p.n_local for p in parts) == N
    seen = 0
    for p in parts:
        lr, lc, lv = p.local_coo(g.rows, g.cols, g.vals, R)
        seen += len(lr)
        # local column r*S + (j - j0) maps back to the global column r*N + j
        r_, jl = lc // p.S, lc % p.S
        assert jl.max(initial=0) < max(p.n_local, 1)
        back = r_ * N + jl + p.j0
        key = set(zip(lr.tolist(), back.tolist()))
        sel = (g.cols % N >= p.j0) & (g.cols % N < p.j1)
        assert key == set(zip(g.rows[sel].tolist(), g.cols[sel].tolist()))
    assert seen == g.nnz  # every entry belongs to exactly one rank
    # parameter / feature sharding round trip
    B, out = 3, 4
    w = torch.arange(B * N * out, dtype=torch.float32).view(B * N, out)
    cat = torch.cat([p.shard_weight_I(w, B).view(B, p.S, out)[:, :p.n_local] for p in parts], 1)
    cat_nm = torch.cat([p.shard_weight_I(w, B, node_major=True)[:p.n_local] for p in parts], 0)  # (N, B, out)
    assert torch.equal(cat_nm.permute(1, 0, 2).reshape(B * N, out), w)
    assert torch.equal(cat.reshape(B * N, out), w)


def _coll_worker(rank, world, port, out):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from mrgcn_amd import partition as P
    dist.init_process_group("gloo")
    S, F = 3, 2
    x = torch.full((world * S, F), float(rank + 1), requires_grad=True)
    y = P._ReduceScatterRows.apply(x, None)            # sum over ranks of my rows
    (y * (rank + 1)).sum().backward()                  # grad of my rows = rank + 1 -> all-gathered
    g = torch.arange(S * F, dtype=torch.float32).view(S, F) + 100 * rank
    out[rank] = (y.detach().tolist(), x.grad.tolist(), P.all_gather_rows(g).tolist(),
                 P.all_reduce_sum_(torch.tensor([float(rank + 1)])).tolist())
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_partition_collectives_gloo_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_coll_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    S, F = 3, 2
    for r in (0, 1):
        y, gx, ag, ar = out[r]
        assert y == [[3.0] * F] * S                     # 1 + 2 on every row
        assert gx == [[1.0] * F] * S + [[2.0] * F] * S  # backward = all-gather of the row grads
        assert ag == (torch.arange(S * F).view(S, F).float().tolist()
                      + (torch.arange(S * F).view(S, F).float() + 100).tolist())
        assert ar == [3.0]


def _hook_worker(rank, world, port, out):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from mrgcn_amd.partition import NodePartition, PartitionedRGCN
    dist.init_process_group("gloo")
    torch.manual_seed(0)
    mods = [(0, 6, "mrgcn", torch.nn.ReLU()), (6, 3, "mrgcn", None)]
    model = PartitionedRGCN(mods, 4, 10, 2, True, True, NodePartition(10, world, rank))
    finish = model.begin_overlapped_grad_reduce()
    # rank r contributes (r + 1) to every replicated gradient and (r + 1) * 10 to its own shard
    loss = sum((p * float(rank + 1)).sum() for p in model.replicated_parameters()) \
        + sum((p * float(10 * (rank + 1))).sum() for p in model.sharded_parameters())
    loss.backward()
    finish()
    rep = sorted({float(v) for p in model.replicated_parameters() for v in p.grad.flatten().tolist()})
    sh = sorted({float(v) for p in model.sharded_parameters() for v in p.grad.flatten().tolist()})
    # a second backward after finish() must not reduce again (hooks removed)
    for p in model.parameters():
        p.grad = None
    sum((p * 1.0).sum() for p in model.replicated_parameters()).backward()
    again = sorted({float(v) for p in model.replicated_parameters() for v in p.grad.flatten().tolist()})
    out[rank] = (rep, sh, again)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_overlapped_reduction_of_replicated_gradients_gloo_world2():
    """partition.py's backward: replicated gradients are summed over the ranks from post-accumulate hooks (on
    RCCL: asynchronously, under the rest of the backward), sharded ones stay local, and the hooks are gone after
    finish()."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_hook_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in (0, 1):
        rep, sh, again = out[r]
        assert rep == [3.0]                    # 1 + 2 on both ranks
        assert sh == [10.0 * (r + 1)]          # never communicated
        assert again == [1.0]


def _exchange_worker(rank, world, port, out):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from mrgcn_amd import partition as P
    dist.init_process_group("gloo")
    S, F = 7, 3
    # uneven live sets: rank 0 has three live rows, rank 1 none at all (a rank without a labelled node in reach)
    flags = torch.zeros(S, dtype=torch.uint8)
    if rank == 0:
        flags[[1, 4, 6]] = 1
    g = torch.arange(S * F, dtype=torch.float32).view(S, F) + 100 * rank
    g = g * flags[:, None].float()                       # rows outside the set hold zeros
    ex = P._live_row_exchange(flags, None)
    recv = P.all_gather_rows(g.index_select(0, ex["send_idx"]))
    compact = torch.zeros((world * S, F))
    compact.index_copy_(0, ex["dst"], recv.index_select(0, ex["src"]))
    dense = P.all_gather_rows(g)
    again = P._live_row_exchange(flags, None)            # kept on the flags tensor: no second round of collectives
    out[rank] = (compact.tolist(), dense.tolist(), ex["total"], ex["n_max"], again is ex)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_live_row_exchange_equals_the_dense_all_gather_gloo_world2():
    """partition._live_row_exchange: the backward all-gather that moves the rows with gradient only (padded to the
    largest count over the ranks) puts exactly the dense all-gather's rows where they belong — with uneven counts and a
    rank that has no live row."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_exchange_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in (0, 1):
        compact, dense, total, n_max, cached = out[r]
        assert compact == dense
        assert total == 3 and n_max == 3 and cached


# ---- row partition with operand-row halo exchange (mrgcn_amd.partition_halo): index maps + exchange over gloo -------
def _halo_worker(rank, world, port, out):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    import torch.distributed as dist
    from mrgcn_amd import synth
    from mrgcn_amd.partition import NodePartition
    from mrgcn_amd.partition_halo import _AllToAllRows, exchange_requests, halo_requests
    dist.init_process_group("gloo")
    g = synth.make_graph("aifb", seed=2, scale=0.2)
    N, R = g.num_nodes, g.num_relations
    part = NodePartition(N, world, rank)
    own, req = halo_requests(part, g.rows, g.cols, R)
    asked = exchange_requests(req, world, rank)
    # every column I am asked for is one I own; every column my rows read is either mine or requested from its owner
    for r, ids in asked.items():
        assert r != rank and ((ids % N) // part.S == rank).all()
    mine = (g.rows >= part.j0) & (g.rows < part.j1)
    need = np.unique(g.cols[mine])
    got = np.sort(np.concatenate([own] + list(req.values())))
    assert np.array_equal(got, need)
    # the exchange itself: "operand row" of literal column c = [c, 2c + 1]; the rows come back in request order, and
    # the gradient of a received row returns to the rank that sent it (each requested row once per requester)
    send_ids = np.concatenate([asked[r] for r in range(world) if r in asked] + [np.zeros(0, np.int64)])
    send = torch.tensor(np.stack([send_ids, 2 * send_ids + 1], 1), dtype=torch.float64, requires_grad=True)
    in_splits = [len(asked.get(r, ())) for r in range(world)]
    out_splits = [len(req.get(o, ())) for o in range(world)]
    recv = _AllToAllRows.apply(send, in_splits, out_splits, None)
    want = np.concatenate([req[o] for o in range(world) if o in req] + [np.zeros(0, np.int64)])
    assert np.array_equal(recv.detach().numpy()[:, 0].astype(np.int64), want)
    (recv * (rank + 1.0)).sum().backward()
    # my row sent to peer r comes back scaled by (r + 1)
    scale = np.concatenate([np.full(len(asked[r]), r + 1.0) for r in range(world) if r in asked] + [np.zeros(0)])
    assert np.allclose(send.grad.numpy(), np.stack([scale, scale], 1))
    out[rank] = (len(own), int(sum(out_splits)), int(sum(in_splits)))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 3])
def test_halo_requests_and_operand_row_exchange_gloo(world):
    import numpy as np
    from mrgcn_amd import synth
    from mrgcn_amd.partition_halo import choose_partition
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_halo_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = dict(out)
    assert sum(v[1] for v in res.values()) == sum(v[2] for v in res.values())   # rows requested == rows served
    g = synth.make_graph("aifb", seed=2, scale=0.2)
    ch = choose_partition(g.rows, g.cols, g.num_nodes, world, [16, 4])
    assert ch["halo_columns_per_rank"] == [res[r][1] for r in range(world)]
    assert ch["choice"] in ("halo", "column") and ch["halo"] == float(np.mean(ch["halo_columns_per_rank"])) * 20 * 4


# ---- the halo engine's exchange nodes: values over gloo (world 2) and the ORDER of the backward -----------------------
def _halo_order_worker(rank, world, port, out):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from mrgcn_amd import partition_halo as ph
    dist.init_process_group("gloo")
    log = []
    real_start = ph.start_rows_exchange

    def logged_start(send, i, o, group, pending):
        log.append("exchange started")
        return real_start(send, i, o, group, pending)

    class _Own(torch.autograd.Function):   # stands in for the product over the rank's own columns
        @staticmethod
        def forward(ctx, M):
            return M * 2.0

        @staticmethod
        def backward(ctx, g):
            log.append("own product backward")
            return g * 2.0

    real_wait = ph.wait_rows_exchange

    def logged_wait(p):
        log.append("exchange awaited")
        return real_wait(p)

    ph.start_rows_exchange, ph.wait_rows_exchange = logged_start, logged_wait
    try:
        # every rank sends its rows 0, 1 to the other rank (world 2) and receives two
        M = (torch.arange(8, dtype=torch.float64).view(4, 2) + 100.0 * rank).requires_grad_(True)
        pos = torch.tensor([0, 1]) if rank == 0 else torch.tensor([3, 1])
        splits_in = [0, 2] if rank == 0 else [2, 0]
        fwd, back = ph._Pending(), ph._Pending()
        send = ph._SendRows.apply(M, pos, back)
        ph.start_rows_exchange(send, splits_in, splits_in, None, fwd)
        Y = _Own.apply(M).sum()
        recv = ph._ExchangedRows.apply(send, splits_in, splits_in, None, fwd, back)
        del log[:]
        (Y + (recv * (rank + 1.0)).sum()).backward()
    finally:
        ph.start_rows_exchange, ph.wait_rows_exchange = real_start, real_wait
    out[rank] = (list(log), recv.detach().tolist(), M.grad.tolist())
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_halo_reverse_exchange_starts_before_the_own_products_backward():
    """`HaloPartitionedRGCN.forward` builds its graph so that autograd STARTS the reverse exchange of gradient rows, then
    runs the backward of the product over the rank's own columns, and only then waits for the rows and adds them into
    the operand's gradient (over RCCL: the collective is in flight under that product's backward).  Values: a row's
    gradient comes back from the rank that read it."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_halo_order_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = dict(out)
    for rank in (0, 1):
        log, recv, grad = res[rank]
        assert log == ["exchange started", "own product backward", "exchange awaited"], log
    # rank 0 received rank 1's rows 3 and 1, rank 1 received rank 0's rows 0 and 1
    assert res[0][1] == [[106.0, 107.0], [102.0, 103.0]] and res[1][1] == [[0.0, 1.0], [2.0, 3.0]]
    # d/dM: 2 everywhere (the own product) + the reader's scale on the rows it read (rank 1 scales by 2, rank 0 by 1)
    assert res[0][2] == [[4.0, 4.0], [4.0, 4.0], [2.0, 2.0], [2.0, 2.0]]
    assert res[1][2] == [[2.0, 2.0], [3.0, 3.0], [2.0, 2.0], [3.0, 3.0]]
