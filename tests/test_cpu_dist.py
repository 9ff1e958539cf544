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
