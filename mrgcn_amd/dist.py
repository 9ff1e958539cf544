"""Process-group glue shared by both multi-GPU modes, over torch.distributed (RCCL on GPUs, gloo in
the CPU tests): process-group start-up from the torchrun environment, the barrier and the max-over-ranks around a
timed region.  Replicas (`bench.py --gpus N` on a shape that fits one GPU) use nothing else; the node-partitioned
engine (`mrgcn_amd/partition.py`: reduce-scatter of a layer's output rows in the forward, all-gather of the rows with
gradient in the backward, all-reduce of the small replicated gradients) issues its collectives itself on the group
this module initialises."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str | None = None, device: torch.device | None = None):
    """Initialises the default process group from the torchrun environment (no-op at world 1)."""
    world, rank, local_rank = env_world()
    if world <= 1 or dist.is_initialized():
        return world, rank, local_rank
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)
    return world, rank, local_rank


def barrier(device: torch.device | None = None):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device: torch.device | None = None) -> float:
    """MAX all-reduce of a host scalar (the timed region of bench.py)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    on = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def replica_value(ms_per_step_local: float, device: torch.device | None = None) -> float:
    """Replicas: every rank runs the same epoch; the job's epoch time is the slowest rank's."""
    return max_over_ranks(ms_per_step_local, device)
