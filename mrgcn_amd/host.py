"""Host-side housekeeping for measurement scripts and training loops: the CPU quota of the container.

The bench boxes give a container a CPU quota (cgroup v2 `cpu.max`: 16 CPUs of a 256-thread host).  torch sizes its
intra-op pool by the HOST's thread count: every parallel CPU op (an `arange` of a million elements, a sort, a CPU-side
`isin`) then wakes 256 spinning workers, the quota of the 100 ms scheduling period is used up within a few ms and the
kernel parks the WHOLE process — also the thread that launches GPU kernels — for the rest of the period
(`/sys/fs/cgroup/cpu.stat`: nr_throttled).  Round 6 found this as 35-60 ms stalls in every second or third eager step of
the full-multimodal model.  `fit_cpu_pool_to_quota()` sizes the pool to the quota; `bench.py` and the probes under
`tools/` call it first, a training script on such a box should too."""
from __future__ import annotations


def cpu_quota():
    """CPUs this container may use per scheduling period (cgroup v2 cpu.max), or None when unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:  # noqa: BLE001  (no cgroup v2 file: no quota known)
        return None


def fit_cpu_pool_to_quota():
    """torch.set_num_threads(min(current, quota)); returns the quota (None: nothing done)."""
    q = cpu_quota()
    if q:
        import torch
        torch.set_num_threads(max(1, min(torch.get_num_threads(), int(q))))
    return q
