#!/usr/bin/env python
"""Times the DistMult rank kernel at BASELINE config 4's decoder size (FB15k-237: 14 541 nodes,
237 relations, h = 200).   python tools/lp_probe.py [--facts 500] [--cpu-facts 50]"""
import argparse
import json
import time

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mrgcn_amd.tasks import link_prediction as lp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=14541)
    ap.add_argument("--rels", type=int, default=237)
    ap.add_argument("--hidden", type=int, default=200)
    ap.add_argument("--facts", type=int, default=500)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cpu-facts", type=int, default=50)
    a = ap.parse_args()
    N, P, H, nf = a.nodes, a.rels, a.hidden, a.facts
    gen = torch.Generator(device="cuda").manual_seed(0)
    E = torch.relu(torch.randn((N, H), device="cuda", generator=gen))
    Rel = torch.randn((2 * P + 1, H), device="cuda", generator=gen)
    rng = np.random.default_rng(0)
    facts = np.stack([rng.integers(0, N, nf), rng.integers(0, P, nf), rng.integers(0, N, nf)], 1)
    out = {"nodes": N, "hidden": H, "facts": nf}
    for filtered in (False, True):
        lp.compute_ranks_fast(facts, E, Rel, filtered=filtered)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            lp.compute_ranks_fast(facts, E, Rel, filtered=filtered)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.iters * 1e3
        out["flt_ms" if filtered else "raw_ms"] = round(ms, 3)
    # useful arithmetic: 2 directions x nf x N x H x (2.5 mul + 1 add)
    out["gflop_per_call"] = round(2 * nf * N * H * 3.5 / 1e9, 2)
    out["tflops_raw"] = round(out["gflop_per_call"] / out["raw_ms"], 2)
    if a.cpu_facts > 0:
        # the reference's op sequence on the host for a slice of the same facts
        # (link_prediction.py:593-643: candidate tensor, broadcast score, compare)
        Ec, Rc = E.cpu(), Rel.cpu()
        f = torch.from_numpy(facts[: a.cpu_facts]).long()
        t0 = time.perf_counter()
        for head in (False, True):
            cand = torch.arange(N).unsqueeze(0).expand(len(f), N)
            s = Ec[cand] if head else Ec[f[:, 0]].unsqueeze(1)
            o = Ec[f[:, 2]].unsqueeze(1) if head else Ec[cand]
            sc = torch.sum(s * Rc[f[:, 1]].unsqueeze(1) * o, dim=-1)
            true = sc[torch.arange(len(f)), f[:, 0] if head else f[:, 2]].unsqueeze(1)
            _ = torch.sum(sc > true, 1) + torch.round((torch.sum(sc == true, 1) - 1) / 2).long() + 1
        cpu_ms = (time.perf_counter() - t0) * 1e3
        out["cpu_ms_per_fact"] = round(cpu_ms / a.cpu_facts, 3)
        out["cpu_threads"] = torch.get_num_threads()
        out["gpu_ms_per_fact"] = round(out["raw_ms"] / nf, 5)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
