#!/usr/bin/env python
"""LAB: do the layer-0 transform (gathers served by the Infinity Cache) and the node-table Adam pass (HBM streams) run
beside each other?  Times each alone and both on two streams (AM shape; the Adam stand-in is the dense kernel over as
many bytes as the epoch's row Adam moves: 8 GB).

    python tools/lab/overlap_probe.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mrgcn_amd import _lib as L  # noqa: E402
from mrgcn_amd import synth  # noqa: E402
from mrgcn_amd.plan import GraphPlan  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = L.load()
    sh = synth.SHAPES["am"]
    g = synth.make_graph("am", seed=0)
    N, R, B, F, K = g.num_nodes, g.num_relations, sh["bases"], sh["hidden"], sh["x_width"]
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals), (N, R * N)).to(dev)
    plan = GraphPlan(A, N, R)
    h = plan.handle
    X = torch.randn((N, K), device=dev)
    W0 = torch.randn((R, K, F), device=dev)
    M2 = torch.empty((plan.ncols, 12), device=dev)
    n_adam = 828598 * B * F      # the live blocks of the epoch
    P, G_, M_, V_ = (torch.randn((n_adam,), device=dev) for _ in range(4))
    V_.abs_()
    coef = torch.ones((), device=dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    V = torch.randn((N, B, F), device=dev)
    comp = torch.randn((R, B), device=dev)
    M = torch.empty((plan.ncols, 10), device=dev)

    def xform(s):
        L.check(lib.mrgcn_rel_transform_fwd_f32(h, X.data_ptr(), K, K, W0.data_ptr(), F, M2.data_ptr(), 12, 0, s.cuda_stream))

    def adam(s):
        L.check(lib.mrgcn_adam_step_f32(P.data_ptr(), G_.data_ptr(), M_.data_ptr(), V_.data_ptr(), n_adam, 0.01, 0.9, 0.999,
                                        1e-8, 0.0, 1, coef.data_ptr(), s.cuda_stream))

    def mix(s):
        L.check(lib.mrgcn_basis_mix_fwd_f32(h, V.data_ptr(), comp.data_ptr(), B, F, M2.data_ptr(), 12, M.data_ptr(), 10,
                                            s.cuda_stream))

    def timed(fn, iters=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / iters * 1e6

    def both(a, b):
        def run():
            s2.wait_stream(s1)
            a(s1)
            b(s2)
            s1.wait_stream(s2)
        return run

    for name, a, b in (("transform | adam", xform, adam), ("mix | adam", mix, adam), ("transform | mix", xform, mix)):
        ta = timed(lambda: a(s1))
        tb = timed(lambda: b(s1))
        tc = timed(both(a, b))
        print(f"{name}: alone {ta:.0f} + {tb:.0f} = {ta + tb:.0f} us, on two streams {tc:.0f} us "
              f"({(ta + tb - tc) / (ta + tb) * 100:.0f} % saved)", flush=True)


if __name__ == "__main__":
    main()
