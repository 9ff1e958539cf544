#!/usr/bin/env python
"""Runs only the stacked-CSR SpMM on the AM-shaped graph (for rocprofv3 --pmc passes and
quick A/B timing of operand layouts).

    python tools/spmm_probe.py [--ld 16] [--view compact|literal|transposed] [--iters 20]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import event_time_ms  # noqa: E402
from mrgcn_amd import _lib as L  # noqa: E402
from mrgcn_amd import synth  # noqa: E402
from mrgcn_amd.plan import GraphPlan  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="am")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--ld", type=int, nargs="+", default=[10])
    ap.add_argument("--F", type=int, default=10)
    ap.add_argument("--view", default="compact")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--prune", action="store_true")
    ap.add_argument("--value-mode", default="norm_f32")
    ap.add_argument("--ldy", type=int, default=0, help="leading dimension of the compact product's output (0 = F rounded up to 4, pad writable; F = dense rows)")
    ap.add_argument("--replicate", type=int, default=-1, help="1 / 0: operand replicas on / off (default: library)")
    ap.add_argument("--row-bytes", type=int, nargs="*", default=None, help="operand row sizes the plan's operand order is "
                    "built for (GraphPlan operand_row_bytes; default: 4 * the first --ld)")
    ap.add_argument("--ab-two-pass", type=int, default=0, help="compact view: alternate N rounds of the two-pass "
                    "form (MRGCN_SPMM_TWO_PASS) and the default in-kernel finalize, in this process")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = synth.make_graph(a.workload, seed=0, scale=a.scale, value_mode=a.value_mode)
    N, R = g.num_nodes, g.num_relations
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals),
                                (N, R * N)).to(dev)
    plan = GraphPlan(A, N, R, prune_zeros=a.prune, replicate=None if a.replicate < 0 else bool(a.replicate),
                     operand_row_bytes=a.row_bytes if a.row_bytes is not None else [4 * a.ld[0]])
    stream = torch.cuda.current_stream(dev).cuda_stream
    F = a.F
    alg = plan.spmm_bytes(F)
    print(f"N={N} R={R} nnz={plan.nnz} ncols={plan.ncols} operand_rows={plan.nop} replicas={plan.n_rep} long_rows={plan.long_rows} "
          f"long_cols={plan.long_cols} alg_bytes={alg}")
    for ld in a.ld:
        if a.view == "compact":
            D = torch.randn((plan.nop * ld + 8,), device=dev)[:plan.nop * ld].view(plan.nop, ld)
            ldy = a.ldy if a.ldy else (F + 3) // 4 * 4   # default: the layer's own layout (rows padded to 16 bytes)
            Y = torch.empty((N, max(ldy, F)), device=dev)[:, :F]
            fn = lambda: plan.spmm(L.VIEW_COMPACT, D, F=F, out=Y, pad_writable=ldy > F)  # noqa: E731
            fn2 = lambda: plan.spmm(L.VIEW_COMPACT, D, F=F, out=Y, pad_writable=ldy > F, two_pass=True)  # noqa: E731
            for r in range(a.ab_two_pass):
                t2 = event_time_ms(fn2, a.iters, stream)
                t1 = event_time_ms(fn, a.iters, stream)
                print(f"round {r}: two-pass {t2*1e3:.1f} us, in-kernel finalize {t1*1e3:.1f} us")
            if plan.n_rep:
                ms_r = event_time_ms(lambda: plan.replicate(D), a.iters, stream)
                print(f"replicate ld={ld}: {ms_r*1e3:.1f} us for {plan.n_rep} rows")
        elif a.view == "literal":
            D = torch.randn((R * N, ld), device=dev)
            Y = torch.empty((N, F), device=dev)
            fn = lambda: plan.spmm(L.VIEW_LITERAL, D, F=F, out=Y)  # noqa: E731
        else:
            D = torch.randn((N, ld), device=dev)
            Y = torch.empty((plan.ncols, 16), device=dev)
            fn = lambda: plan.spmm(L.VIEW_TRANSPOSED, D, F=F, out=Y)  # noqa: E731
        ms = event_time_ms(fn, a.iters, stream)
        print(f"view={a.view} F={F} ld={ld}: {ms*1e3:.1f} us  {alg/ms/1e6:.1f} GB/s algorithmic "
              f"({alg/ms/1e6/8000*100:.1f}% of 8 TB/s)")
        del D, Y


if __name__ == "__main__":
    main()
