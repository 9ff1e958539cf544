#!/usr/bin/env python
"""Times the individual C-ABI calls of one AM-shaped layer (HIP events on the stream), for
A/B work on single kernels.   python tools/kernel_probe.py [--which mix_fwd,mix_bwd,...]"""
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
    ap.add_argument("--which", default="mix_fwd,mix_fwd_add,mix_bwd,xf_fwd0,xf_fwd1,xf_bwd0,xf_bwd1,spmm,spmm_t10,spmm_t11,adam")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--ldm", type=int, default=16)
    ap.add_argument("--dy-zero-frac", type=float, default=0.997,
                    help="fraction of dY rows set to zeros for spmm_tl10 (the AM epoch: 0.997 / 0.9994)")
    ap.add_argument("--dm-zero-frac", type=float, default=0.0,
                    help="fraction of dM rows set to exact zeros (the AM epoch has ~0.9)")
    ap.add_argument("--x-ld", type=int, default=0, help="row stride of X in floats (default: its width)")
    ap.add_argument("--sweep", default="", help="library configurations to time every call under, ';' separated sets "
                    "of k=v pairs: 'xform_hot=0;xform_hot=1,xform_hot_tile=96'")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.load()
    sh = synth.SHAPES[a.workload]
    g = synth.make_graph(a.workload, seed=0, scale=a.scale)
    N, R, B, F, K = g.num_nodes, g.num_relations, sh["bases"], sh["hidden"], sh["x_width"]
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals),
                                (N, R * N)).to(dev)
    plan = GraphPlan(A, N, R)
    s = torch.cuda.current_stream(dev).cuda_stream
    h = plan.handle
    nc = plan.ncols
    ld = a.ldm
    V = torch.randn((N, B, F), device=dev)   # node-major basis table
    comp = torch.randn((R, B), device=dev)
    M = torch.empty((nc, ld), device=dev)
    M2 = torch.randn((nc, 12), device=dev)
    dM = torch.randn((nc, 12), device=dev)
    if a.dm_zero_frac > 0:
        dM[torch.rand(nc, device=dev) < a.dm_zero_frac] = 0
    dV = torch.empty_like(V)
    dcomp = torch.empty_like(comp)
    ldx = a.x_ld if a.x_ld >= K else K
    X = torch.randn((N, ldx), device=dev)[:, :K]     # row stride ldx (e.g. 160: whole 128-byte lines at K = 155)
    W0 = torch.randn((R, K, F), device=dev)
    H = torch.randn((N, F), device=dev)
    C = sh["classes"]
    W1 = torch.randn((R, F, C), device=dev)
    dW0, dW1 = torch.empty_like(W0), torch.empty_like(W1)
    dH = torch.empty_like(H)
    dM1 = torch.randn((nc, 12), device=dev)
    Y = torch.empty((N, F), device=dev)
    dY10 = torch.randn((N, 10), device=dev)
    dY11 = torch.randn((N, 11), device=dev)
    nws = max(int(lib.mrgcn_rel_transform_bwd_workspace(h, K, F, 0, 1)),
              int(lib.mrgcn_rel_transform_bwd_workspace(h, F, C, 1, 1)))
    ws = torch.empty((nws,), device=dev)
    P, G_, M_, V_ = (torch.randn((B * N * F,), device=dev) for _ in range(4))
    coef = torch.ones((), device=dev)
    dYz = dY10.clone()
    if a.dy_zero_frac > 0:
        dYz[torch.rand(N, device=dev) < a.dy_zero_frac] = 0
    scratch = torch.empty(int(lib.mrgcn_spmm_transposed_live_scratch(h)), dtype=torch.uint8, device=dev)
    clive = torch.ones(nc, dtype=torch.uint8, device=dev)
    if a.dm_zero_frac > 0:
        clive = (dM.abs().sum(1) > 0).to(torch.uint8)
    ncur = torch.zeros(N, dtype=torch.uint8, device=dev)    # row-sparse gradient: which node blocks were written
    never = torch.zeros(N, dtype=torch.uint8, device=dev)
    sq = torch.zeros((), device=dev, dtype=torch.float64)

    def chk(rc):
        L.check(rc)

    # the bf16 pipeline's operands (X rows of 160 bf16, feature-term rows of 16 bf16, M rows of 10 bf16)
    ldXb = (K + 7) // 8 * 8
    Xb = torch.zeros((N, ldXb), dtype=torch.bfloat16, device=dev)
    M2b = torch.randn((nc, 16), device=dev).to(torch.bfloat16)
    Mb = torch.empty((plan.nop, F), dtype=torch.bfloat16, device=dev)
    M2f = torch.empty((nc, 12), device=dev)
    chk(lib.mrgcn_cast_rows_bf16(X.data_ptr(), ldx, N, K, Xb.data_ptr(), ldXb, s))
    calls = {
        "cast_x": lambda: chk(lib.mrgcn_cast_rows_bf16(X.data_ptr(), ldx, N, K, Xb.data_ptr(), ldXb, s)),
        "xf_fwd0_xb": lambda: chk(lib.mrgcn_rel_transform_fwd_xbf16(h, Xb.data_ptr(), ldXb, K, W0.data_ptr(), F, M2b.data_ptr(), 16, 0, 1, s)),
        "xf_fwd0_xb_f32": lambda: chk(lib.mrgcn_rel_transform_fwd_xbf16(h, Xb.data_ptr(), ldXb, K, W0.data_ptr(), F, M2f.data_ptr(), 12, 0, 0, s)),
        "mix_fwd_addb": lambda: chk(lib.mrgcn_basis_mix_fwd_abf16(h, V.data_ptr(), comp.data_ptr(), B, F, M2b.data_ptr(), 16, Mb.data_ptr(), F, 1, s)),
        "mix_fwd_add_b": lambda: chk(lib.mrgcn_basis_mix_fwd_bf16(h, V.data_ptr(), comp.data_ptr(), B, F, M2.data_ptr(), 12, Mb.data_ptr(), F, s)),
        "mix_fwd": lambda: chk(lib.mrgcn_basis_mix_fwd_f32(h, V.data_ptr(), comp.data_ptr(), B, F, 0, 0, M.data_ptr(), ld, s)),
        "mix_fwd_add": lambda: chk(lib.mrgcn_basis_mix_fwd_f32(h, V.data_ptr(), comp.data_ptr(), B, F, M2.data_ptr(), 12, M.data_ptr(), ld, s)),
        "mix_bwd": lambda: chk(lib.mrgcn_basis_mix_bwd_f32(h, dM.data_ptr(), 12, 0, V.data_ptr(), comp.data_ptr(), B, F, dV.data_ptr(), 0, dcomp.data_ptr(), 0, s)),
        "mix_bwd_rows": lambda: chk(lib.mrgcn_basis_mix_bwd_f32(h, dM.data_ptr(), 12, clive.data_ptr(), V.data_ptr(), comp.data_ptr(), B, F, dV.data_ptr(), ncur.data_ptr(), dcomp.data_ptr(), sq.data_ptr(), s)),
        "mix_bwd_norm": lambda: chk(lib.mrgcn_basis_mix_bwd_f32(h, dM.data_ptr(), 12, clive.data_ptr(), V.data_ptr(), comp.data_ptr(), B, F, 0, ncur.data_ptr(), dcomp.data_ptr(), sq.data_ptr(), s)),
        "adam_rows": lambda: chk(lib.mrgcn_adam_step_rows_f32(P.data_ptr(), dV.data_ptr(), M_.data_ptr(), V_.abs_().data_ptr(), N, B * F, ncur.data_ptr(), never.data_ptr(), 0.01, 0.9, 0.999, 1e-8, 1, 0, coef.data_ptr(), s)),
        "spmm_tl10": lambda: chk(lib.mrgcn_spmm_transposed_live_f32(h, dYz.data_ptr(), 10, 10, dM.data_ptr(), 12, scratch.data_ptr(), clive.data_ptr(), 0, 1, s)),
        "spmm_tl10_nd": lambda: chk(lib.mrgcn_spmm_transposed_live_f32(h, dYz.data_ptr(), 10, 10, dM.data_ptr(), 12, scratch.data_ptr(), clive.data_ptr(), 0, 0, s)),
        "xf_fwd0": lambda: chk(lib.mrgcn_rel_transform_fwd_f32(h, X.data_ptr(), ldx, K, W0.data_ptr(), F, M2.data_ptr(), 12, 0, s)),
        "xf_fwd1": lambda: chk(lib.mrgcn_rel_transform_fwd_f32(h, H.data_ptr(), F, F, W1.data_ptr(), C, M.data_ptr(), ld, 1, s)),
        "xf_bwd0": lambda: chk(lib.mrgcn_rel_transform_bwd_f32(h, dM.data_ptr(), 12, X.data_ptr(), ldx, K, W0.data_ptr(), F, 0, K, dW0.data_ptr(), ws.data_ptr(), nws, s)),
        "xf_bwd1": lambda: chk(lib.mrgcn_rel_transform_bwd_f32(h, dM1.data_ptr(), 12, H.data_ptr(), F, F, W1.data_ptr(), C, dH.data_ptr(), F, dW1.data_ptr(), ws.data_ptr(), nws, s)),
        "spmm": lambda: plan.spmm(L.VIEW_COMPACT, M, F=F, out=Y),
        "spmm_t10": lambda: plan.spmm(L.VIEW_TRANSPOSED, dY10, F=10, out=dM),
        "spmm_t11": lambda: plan.spmm(L.VIEW_TRANSPOSED, dY11, F=11, out=dM),
        "adam": lambda: chk(lib.mrgcn_adam_step_f32(P.data_ptr(), G_.data_ptr(), M_.data_ptr(), V_.data_ptr(), P.numel(), 0.01, 0.9, 0.999, 1e-8, 0.0, 1, coef.data_ptr(), s)),
    }
    print(f"N={N} R={R} B={B} F={F} K={K} ncols={nc} nnz={plan.nnz} ldM={ld}")
    sweeps = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in part.split(",") if kv)
              for part in a.sweep.split(";")] if a.sweep else [{}]
    for cfg in sweeps:
        old = L.set_config(**cfg)
        for name in a.which.split(","):
            ms = event_time_ms(calls[name], a.iters, s)
            print(f"{name:12s} {ms*1e3:9.1f} us   {cfg if cfg else ''}", flush=True)
        L.set_config(**old)


if __name__ == "__main__":
    main()
