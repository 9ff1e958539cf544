#!/usr/bin/env python
"""LAB driver: the stacked-CSR product with the hottest operand rows staged in LDS (tools/lab/spmm_hot_lab.hip) on the
AM-shaped graph — the A/B round 5's verdict asked for.

  python tools/lab/spmm_hot_lab.py [--H 0 800 1600 3400] [--F 10]

The operand is re-ordered by READER COUNT (most-read rows first; the plan's own order keeps columns of >= 16 readers in
a dense region but not sorted), the short rows (<= 8 entries: 1.5 M of the 1.67 M rows) are walked by one persistent
1 024-thread workgroup per CU, and H rows live in LDS.  H = 0 runs the same kernel without the staging.  Printed: the
share of the short rows' entries the H rows serve, the time of the S-row pass, and — for scale — the production kernel
over ALL rows.  Every variant is checked against a float64 product."""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import event_time_ms  # noqa: E402
from mrgcn_amd import _lib as L  # noqa: E402
from mrgcn_amd import synth  # noqa: E402
from mrgcn_amd.plan import GraphPlan  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--F", type=int, default=10)
    ap.add_argument("--H", type=int, nargs="*", default=[0, 800, 1600, 3400])
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "libspmm_hot_lab.so")
    src = os.path.join(here, "spmm_hot_lab.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", src, "-o", so], check=True)
    lab = C.CDLL(so)
    lab.lab_hot.restype = C.c_int
    lab.lab_hot.argtypes = [C.c_int64] + [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    dev = torch.device("cuda:0")
    g = synth.make_graph("am", seed=0, scale=a.scale)
    N, R, F = g.num_nodes, g.num_relations, a.F
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals), (N, R * N)).to(dev)
    plan = GraphPlan(A, N, R, operand_row_bytes=[4 * F])
    ptr3 = plan.export(L.ARR_PTR3).astype(np.int64)
    mcol = plan.export(L.ARR_MCOL).astype(np.int64)
    mval = plan.export(L.ARR_MVAL).astype(np.float32)
    rowmap = plan.export(L.ARR_ROWMAP).astype(np.int64)
    lens = np.diff(ptr3)
    n_short = int((lens <= 8).sum())
    assert (lens[:n_short] <= 8).all() and (n_short == len(lens) or lens[n_short] > 8), "ranks are class-major"
    e_short = int(ptr3[n_short])
    print(f"N={N} R={R} nnz={plan.nnz} operand rows={plan.nop}; short rows {n_short} ({n_short / N:.1%}) hold "
          f"{e_short} entries ({e_short / plan.nnz:.1%})")
    # operand order by reader count (all rows' readers), descending
    readers = np.bincount(mcol, minlength=plan.nop)
    order = np.argsort(-readers, kind="stable")
    newpos = np.empty(plan.nop, dtype=np.int64)
    newpos[order] = np.arange(plan.nop)
    idx2 = newpos[mcol].astype(np.int32)
    M0 = torch.randn((plan.nop, F), device=dev)
    M_re = M0[torch.from_numpy(order).to(dev)].contiguous()       # row p of M_re = old row order[p]
    s = torch.cuda.current_stream(dev).cuda_stream
    d_ptr = torch.from_numpy(ptr3[:n_short + 1].astype(np.int32)).to(dev)
    d_idx = torch.from_numpy(idx2).to(dev)
    d_val = torch.from_numpy(mval).to(dev)
    d_map = torch.from_numpy(rowmap.astype(np.int32)).to(dev)
    # float64 reference of the short rows
    rows_of_entry = np.repeat(np.arange(n_short), lens[:n_short])
    Mh = M_re.double().cpu().numpy()
    ref = np.zeros((N, F))
    np.add.at(ref, rowmap[rows_of_entry], mval[:e_short, None].astype(np.float64) * Mh[idx2[:e_short]])
    short_out = rowmap[:n_short]
    bytes_alg = e_short * 8 + (n_short + 1) * 4 + len(np.unique(idx2[:e_short])) * F * 4 + n_short * F * 4
    for H in a.H:
        Y = torch.zeros((N, 12), device=dev)
        def call():
            rc = lab.lab_hot(n_short, d_ptr.data_ptr(), d_idx.data_ptr(), d_val.data_ptr(), d_map.data_ptr(),
                             M_re.data_ptr(), F, F, H, Y.data_ptr(), 12, 256, s)
            assert rc == 0, rc
        call()
        torch.cuda.synchronize()
        err = float(np.abs(Y[:, :F].double().cpu().numpy()[short_out] - ref[short_out]).max())
        ms = event_time_ms(call, a.iters, s)
        share = float((idx2[:e_short] < H).mean())
        print(f"H={H:5d} ({H * 48 / 1024:6.1f} KB LDS): hot rows serve {share:6.2%} of the short rows' entries   "
              f"S-row pass {ms * 1e3:7.1f} us   {bytes_alg / (ms * 1e-3) / 1e9:6.0f} GB/s algorithmic   max err {err:.1e}",
              flush=True)
    # scale: the production kernel over all rows, and over the same re-ordered operand
    Yp = torch.empty((N, 12), device=dev)[:, :F]
    t = event_time_ms(lambda: plan.spmm(L.VIEW_COMPACT, M0, F=F, out=Yp, pad_writable=True), a.iters, s)
    print(f"production k_spmm3 over ALL rows (plan's operand order): {t * 1e3:.1f} us")


if __name__ == "__main__":
    main()
