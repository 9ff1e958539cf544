#!/usr/bin/env python
"""LAB driver: times layouts of the compact operand for the stacked-CSR product on the AM-shaped graph.

  python tools/lab/spmm_lab.py [--scale 1.0] [--H 2 4 16 64 1000000000]

1. the production kernel as it is, then with its index array replaced by (a) perfectly sequential
   operand rows and (b) 1024 cached rows — what its structure costs without the random reads;
2. the stream + gather form (spmm_lab.hip) with gather-region threshold H: columns read by >= H rows
   live in the dense gather region, every other ENTRY has its own stream row (H = 2: nothing is
   replicated, single-reader columns stream; H large: everything streams).
Every variant is checked against a float64 index_add."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import event_time_ms  # noqa: E402
from mrgcn_amd import _lib as L  # noqa: E402
from mrgcn_amd import synth  # noqa: E402
from mrgcn_amd.plan import GraphPlan  # noqa: E402

_p, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64


def load_lab():
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libspmm_lab.so"))
    lib.lab_sg.restype = C.c_int
    lib.lab_sg.argtypes = [_i32, _i32, _i64] + [_p] * 8 + [_i32] + [_p] * 6 + [_i32, _p, _p, _i32, _p]
    lib.lab_rows_p.restype = C.c_int
    lib.lab_rows_p.argtypes = [_i32, _i32, _i32, _i32, _i64, _p, _p, _p, _p, _i64, _i32, _p, _i32, _p, _p, _p, _p,
                               _i32, _p, _p, _i32, _p]
    lib.lab_v3.restype = C.c_int
    lib.lab_v3.argtypes = [_i64, _p, _p, _p, _p, _i64, _i32, _p, _i32, _p, _p, _p, _p, _i32, _p, _p, _i32, _p, _i32,
                           _i32, _p]
    lib.lab_seg.restype = C.c_int
    lib.lab_seg.argtypes = [_i32, _i32, _i64, _p, _p, _p, _p, _i64, _i32, _p, _i64, _p, _p, _i32, _p]
    lib.lab_copy_i32.restype = C.c_int
    lib.lab_copy_i32.argtypes = [_p, _p, _i64, _p]
    return lib


def dev_i32(plan, which, dev):
    return torch.from_numpy(plan.export(which).astype(np.int64)).to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="am")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--F", type=int, default=10)
    ap.add_argument("--H", type=int, nargs="+", default=[2, 4, 16, 64, 10 ** 9])
    ap.add_argument("--su", type=int, nargs="+", default=[4])
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--seg", type=int, default=1)
    ap.add_argument("--pers", type=int, default=1)
    ap.add_argument("--v3", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=128)
    ap.add_argument("--sg", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lab = load_lab()
    g = synth.make_graph(a.workload, seed=0, scale=a.scale)
    N, R, F = g.num_nodes, g.num_relations, a.F
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals),
                                (N, R * N)).to(dev)
    plan = GraphPlan(A, N, R)
    del A
    st = torch.cuda.current_stream(dev).cuda_stream
    alg = plan.spmm_bytes(F)
    print(f"N={N} R={R} nnz={plan.nnz} ncols={plan.ncols} alg_bytes={alg}", flush=True)

    def report(name, ms, extra=""):
        print(f"{name:48s} {ms * 1e3:8.1f} us  {alg / ms / 1e6:8.1f} GB/s  {alg / ms / 1e6 / 80:5.1f} %  {extra}",
              flush=True)

    # entries in row-major order, compact column of each, values
    rowidx = dev_i32(plan, L.ARR_ROWIDX, dev)
    ccol = dev_i32(plan, L.ARR_CCOL, dev)
    val = torch.from_numpy(plan.export(L.ARR_VAL)).to(dev)
    cptr = dev_i32(plan, L.ARR_CPTR, dev)
    cnt = cptr[1:] - cptr[:-1]
    mpos = dev_i32(plan, L.ARR_MPOS, dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    Mc = torch.randn((plan.ncols, F), device=dev, generator=gen)
    Yref = torch.zeros((N, F), dtype=torch.float64, device=dev)
    Yref.index_add_(0, rowidx, val.double()[:, None] * Mc.double()[ccol])
    tol = 1e-4 * (1.0 + float(Yref.abs().max()))

    def check(Y, name):
        err = float((Y.double() - Yref).abs().max())
        assert err <= tol, (name, err, tol)
        return err

    # ---- 1. production kernel ---------------------------------------------------------------
    for ld in (12, 16):
        M = torch.zeros((plan.ncols * ld + 8,), device=dev)[:plan.ncols * ld].view(plan.ncols, ld)
        M[mpos, :F] = Mc
        Y = torch.empty((N, F), device=dev)
        ms = event_time_ms(lambda: plan.spmm(L.VIEW_COMPACT, M, F=F, out=Y), a.iters, st)
        check(Y, "prod")
        report(f"production k_spmm, ld={ld}", ms)
    ld = 12
    M = torch.zeros((plan.ncols * ld + 8,), device=dev)[:plan.ncols * ld].view(plan.ncols, ld)
    M[mpos, :F] = Mc
    mcol_ptr, nn = plan.array_ptr(L.ARR_MCOL)
    saved = torch.empty(nn, dtype=torch.int32, device=dev)
    assert lab.lab_copy_i32(saved.data_ptr(), mcol_ptr, nn, st) == 0
    seq = (torch.arange(nn, device=dev, dtype=torch.int64) % plan.ncols).to(torch.int32)
    assert lab.lab_copy_i32(mcol_ptr, seq.data_ptr(), nn, st) == 0
    ms = event_time_ms(lambda: plan.spmm(L.VIEW_COMPACT, M, F=F, out=Y), a.iters, st)
    report("production, operand rows = entry id (sequential)", ms, "no random reads, no re-reads")
    cached = (torch.arange(nn, device=dev, dtype=torch.int64) % 1024).to(torch.int32)
    assert lab.lab_copy_i32(mcol_ptr, cached.data_ptr(), nn, st) == 0
    ms = event_time_ms(lambda: plan.spmm(L.VIEW_COMPACT, M, F=F, out=Y), a.iters, st)
    report("production, operand rows = entry id % 1024 (cached)", ms, "index/value streams + machinery only")
    assert lab.lab_copy_i32(mcol_ptr, saved.data_ptr(), nn, st) == 0
    ms = event_time_ms(lambda: plan.spmm(L.VIEW_COMPACT, M, F=F, out=Y), a.iters, st)
    check(Y, "prod-restored")
    report("production, restored", ms)
    del M, seq, cached, saved

    # ---- 5. v3: one gather batch per wave -------------------------------------------------------------
    if a.v3:
        mcol = torch.from_numpy(plan.export(L.ARR_MCOL)).to(dev)
        mval = torch.from_numpy(plan.export(L.ARR_MVAL)).to(dev)
        rowptr = torch.from_numpy(plan.export(L.ARR_ROWPTR)).to(dev)
        rp64 = rowptr.long()
        lens = rp64[1:] - rp64[:-1]
        CH = a.chunk
        long_rows = torch.nonzero(lens > 32).flatten()
        nch = (lens[long_rows] + CH - 1) // CH
        n_long, n_chunks = len(long_rows), int(nch.sum())
        lcptr = torch.zeros(n_long + 1, dtype=torch.int64, device=dev)
        lcptr[1:] = torch.cumsum(nch, 0)
        crl = torch.repeat_interleave(torch.arange(n_long, device=dev), nch)
        k = torch.arange(n_chunks, device=dev) - lcptr[crl]
        r = long_rows[crl]
        c_beg = (rp64[r] + k * CH).to(torch.int32)
        c_end = torch.minimum(rp64[r] + (k + 1) * CH, rp64[r + 1]).to(torch.int32)
        c_row = torch.where(nch[crl] == 1, r, -(r + 2)).to(torch.int32)
        long_row32, lcptr32 = long_rows.to(torch.int32), lcptr.to(torch.int32)
        mid = torch.nonzero((lens > 8) & (lens <= 32)).flatten().to(torch.int32)
        partials = torch.zeros((max(n_chunks, 1), 16), device=dev)
        print(f"v3: rows<=8 {int((lens <= 8).sum())}, mid {len(mid)}, long {n_long}, chunks {n_chunks}; entries in "
              f"rows<=8 {int(lens[lens <= 8].sum())}, mid {int(lens[(lens > 8) & (lens <= 32)].sum())}, "
              f"long {int(lens[lens > 32].sum())}", flush=True)
        idx_modes = {"real": mcol,
                     "seq": (torch.arange(plan.nnz, device=dev) % plan.ncols).to(torch.int32),
                     "cached": (torch.arange(plan.nnz, device=dev) % 1024).to(torch.int32)}
        for ld in (12,):
            M = torch.zeros((plan.ncols * ld + 8,), device=dev)[:plan.ncols * ld].view(plan.ncols, ld)
            M[mpos, :F] = Mc
            Y = torch.empty((N, F), device=dev)
            for mode in ("real", "seq", "cached"):
                ix = idx_modes[mode]
                for which in (7, 1, 2, 4):
                    for xcd in ((1, 0) if which == 7 else (1,)):
                        def run():
                            rc = lab.lab_v3(N, rowptr.data_ptr(), ix.data_ptr(), mval.data_ptr(), M.data_ptr(), ld, F,
                                            Y.data_ptr(), n_chunks, c_beg.data_ptr(), c_end.data_ptr(),
                                            c_row.data_ptr(), partials.data_ptr(), n_long, long_row32.data_ptr(),
                                            lcptr32.data_ptr(), len(mid), mid.data_ptr(), xcd, which, st)
                            assert rc == 0, rc
                        Y.fill_(float("nan"))
                        run()
                        torch.cuda.synchronize()
                        tag = ""
                        if mode == "real" and which == 7:
                            err = float((Y.double() - Yref).abs().max())
                            tag = f"err {err:.1e} {'OK' if err <= tol else 'WRONG'}"
                        ms = event_time_ms(run, a.iters, st)
                        report(f"v3 ld={ld} idx={mode} part={which} xcd={xcd}", ms, tag)
            del M

    # ---- 4. persistent pipelined row-owner kernels on the production layout --------------------------
    if a.pers:
        ld = 12
        M = torch.zeros((plan.ncols * ld + 8,), device=dev)[:plan.ncols * ld].view(plan.ncols, ld)
        M[mpos, :F] = Mc
        mcol = torch.from_numpy(plan.export(L.ARR_MCOL)).to(dev)
        mval = torch.from_numpy(plan.export(L.ARR_MVAL)).to(dev)
        rowptr = torch.from_numpy(plan.export(L.ARR_ROWPTR)).to(dev)
        rp64 = rowptr.long()
        lens = rp64[1:] - rp64[:-1]
        idx_modes = {"real": mcol,
                     "seq": (torch.arange(plan.nnz, device=dev) % plan.ncols).to(torch.int32),
                     "cached": (torch.arange(plan.nnz, device=dev) % 1024).to(torch.int32)}
        for T in (8, 4):
            thr = 4 * T
            long_rows = torch.nonzero(lens > thr).flatten()
            nch = (lens[long_rows] + 511) // 512
            n_long, n_chunks = len(long_rows), int(nch.sum())
            lcptr = torch.zeros(n_long + 1, dtype=torch.int64, device=dev)
            lcptr[1:] = torch.cumsum(nch, 0)
            crl = torch.repeat_interleave(torch.arange(n_long, device=dev), nch)
            k = torch.arange(n_chunks, device=dev) - lcptr[crl]
            r = long_rows[crl]
            c_beg = (rp64[r] + k * 512).to(torch.int32)
            c_end = torch.minimum(rp64[r] + (k + 1) * 512, rp64[r + 1]).to(torch.int32)
            c_row = torch.where(nch[crl] == 1, r, -(r + 2)).to(torch.int32)
            long_row32, lcptr32 = long_rows.to(torch.int32), lcptr.to(torch.int32)
            partials = torch.zeros((max(n_chunks, 1), 16), device=dev)
            Y = torch.empty((N, F), device=dev)
            for GB in ((1, 2, 4) if T == 8 else (2, 4)):
                for wpc, cwpc in ((16, 8), (32, 8), (8, 8), (16, 16), (16, 4)):
                    for mode in (("real", "cached") if (wpc, cwpc) == (16, 8) else ("real",)):
                        ix = idx_modes[mode]
                        for which in ((3, 1, 2) if (wpc, cwpc) == (16, 8) else (3,)):
                            def run():
                                rc = lab.lab_rows_p(T, GB, wpc, cwpc, N, rowptr.data_ptr(), ix.data_ptr(),
                                                    mval.data_ptr(), M.data_ptr(), ld, F, Y.data_ptr(), n_chunks,
                                                    c_beg.data_ptr(), c_end.data_ptr(), c_row.data_ptr(),
                                                    partials.data_ptr(), n_long, long_row32.data_ptr(),
                                                    lcptr32.data_ptr(), which, st)
                                assert rc == 0, rc
                            Y.fill_(float("nan"))
                            run()
                            torch.cuda.synchronize()
                            tag = ""
                            if mode == "real" and which == 3:
                                err = float((Y.double() - Yref).abs().max())
                                tag = f"err {err:.1e} {'OK' if err <= tol else 'WRONG'}"
                            ms = event_time_ms(run, a.iters, st)
                            report(f"persistent T={T} GB={GB} w/CU={wpc},{cwpc} idx={mode} part={which}", ms,
                                   f"chunks {n_chunks} {tag}")
        del M

    # ---- 3. entry-sliced kernel on the production operand layout ---------------------------------
    if a.seg:
        mcol = torch.from_numpy(plan.export(L.ARR_MCOL)).to(dev)
        mval = torch.from_numpy(plan.export(L.ARR_MVAL)).to(dev)
        for ld in (12, 16):
            M = torch.zeros((plan.ncols * ld + 8,), device=dev)[:plan.ncols * ld].view(plan.ncols, ld)
            M[mpos, :F] = Mc
            for K, srw in ((8, 8), (8, 4), (8, 16), (4, 16), (16, 4)):
                unit = 16 * K * srw
                npad = (plan.nnz + unit - 1) // unit * unit
                pad = npad - plan.nnz
                idx_p = torch.cat([mcol, mcol[-1:].expand(pad)]).contiguous()
                val_p = torch.cat([mval, torch.zeros(pad, device=dev)]).contiguous()
                row_p = torch.cat([rowidx.to(torch.int32), rowidx[-1:].to(torch.int32).expand(pad)]).contiguous()
                nw = npad // unit
                rec_row = torch.full((2 * nw,), -7, dtype=torch.int32, device=dev)
                rec_val = torch.zeros((2 * nw, 16), device=dev)
                Y = torch.empty((N, F), device=dev)
                for xcd in (1, 0):
                    def run():
                        rc = lab.lab_seg(K, srw, npad, idx_p.data_ptr(), val_p.data_ptr(), row_p.data_ptr(),
                                         M.data_ptr(), ld, F, Y.data_ptr(), F, rec_row.data_ptr(),
                                         rec_val.data_ptr(), xcd, st)
                        assert rc == 0, rc
                    Y.fill_(float("nan"))
                    run()
                    torch.cuda.synchronize()
                    err = float((Y.double() - Yref).abs().max())
                    ms = event_time_ms(run, a.iters, st)
                    report(f"entry-sliced K={K} srw={srw} ld={ld} xcd={xcd}", ms,
                           f"waves {nw}, err {err:.1e} {'OK' if err <= tol else 'WRONG'}")
            del M

    # ---- 2. stream + gather -------------------------------------------------------------------
    nnz = plan.nnz
    ent_cnt = cnt[ccol]
    for H in (a.H if a.sg else []):
        is_g = ent_cnt >= H
        # per row: gathered entries first (in operand-region order), then the stream entries
        gcols = torch.nonzero(cnt >= H).flatten()
        order = torch.argsort(cnt[gcols], descending=True, stable=True)
        gpos = torch.full((plan.ncols,), -1, dtype=torch.int64, device=dev)
        gpos[gcols[order]] = torch.arange(len(gcols), device=dev)
        ng_row = torch.zeros(N, dtype=torch.int64, device=dev).index_add_(0, rowidx, is_g.long())
        ns_row = torch.zeros(N, dtype=torch.int64, device=dev).index_add_(0, rowidx, (~is_g).long())
        gptr = torch.zeros(N + 1, dtype=torch.int64, device=dev)
        gptr[1:] = torch.cumsum(ng_row, 0)
        sptr = torch.zeros(N + 1, dtype=torch.int64, device=dev)
        sptr[1:] = torch.cumsum(ns_row, 0)
        ge = torch.nonzero(is_g).flatten()      # row-major order is kept inside both classes
        se = torch.nonzero(~is_g).flatten()
        # gathered entries of a row sorted by operand row
        key = rowidx[ge] * (len(gcols) + 1) + gpos[ccol[ge]]
        ge = ge[torch.argsort(key)]
        gidx = gpos[ccol[ge]].to(torch.int32)
        gval = val[ge].contiguous()
        sval = val[se].contiguous()
        n_g, n_s = len(gcols), len(se)
        Mg = torch.zeros((max(n_g, 1), 16), device=dev)
        Mg[gpos[gcols], :F] = Mc[gcols]
        Ms = torch.zeros((n_s * F + 8,), device=dev)
        Ms[:n_s * F].view(n_s, F).copy_(Mc[ccol[se]])
        # long rows -> chunks of <= 512 entries over [gathered | stream]
        tot = ng_row + ns_row
        long_rows = torch.nonzero(tot > 32).flatten()
        nch = (tot[long_rows] + 511) // 512
        n_long, n_chunks = len(long_rows), int(nch.sum())
        lcptr = torch.zeros(n_long + 1, dtype=torch.int64, device=dev)
        lcptr[1:] = torch.cumsum(nch, 0)
        crow_l = torch.repeat_interleave(torch.arange(n_long, device=dev), nch)
        k = torch.arange(n_chunks, device=dev) - lcptr[crow_l]
        r = long_rows[crow_l]
        lo, hi = k * 512, torch.minimum((k + 1) * 512, tot[r])
        c_gb = gptr[r] + torch.minimum(lo, ng_row[r])
        c_ge = gptr[r] + torch.minimum(hi, ng_row[r])
        c_sb = sptr[r] + torch.clamp(lo - ng_row[r], min=0)
        c_se = sptr[r] + torch.clamp(hi - ng_row[r], min=0)
        c_row = torch.where(nch[crow_l] == 1, r, -(r + 2))
        i32 = lambda t: t.to(torch.int32).contiguous()  # noqa: E731
        gptr32, sptr32 = i32(gptr), i32(sptr)
        c_gb, c_ge, c_sb, c_se, c_row = i32(c_gb), i32(c_ge), i32(c_sb), i32(c_se), i32(c_row)
        long_row32, lcptr32 = i32(long_rows), i32(lcptr)
        partials = torch.zeros((max(n_chunks, 1), 16), device=dev)
        Y = torch.empty((N, F), device=dev)
        bytes_read = (len(ge) * 8 + n_s * (4 + 4 * F) + n_g * 64 + (N + 1) * 8 + N * F * 4)
        for su in a.su:
            for xcd in (1, 0):
                def run():
                    rc = lab.lab_sg(F, su, N, gptr32.data_ptr(), gidx.data_ptr(), gval.data_ptr(), sptr32.data_ptr(),
                                    sval.data_ptr(), Mg.data_ptr(), Ms.data_ptr(), Y.data_ptr(), n_chunks,
                                    c_row.data_ptr(), c_gb.data_ptr(), c_ge.data_ptr(), c_sb.data_ptr(),
                                    c_se.data_ptr(), partials.data_ptr(), n_long, long_row32.data_ptr(),
                                    lcptr32.data_ptr(), xcd, st)
                    assert rc == 0, rc
                Y.fill_(float("nan"))
                run()
                err = check(Y, f"sg H={H}")
                ms = event_time_ms(run, a.iters, st)
                report(f"stream+gather H={H} su={su} xcd={xcd}", ms,
                       f"gather rows {n_g} ({n_g * 64 / 1e6:.1f} MB), gathered entries {len(ge)}, stream rows {n_s} "
                       f"(x{n_s / plan.ncols:.2f} of ncols), long rows {n_long}, chunks {n_chunks}, "
                       f"min bytes {bytes_read / 1e6:.0f} MB, err {err:.1e}")
        del Mg, Ms, gidx, gval, sval, partials


if __name__ == "__main__":
    main()
