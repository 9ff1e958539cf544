"""Times every product of a TCNN (forward, dX, dW of each Conv1d; the fully connected tail) on its own through
`mrgcn_gemm_f32` and prints TFLOP/s per product and over all of them (fp32 MFMA peak: 157 TFLOP/s).
    python tools/gemm_probe.py [--size M] [--batch 2048] [--features 37] [--length 300] [--iters 10] [--json]
"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from mrgcn_amd import dense  # noqa: E402
from mrgcn_amd.models.temporal_cnn import _SPECS  # noqa: E402


ONLY = ""


def timed(fn, iters, name=""):
    if ONLY not in name:
        return None
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="M")
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--features", type=int, default=37)
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--wt", action="store_true", help="forward products read the transposed weight ([Cin KW][Cout])")
    ap.add_argument("--only", default="", help="substring of the product names to time")
    ap.add_argument("--mm", default="f32", choices=["f32", "bf16"],
                    help="bf16: tiles rounded to bf16 as they are staged, v_mfma_f32_16x16x32_bf16 (the bf16 pipeline)")
    a = ap.parse_args()
    if a.mm == "bf16":
        dense.set_matmul_dtype("bf16")
    peak, peak_name = (2500.0, "bf16 MFMA (dense)") if a.mm == "bf16" else (157.3, "fp32 MFMA")
    global ONLY
    ONLY = a.only
    dev = torch.device("cuda:0")
    Bn, T, C = a.batch, a.length, a.features
    rows = []
    first = True
    for kind, *args in _SPECS[a.size][1]:
        if kind == "p":
            T //= args[0]
            continue
        if kind == "a":
            T = args[0]
            continue
        Cout, KW, pad = args
        Tout = T + 2 * pad - KW + 1
        x = torch.randn(Bn, C, T, device=dev)
        W = torch.randn(Cout, C, KW, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        y = torch.empty(Bn, Cout, Tout, device=dev)
        dy = torch.randn(Bn, Cout, Tout, device=dev)
        geom = (C, T, KW, pad, Tout, Cout)
        flop = 2.0 * Bn * Tout * Cout * C * KW
        tag = f"conv {C}->{Cout} k{KW} T{T}"
        Wv = W.view(Cout, C * KW)
        if a.wt:
            Wt = Wv.t().contiguous()
            ms = timed(lambda: dense._gemm(2, 0, 2, Bn * Tout, Cout, C * KW, x, 0, Wt, Cout, y, 0, bias=b, geom=geom), a.iters, tag + " fwd")
        else:
            ms = timed(lambda: dense._gemm(2, 1, 2, Bn * Tout, Cout, C * KW, x, 0, Wv, C * KW, y, 0, bias=b, geom=geom), a.iters, tag + " fwd")
        rows.append((tag + " fwd", Bn * Tout, Cout, C * KW, flop, ms))
        if not first:
            Wf = W.flip(2).permute(0, 2, 1).reshape(Cout * KW, C).contiguous()
            dx = torch.empty_like(x)
            g2 = (Cout, Tout, KW, KW - 1 - pad, T, C)
            ms = timed(lambda: dense._gemm(2, 0, 2, Bn * T, C, Cout * KW, dy, 0, Wf, C, dx, 0, geom=g2), a.iters, tag + " dX")
            rows.append((tag + " dX", Bn * T, C, Cout * KW, 2.0 * Bn * T * C * Cout * KW, ms))
        dWt = torch.empty(C * KW, Cout, device=dev)
        ms = timed(lambda: dense._gemm(3, 2, 0, C * KW, Cout, Bn * Tout, x, 0, dy, 0, dWt, Cout, geom=geom), a.iters, tag + " dW")
        rows.append((tag + " dW", C * KW, Cout, Bn * Tout, flop, ms))
        C, T, first = Cout, Tout, False
    # fully connected tail: C -> C (ReLU), the closing C -> 16 is too small to matter
    x = torch.randn(Bn, C, device=dev)
    W = torch.randn(C, C, device=dev) * 0.02
    b = torch.randn(C, device=dev)
    y = torch.empty(Bn, C, device=dev)
    flop = 2.0 * Bn * C * C
    ms = timed(lambda: dense._gemm(0, 1, 0, Bn, C, C, x, C, W, C, y, C, bias=b, relu=True), a.iters, f"fc {C}->{C} fwd")
    rows.append((f"fc {C}->{C} fwd", Bn, C, C, flop, ms))
    ms = timed(lambda: dense._gemm(0, 0, 0, Bn, C, C, y, C, W, C, x, C), a.iters, f"fc {C}->{C} dX")
    rows.append((f"fc {C}->{C} dX", Bn, C, C, flop, ms))
    dW = torch.empty(C, C, device=dev)
    ms = timed(lambda: dense._gemm(1, 0, 0, C, C, Bn, y, C, x, C, dW, C), a.iters, f"fc {C}->{C} dW")
    rows.append((f"fc {C}->{C} dW", C, C, Bn, flop, ms))

    rows = [r for r in rows if r[5] is not None]
    tot_f = sum(r[4] for r in rows)
    tot_ms = sum(r[5] for r in rows)
    if a.json:
        print(json.dumps({"products": [{"name": r[0], "M": r[1], "N": r[2], "K": r[3], "ms": round(r[5], 4),
                                        "tflops": round(r[4] / r[5] / 1e9, 1)} for r in rows],
                          "total_ms": round(tot_ms, 3), "total_tflops": round(tot_f / tot_ms / 1e9, 1),
                          "arithmetic": a.mm, "peak_tflops": peak, "peak": peak_name,
                          "frac": round(tot_f / tot_ms / 1e9 / peak, 4)}))
        return
    print(f"arithmetic: {a.mm}")
    print(f"{'product':34s} {'M':>8s} {'N':>5s} {'K':>7s} {'ms':>8s} {'TFLOP/s':>8s}")
    for name, M, N, K, flop, ms in rows:
        print(f"{name:34s} {M:8d} {N:5d} {K:7d} {ms:8.4f} {flop / ms / 1e9:8.1f}")
    print(f"{'all products':34s} {'':8s} {'':5s} {'':7s} {tot_ms:8.3f} {tot_f / tot_ms / 1e9:8.1f}   "
          f"({tot_f / tot_ms / 1e9 / peak * 100:.1f} % of {peak:g}: {peak_name})")


if __name__ == "__main__":
    main()
