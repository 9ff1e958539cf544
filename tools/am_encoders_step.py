#!/usr/bin/env python
"""One replayed step of bench.am_encoders_record (BASELINE config 3, full multimodal) kernel by kernel.

  rocprofv3 --kernel-trace --output-format csv -d <dir> -o run -- python3 tools/am_encoders_step.py run [f32|bf16]
  python3 tools/am_encoders_step.py summary <dir>        -> markdown on stdout
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(compute="f32"):
    import torch
    import bench
    rec = bench.am_encoders_record(argparse.Namespace(seed=0), torch.device("cuda:0"), steps=10, warm=2, compute=compute)
    print(json.dumps({k: v for k, v in rec.items() if k != "config"}))


def summary(d):
    f = glob.glob(os.path.join(d, "**", "run_kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a step of the encoders model = between two Adam passes over the node table with encoder kernels in between
    marks = [i for i, r in enumerate(rows) if "k_adam_rows_list" in r["Kernel_Name"] or "k_adam_rows_once" in r["Kernel_Name"]]
    best = None
    for a, b in zip(marks[:-1], marks[1:]):
        if any("k_mm_tile" in rows[i]["Kernel_Name"] for i in range(a, b)):
            best = (a, b)
    a, b = best
    seg = rows[a + 1:b + 1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print(f"source: {f}")
    print(f"one replayed step of `extra.workloads.am_encoders`: {len(seg)} launches, {(t1 - t0) / 1e6:.2f} ms from the first "
          f"kernel to the end of the last, {busy / 1e6:.2f} ms of kernel time\n")
    agg = collections.defaultdict(lambda: [0, 0.0])
    group = collections.defaultdict(float)
    for r in seg:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg[k[:90]][0] += 1
        agg[k[:90]][1] += us
        g = ("this package: encoders (MLP / TCNN / heads)" if any(t in k for t in ("k_mm_tile", "k_bn_", "k_mlp_", "k_gemm", "k_colsum_f32", "k_pool", "k_chan_sum"))
             else "this package: R-GCN layers, loss, optimizer" if "mrgcn::" in k or k.startswith("k_basis") or "k_basis" in k
             else "torch / MIOpen / hipBLASLt (stand-in backbones, glue)")
        group[g] += us
    print("| part | us |\n|---|---:|")
    for g, us in sorted(group.items(), key=lambda kv: -kv[1]):
        print(f"| {g} | {us:.0f} |")
    print("\n| kernel | launches | total us |\n|---|---:|---:|")
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"| {k} | {n} | {us:.1f} |")


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "summary":
        summary(sys.argv[2])
    else:
        run(sys.argv[2] if len(sys.argv) >= 3 and sys.argv[1] == "run" else "f32")
