#!/usr/bin/env python
"""Condenses a rocprofv3 `*_kernel_stats.csv` (kernel names shortened) into a small table.
usage: python tools/prof_summary.py gpurun_out/prof1 [top_n] > profiles/r01_xxx.md"""
import csv
import glob
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"(mrgcn::k_[a-z_0-9]+(<[^>]*>)?)", name)
    if m:
        return m.group(1)
    if name.startswith("Cijk_"):
        return "rocBLAS/Tensile GEMM " + name[:24]
    m = re.search(r"rocprim::\w+::detail::(\w+)", name)
    if "radix_sort" in name:
        return "rocprim radix_sort (plan build)"
    if "scan_impl" in name or "lookback_scan" in name:
        return "rocprim scan (plan build)"
    m = re.search(r"at::native::(\w+)", name)
    if m:
        f = re.search(r"(\w+Functor|normal_kernel|FillFunctor|\w+_kernel_cuda)", name)
        return "torch " + m.group(1) + ("/" + f.group(1) if f else "")
    return name[:60]


def main():
    d = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
    agg = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Name"])
            a = agg.setdefault(k, [0, 0.0])
            a[0] += int(row["Calls"])
            a[1] += float(row["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    print(f"source: {f}")
    print(f"total kernel time: {tot/1e6:.3f} ms\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"| {k} | {c} | {t/1e6:.3f} | {t/c/1e3:.1f} | {100*t/tot:.2f} |")


if __name__ == "__main__":
    main()
