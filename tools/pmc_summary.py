#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counters (`*_counter_collection.csv`).
usage: python tools/pmc_summary.py <dir> [name-filter]"""
import csv
import glob
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short  # noqa: E402


def main():
    d = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
    agg = defaultdict(lambda: [0, 0.0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if filt and filt not in k:
                    continue
                a = agg[(k, row["Counter_Name"], row.get("Grid_Size", ""))]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    print("| kernel | grid | counter | dispatches | avg value |")
    print("|---|---|---|---:|---:|")
    for (k, c, g), (n, s) in sorted(agg.items()):
        print(f"| {k} | {g} | {c} | {n} | {s / n:.1f} |")


if __name__ == "__main__":
    main()
