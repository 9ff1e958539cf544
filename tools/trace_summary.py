#!/usr/bin/env python
"""Per (kernel, launch shape) medians from a rocprofv3 kernel_trace.csv.
usage: python tools/trace_summary.py <dir> [name-filter]"""
import collections
import csv
import glob
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short  # noqa: E402


def main():
    f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
    filt = sys.argv[2] if len(sys.argv) > 2 else "mrgcn"
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if filt not in k:
            continue
        key = (k, r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r["Accum_VGPR_Count"],
               r["LDS_Block_Size"])
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("| kernel | grid | wg | vgpr | agpr | lds | calls | median us | min us |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for k, v in sorted(d.items()):
        v = sorted(v)
        print("| " + " | ".join(str(x) for x in k) + f" | {len(v)} | {v[len(v)//2]:.1f} | {v[0]:.1f} |")


if __name__ == "__main__":
    main()
