#!/usr/bin/env python
"""The kernels of ONE epoch in launch order, from a rocprofv3 kernel_trace.csv: the last stretch between two
launches of the epoch's first kernel (default: the layer-0 relation transform).  Shows every launch — this package's
kernels, torch's, the runtime's fill / copy kernels — with its start offset, duration, grid and stream (queue), and
the gaps in which the device idled.
usage: python tools/epoch_sequence.py <dir> [first-kernel-substring [launches of that kernel per epoch | shortest]]"""
import csv
import glob
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short  # noqa: E402


# kernels of this package that live outside namespace mrgcn (file-local helpers of rgcn_fused.hip / distmult.hip)
OWN_UNSCOPED = ("k_basis_contract", "k_corrupt_triples", "k_count3", "k_scan3", "k_fill3", "k_random_subset")


def main():
    f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
    first = sys.argv[2] if len(sys.argv) > 2 else "k_xform_mfma_fwd<1, false, 16"
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    shortest = len(sys.argv) > 3 and sys.argv[3] == "shortest"
    per = int(sys.argv[3]) if len(sys.argv) > 3 and not shortest else 1
    if len(starts) < per + 1:
        raise SystemExit("fewer than two epochs in the trace")
    a, b = starts[-1 - per], starts[-1]
    if shortest:  # the shortest stretch: a replayed epoch of a small graph (the run's last epochs may be eager ones)
        a, b = min(zip(starts[:-1], starts[1:]),
                   key=lambda ab: int(rows[ab[1]]["Start_Timestamp"]) - int(rows[ab[0]]["Start_Timestamp"]))
    t0 = int(rows[a]["Start_Timestamp"])
    print(f"source: {f}\nepoch = launches {a}..{b - 1}: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us from first "
          f"kernel to the next epoch's first kernel\n")
    print("| # | start us | dur us | idle before us | queue | grid | kernel |")
    print("|---:|---:|---:|---:|---:|---:|---|")
    busy_until, other = t0, 0.0
    for i, r in enumerate(rows[a:b]):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = max(0, s - busy_until) / 1e3
        busy_until = max(busy_until, e)
        name = short(r["Kernel_Name"])
        if "mrgcn::" not in name and not any(k in name for k in OWN_UNSCOPED):
            other += (e - s) / 1e3
        print(f"| {i} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap:.1f} | {r.get('Queue_Id', '')} | "
              f"{r['Grid_Size_X']} | {name} |")
    print(f"\nkernels that are not this package's (fills, copies, torch elementwise, library GEMMs): {other:.1f} us")


if __name__ == "__main__":
    main()
