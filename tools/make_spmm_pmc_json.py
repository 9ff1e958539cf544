#!/usr/bin/env python
"""Builds profiles/spmm_pmc_latest.json from rocprofv3 --pmc passes of tools/spmm_probe.py.
HBM bytes per launch follow MI355X_MICROARCH.md §HBM: reads = TCC_EA0_RDREQ requests by size
(on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so FETCH_SIZE*2 when all requests are
128 B — both are recorded), writes = WRITE_SIZE (KB, exact)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrgcn_amd.build import source_digest  # noqa: E402


def avg(dirpat, kernel_sub):
    out = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(dirpat + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_sub in row["Kernel_Name"] and "finalize" not in row["Kernel_Name"]:
                a = out[row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return {k: v[1] / v[0] for k, v in out.items()}


def main():
    """make_spmm_pmc_json.py <pass prefix> <kernel substring> [workload=am] [F=10] [operand=f32]"""
    base, kernel = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "am"
    F = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    operand = sys.argv[5] if len(sys.argv) > 5 else "f32"
    c = {}
    for d in glob.glob(base + "*"):
        c.update(avg(d, kernel))
    rd = c.get("TCC_EA0_RDREQ_32B_sum", 0) * 32 + c.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 \
        + c.get("TCC_EA0_RDREQ_128B_sum", 0) * 128
    wr = c.get("WRITE_SIZE", 0) * 1024
    out = {"workload": workload, "F": F, "operand": operand, "kernel": kernel, "source_digest": source_digest(),
           "counters": c,
           "read_bytes_from_rdreq_sizes": rd, "read_bytes_fetch_size_x2": c.get("FETCH_SIZE", 0) * 1024 * 2,
           "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
