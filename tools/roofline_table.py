#!/usr/bin/env python
"""Per-kernel roofline table of one AM-shaped epoch: algorithmic bytes, median launch duration from a
rocprofv3 kernel trace, achieved GB/s and the fraction of the 8 TB/s HBM roofline; with a PMC summary directory the
counter traffic (TCC_EA0_RDREQ by size + WRITE_SIZE) next to it.
    python tools/roofline_table.py <trace dir> [bench_line.json] [pmc dir] > profiles/rNN_kernel_roofline.md
The counts of columns / nodes / entries that carry gradient come from the bench line (extra.gradient_support)."""
import collections
import csv
import glob
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short  # noqa: E402

# AM-shaped graph (mrgcn_amd/synth.py, seed 0) and model
N, R, B, NNZ, NCOLS = 1666764, 267, 40, 13643406, 8165256
K0, F0, F1 = 155, 10, 11
LD = 12
GB = 1e9
PEAK = 8000.0


def spmm_bytes(rows, ncols, F):
    return NNZ * 8 + (rows + 1) * 4 + ncols * F * 4 + rows * F * 4


def main():
    f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
    L0, L1, NL0, NL1, E0, E1 = 2028994, 6538, 828598, 5565, 2883183, 6783
    if len(sys.argv) > 2:
        try:
            gs = json.load(open(sys.argv[2]))["extra"]["gradient_support"]
            (L0, L1), (NL0, NL1), (E0, E1) = gs["live_cols"], gs["live_nodes"], gs["live_entries"]
        except Exception as e:  # noqa: BLE001
            print(f"(bench line unreadable: {e}; counts of the seed-0 AM graph used)\n")
    pmc = {}
    if len(sys.argv) > 3:
        for fn in glob.glob(sys.argv[3] + "*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                k = short(r["Kernel_Name"])
                a = pmc.setdefault(k, collections.defaultdict(lambda: [0, 0.0]))[r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    ALG = collections.OrderedDict([  # kernel-name prefix -> (what is counted, bytes)
        ("mrgcn::k_xform_mfma_fwd<1, false, 10, float, false>",
         ("layer-0 transform: X read once + W + addend written + indices", N * K0 * 4 + R * K0 * F0 * 4 + NCOLS * LD * 4 + NCOLS * 8)),
        ("mrgcn::k_mix_fwd_mfma<3, 2, 2, 2, float",
         ("basis mix: V read once + addend read + M written + 3 index arrays", 4 * B * N * F0 + NCOLS * (LD * 4 + F0 * 4) + NCOLS * 8 + N * 4)),
        ("mrgcn::k_spmm3<4, 4, false, float, true, 7>", ("forward product, F=10 (SURVEY 8d formula; the F=11 launches move 7.7 % more)", spmm_bytes(N, NCOLS, F0))),
        ("mrgcn::k_xform_mfma_fwd<1, false, 1, float, false>",
         ("layer-1 transform: H read once + W + M written + indices", N * F0 * 4 + R * F0 * F1 * 4 + NCOLS * F1 * 4 + NCOLS * 12)),
        ("mrgcn::k_xform_cols_lds<12, float>",
         ("layer-1 transform, output order, all weights in LDS: H read once + W + M written + (node, relation) ids", N * F0 * 4 + R * F0 * F1 * 4 + NCOLS * F1 * 4 + NCOLS * 8)),
        ("mrgcn::k_mix_bwd_sup<10, 2, 512, true>",
         ("norm-only mix backward: V of the live nodes + their dM rows in, D rows out", NL0 * B * F0 * 4 + L0 * (LD * 4 + B * 4 + 4) + NL0 * 8)),
        ("mrgcn::k_dcomp_chunks<10>", ("dcomp: D rows in, by relation", L0 * (B * 4 + 4))),
        ("mrgcn::k_xform_mfma_dw<3, 2, false, float>",
         ("layer-0 dW over the live columns: X rows of the live NODES once + dM rows + lists", NL0 * K0 * 4 + L0 * (LD * 4 + 8))),
        ("mrgcn::k_adam_rows_fused",
         ("Adam, gradient formed on the fly: p, m, v of the live blocks in and out + their dM rows", 6 * 4 * B * NL0 * F0 + L0 * LD * 4 + N * 6)),
        ("mrgcn::k_adam_rows_once<2, 1>",
         ("Adam as a one-shot grid over the support's node list (a wave per node), gradient formed on the fly: p, m, v of the live blocks in and out + their dM rows + the list", 6 * 4 * B * NL0 * F0 + L0 * (LD * 4 + 4) + NL0 * 8)),
        ("mrgcn::k_adam_rows_list<2, true>",
         ("Adam over the support's node list, gradient formed on the fly: p, m, v of the live blocks in and out + their dM rows + the list", 6 * 4 * B * NL0 * F0 + L0 * (LD * 4 + 4) + NL0 * 8)),
    ])
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"source: {f}\n")
    print("HBM roofline 8.0 TB/s (MI355X spec; on these boxes a float4 copy reaches 6.25 TB/s and a 3-read / 3-write triad 6.0 TB/s as ONE-SHOT grids, 4.8 / 4.9 as persistent grid-stride loops: profiles/r06_copy_lab.txt).")
    print("`alg. MB` are ALGORITHMIC bytes: every operand counted once, gathered rows counted once however often they are")
    print("re-read.  `counter MB` = TCC_EA0_RDREQ (32 / 64 / 128-byte requests) + WRITE_SIZE of separate --pmc passes.\n")
    print("| kernel | what is counted | alg. MB | launches | median us | achieved GB/s | % of 8 TB/s | counter MB | counter / alg. |")
    print("|---|---|---:|---:|---:|---:|---:|---:|---:|")
    def pick(table, prefix):  # (kernel names grow template arguments: match by prefix)
        for name in table:
            if name.startswith(prefix):
                return name
        return prefix

    for k, (label, nbytes) in list(ALG.items()):
        full = pick(d, k)
        if full != k:
            ALG[full] = ALG.pop(k)
    for k, (label, nbytes) in ALG.items():
        v = sorted(d.get(k, []))
        if not v:
            continue
        med = v[len(v) // 2]
        gbs = nbytes / (med * 1e-6) / GB
        c = pmc.get(k)
        cm, ratio = "", ""
        if c:
            avg = {n: a[1] / a[0] for n, a in c.items()}
            tot = (avg.get("TCC_EA0_RDREQ_32B_sum", 0) * 32 + avg.get("TCC_EA0_RDREQ_64B_sum", 0) * 64
                   + avg.get("TCC_EA0_RDREQ_128B_sum", 0) * 128 + avg.get("WRITE_SIZE", 0) * 1024)
            if tot > 0:
                cm, ratio = f"{tot / 1e6:.0f}", f"{tot / nbytes:.2f}"
        print(f"| {k} | {label} | {nbytes/1e6:.0f} | {len(v)} | {med:.0f} | {gbs:.0f} | {100*gbs/PEAK:.1f} | {cm} | {ratio} |")
    print("\nEvery other kernel of the epoch (launch-latency territory), medians:\n")
    print("| kernel | launches | median us |")
    print("|---|---:|---:|")
    for k, v in sorted(d.items(), key=lambda kv: -sorted(kv[1])[len(kv[1]) // 2]):
        if k in ALG or "mrgcn::" not in k and "k_basis" not in k:
            continue
        v = sorted(v)
        if v[len(v) // 2] >= 3.0:
            print(f"| {k} | {len(v)} | {v[len(v) // 2]:.0f} |")


if __name__ == "__main__":
    main()
