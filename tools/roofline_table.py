#!/usr/bin/env python
"""Per-kernel roofline table of one AM-shaped epoch: algorithmic bytes, median
launch duration from a rocprofv3 kernel trace, achieved GB/s and the fraction of the 8 TB/s HBM
roofline.   python tools/roofline_table.py <trace dir> > profiles/rNN_kernel_roofline.md"""
import collections
import csv
import glob
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


LIVE_NODES = 0.497  # share of the nodes with a live column on this graph (1 000 labels, two hops)
LIVE_COLS = 0.2485  # share of the compact columns with gradient in layer 0

ALG = {  # kernel-name prefix -> (label, bytes)
    "mrgcn::k_adam<false>": ("Adam on weight_I: 7 streams x 4 B x B*N*F0", 7 * 4 * B * N * F0),
    "mrgcn::k_mix_fwd_mfma<3, 2, 2, true, float>": ("V read once + addend read + M written + 3 index arrays",
                                                    4 * B * N * F0 + 2 * NCOLS * LD * 4 + NCOLS * 8 + N * 4),
    "mrgcn::k_mix_fwd<40, float>": ("V read once + addend read + M written + 3 index arrays",
                                    4 * B * N * F0 + 2 * NCOLS * LD * 4 + NCOLS * 8 + N * 4),
    "mrgcn::k_mix_bwd_nm<10>": ("norm-only pass (dcomp, ||dV||^2, node flags): V read for the live nodes + live dM "
                                "rows + flags", int(4 * B * N * F0 * LIVE_NODES) + int(NCOLS * LIVE_COLS) * LD * 4
                                + NCOLS * 2 + N * 5),
    "mrgcn::k_adam_rows_fused": ("Adam with the gradient formed on the fly: p, m, v read and written for the node "
                                 "blocks that ever had gradient + their live dM rows",
                                 int(6 * 4 * B * N * F0 * LIVE_NODES) + int(NCOLS * LIVE_COLS) * LD * 4 + N * 10),
    "mrgcn::k_adam_rows<4>": ("row-sparse Adam from a stored gradient: p, g, m, v read and p, m, v written for the "
                              "node blocks that ever had gradient", int(7 * 4 * B * N * F0 * LIVE_NODES)),
    "mrgcn::k_spmm3<4, 4, false, float, true, 7>": ("forward product, F=10 (SURVEY 8d formula)", spmm_bytes(N, NCOLS, F0)),
    "mrgcn::k_spmm<4, 4, true, float, false>": ("general transposed product, F=10 (probe leg only; the epoch runs k_spmm_t_live)",
                                                spmm_bytes(NCOLS, N, F0)),
    "mrgcn::k_xform_mfma_fwd<1, false, 16, float, false>": ("layer-0 transform: X read once + W + M2 written + indices",
                                              N * K0 * 4 + R * K0 * F0 * 4 + NCOLS * LD * 4 + NCOLS * 8),
    "mrgcn::k_xform_mfma_fwd<1, false, 1, float, false>": ("layer-1 transform: H read once + W + M written + indices",
                                             N * F0 * 4 + R * F0 * F1 * 4 + NCOLS * LD * 4 + NCOLS * 12),
}
# kernels whose traffic depends on how many columns carry gradient: listed with their times only
LIVE = ["mrgcn::k_spmm_t_live<12>", "mrgcn::k_xform_mfma_dw<3, 2, true>", "mrgcn::k_xform_mfma_dw<1, 2, true>",
        "mrgcn::k_xform_mfma_fwd<1, true, 1, float, true>", "mrgcn::k_segment_sum_live<16>",
        "mrgcn::k_rows_live_mark", "mrgcn::k_long_rows_mark", "mrgcn::k_spmm<4, 4, true, float, true>"]


def main():
    f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k in ALG or k in LIVE:
            d[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"source: {f}\n")
    print("HBM roofline 8.0 TB/s (MI355X spec; about 6.3 TB/s is reachable by a copy).  `bytes` are ALGORITHMIC:")
    print("every operand counted once, gathered rows counted once however often they are re-read.\n")
    print("| kernel | what is counted | alg. MB | launches | median us | achieved GB/s | % of 8 TB/s |")
    print("|---|---|---:|---:|---:|---:|---:|")
    for k, (label, nbytes) in ALG.items():
        v = sorted(d.get(k, []))
        if not v:
            continue
        if k == "mrgcn::k_adam<false>":
            v = [x for x in v if x > 1000]  # the weight_I launch (absent when the node-major Adam runs)
            if not v:
                continue
        med = v[len(v) // 2]
        gbs = nbytes / (med * 1e-6) / GB
        print(f"| {k} | {label} | {nbytes/1e6:.0f} | {len(v)} | {med:.0f} | {gbs:.0f} | {100*gbs/PEAK:.1f} |")
    print("\nBackward kernels that sweep only the compact columns with gradient (traffic depends on the labels):\n")
    print("| kernel | launches | median us |")
    print("|---|---:|---:|")
    for k in LIVE:
        v = sorted(d.get(k, []))
        if v:
            print(f"| {k} | {len(v)} | {v[len(v) // 2]:.0f} |")


if __name__ == "__main__":
    main()
