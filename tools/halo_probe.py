#!/usr/bin/env python
"""Partition choice of SURVEY §8(e), measured on the synthetic graphs (host only, no GPU):

  column / owner partition (mrgcn_amd.partition): rank g owns the source nodes of range g and the columns (r, j in g) of
      A; per layer every rank RECEIVES its own rows of the other ranks' partial outputs: (G-1)/G * Np * out * 4 bytes
      in the forward reduce-scatter (and sends as much), the same again for the all-gather of the output gradient.
  row partition + halo exchange (north_star's wording): rank g owns the output rows of range g and needs, per layer,
      the input rows H[j] of every source node j outside its range that its rows read — its HALO: halo_g * in * 4
      bytes received in the forward, the same volume back (gradient of those rows) in the backward.  The input term's
      node table would additionally have to be read remotely or replicated (it is indexed by source node too).

    python tools/halo_probe.py [am synth10m ...] > profiles/r04_halo.json
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrgcn_amd import synth  # noqa: E402


def probe(name, worlds=(2, 4, 8)):
    t0 = time.time()
    g = synth.make_graph(name, seed=0)
    N, R = g.num_nodes, g.num_relations
    dims = synth.layer_dims(name)
    rows = g.rows.astype(np.int64)
    src = (g.cols % N).astype(np.int64)
    out = {"N": N, "R": R, "nnz": int(g.nnz), "layers": dims, "gen_s": round(time.time() - t0, 1), "worlds": {}}
    for G in worlds:
        S = (N + G - 1) // G
        ro, so = rows // S, src // S
        halo, halo_cols = [], []
        cols = g.cols.astype(np.int64)
        for r in range(G):
            m = (ro == r) & (so != r)
            halo.append(int(np.unique(src[m]).size))
            halo_cols.append(int(np.unique(cols[m]).size))   # distinct remote (relation, node) columns: operand rows
        Np = S * G
        rec = {"halo_nodes_per_rank": halo, "halo_share_of_remote_nodes": float(np.mean(halo)) / max(N - S, 1)}
        per_layer = []
        for (k_in, k_out) in dims:
            # bytes one rank RECEIVES in the forward of the layer
            col = (G - 1) / G * Np * k_out * 4
            # the halo carries the layer INPUT (layer 0 with features: X rows, K = x_width; featureless layer 0: nothing
            # but the node table's blocks, B * out floats per halo node when bases are used)
            width = k_in if k_in > 0 else 0
            row = float(np.mean(halo)) * width * 4
            per_layer.append({"in": k_in, "out": k_out, "column_partition_recv_bytes": col,
                              "row_partition_halo_recv_bytes": row})
        rec["per_layer"] = per_layer
        B = synth.SHAPES[name]["bases"]
        f0 = dims[0][1]
        rec["node_table_halo_bytes_if_not_replicated"] = float(np.mean(halo)) * (B if B > 0 else R) * f0 * 4
        tot_c = sum(p["column_partition_recv_bytes"] for p in per_layer)
        tot_r = sum(p["row_partition_halo_recv_bytes"] for p in per_layer) + rec["node_table_halo_bytes_if_not_replicated"]
        rec["forward_recv_bytes_per_rank"] = {"column_partition": tot_c, "row_partition_halo": tot_r}
        # the cheaper row-partition form exchanges OPERAND rows (the owner of node j mixes / transforms its columns and
        # ships the rows remote readers need): distinct remote columns x out x 4 per layer
        tot_o = sum(float(np.mean(halo_cols)) * p["out"] * 4 for p in per_layer)
        rec["halo_columns_per_rank"] = halo_cols
        rec["forward_recv_bytes_per_rank"]["row_partition_operand_rows"] = tot_o
        best = min(rec["forward_recv_bytes_per_rank"].items(), key=lambda kv: kv[1])
        rec["smaller"] = best[0]
        out["worlds"][str(G)] = rec
    return out


def main():
    names = sys.argv[1:] or ["am", "synth10m", "fb15k"]
    res = {n: probe(n) for n in names}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
