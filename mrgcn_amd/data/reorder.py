"""Node numbering for semi-supervised training.

With a handful of labelled nodes only the nodes within `num_layers` in-neighbour steps of a label
ever receive gradient in the input layer's node table (`weight_I`, graph.py:62-75): every other
row of it has a zero gradient, zero Adam moments and never moves.  The HIP backward and `ClipAdam`
skip such rows per node (the node-major table keeps a node's block contiguous: functional.py, row-sparse
gradient), in any numbering; numbering the reachable nodes first additionally makes the touched blocks one
contiguous range (whole pages of the table and of its moments are then never touched: `bench.py`'s
`extra.epoch_ms_nodes_renumbered`).  The reference numbers nodes arbitrarily (the order of
`mkdataset`'s entity index), so renumbering is a pure relabelling of the dataset:

    order, inv = label_reach_order(rows, cols, N, R, idx, hops=num_layers)
    rows, cols = relabel_coo(rows, cols, N, inv)         # adjacency N x (R*N)
    X, idx     = X[order], inv[idx]                      # features, labelled nodes
    # logits_new[inv] are the logits of the original numbering
"""
import numpy as np
import scipy.sparse as sp


def label_reach_order(rows, cols, num_nodes: int, num_relations: int, idx, hops: int = 2):
    """`order` (new id -> old id) and `inv` (old id -> new id): first, in rising old id, the nodes
    that rows within `hops - 1` in-neighbour steps of `idx` read from (the nodes whose `weight_I`
    rows can receive gradient in a `hops`-layer R-GCN); then all others, in rising old id."""
    rows = np.asarray(rows, dtype=np.int64)
    src = np.asarray(cols, dtype=np.int64) % num_nodes
    reads = sp.csr_matrix((np.ones(len(rows), dtype=np.int8), (rows, src)), shape=(num_nodes, num_nodes))
    reach = np.zeros(num_nodes, dtype=bool)
    reach[np.asarray(idx, dtype=np.int64)] = True
    for _ in range(hops):
        nxt = reach.copy()
        nxt[np.unique(reads[np.flatnonzero(reach)].indices)] = True
        reach = nxt
    order = np.concatenate([np.flatnonzero(reach), np.flatnonzero(~reach)]).astype(np.int64)
    inv = np.empty(num_nodes, dtype=np.int64)
    inv[order] = np.arange(num_nodes, dtype=np.int64)
    return order, inv


def relabel_coo(rows, cols, num_nodes: int, inv):
    """The stacked adjacency N x (R*N) with node ids mapped through `inv` (rows and source nodes)."""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    rel, src = cols // num_nodes, cols % num_nodes
    return inv[rows], rel * num_nodes + inv[src]


def relabel_csr(A, num_nodes: int, inv):
    """The stacked CSR adjacency N x (R*N) (archive member `A`, tarball.py:151-157) renumbered."""
    coo = sp.coo_matrix(A)
    rows, cols = relabel_coo(coo.row, coo.col, num_nodes, inv)
    return sp.csr_matrix((coo.data, (rows, cols)), shape=A.shape)


def renumber_dataset(d: dict, order, inv) -> dict:
    """A dataset archive as `mrgcn_amd.data.dataset.load_tarball` returns it (members of
    `mkdataset.py`: A, Y, F, data, sample_map, class_map), with every node id mapped through `inv`:
    adjacency rows / source nodes, the rows of the label matrices, the node indices of the feature
    encodings (`[encodings, node_idx, seq_len]` per datatype) and subject / object of the link-
    prediction triples.  Logits of a model trained on the result are those of the original numbering
    at `logits_new[inv]`."""
    order = np.asarray(order, dtype=np.int64)
    inv = np.asarray(inv, dtype=np.int64)
    N = len(order)
    out = dict(d)
    if d.get("A") is not None:
        out["A"] = relabel_csr(d["A"], N, inv)
    if d.get("Y") is not None:
        out["Y"] = {split: (Y[order] if Y is not None else None) for split, Y in d["Y"].items()}
    if d.get("F") is not None:
        out["F"] = {dt: [[enc, inv[np.asarray(node_idx, dtype=np.int64)], seq_len] + list(rest)
                         for enc, node_idx, seq_len, *rest in sets]
                    for dt, sets in d["F"].items()}
    if d.get("data") is not None:
        data = {}
        for split, triples in d["data"].items():
            t = np.array(triples, copy=True)
            if t.ndim == 2 and t.shape[1] >= 3 and t.size:
                t[:, 0], t[:, 2] = inv[t[:, 0]], inv[t[:, 2]]
            data[split] = t
        out["data"] = data
    return out
