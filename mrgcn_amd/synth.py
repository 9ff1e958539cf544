"""Deterministic synthetic knowledge graphs of the BASELINE shapes.

There is no network for the real AIFB / MUTAG / AM / FB15k-237 dumps, so the benchmark and
the full-size tests use random graphs with the same node / predicate / triple counts:
predicate frequencies ~ Zipf(1), subjects and objects drawn with power-law (alpha = 2.1)
popularity, unique (s, p, o).  The stacked adjacency follows the reference's layout contract
(mrgcn/encodings/graph_structure.py:13-38, :70-108, :162-169): blocks
[p0, p0^-1, p1, p1^-1, ..., identity], column r*N + j, values 1/deg_r(row); the reference's
boundary cast (mrgcn/data/batch.py:144-149) is applied in `value_mode="ref_int8"`."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .data.graph_structure import stacked_coo

# name -> (N, P, T, layers [(in, out)], num_bases, X width, labelled nodes, classes)
SHAPES = {
    "aifb":      dict(N=8285, P=45, T=29043, hidden=16, classes=4, bases=0, x_width=0, labelled=176),
    "mutag":     dict(N=23644, P=23, T=74227, hidden=16, classes=2, bases=30, x_width=8, labelled=340),
    "am":        dict(N=1666764, P=133, T=5988321, hidden=10, classes=11, bases=40, x_width=155,
                      labelled=1000),
    "fb15k":     dict(N=14541, P=237, T=310116, hidden=200, classes=0, bases=2, x_width=0, labelled=0),
    "synth10m":  dict(N=10_000_000, P=50, T=40_000_000, hidden=16, classes=11, bases=10, x_width=155,
                      labelled=10000),
}


@dataclass
class SynthGraph:
    name: str
    num_nodes: int
    num_pred: int
    num_relations: int
    triples: np.ndarray      # [T, 3] int64 (s, p, o), unique
    rows: np.ndarray         # [nnz] int64 COO row
    cols: np.ndarray         # [nnz] int64 COO column r*N + j
    vals: np.ndarray         # [nnz] float32 (norm_f32) or int8 (ref_int8)
    value_mode: str

    @property
    def nnz(self):
        return int(self.rows.shape[0])


def _powerlaw_ids(rng, n_items, size, alpha=2.1):
    """ids in [0, n_items) with P(rank k) ~ k^-gamma, gamma = 1/(alpha-1), via the inverse
    CDF of the continuous approximation; ranks are mapped through a fixed permutation."""
    gamma = 1.0 / (alpha - 1.0)
    u = rng.random(size)
    a = 1.0 - gamma
    x = ((n_items ** a - 1.0) * u + 1.0) ** (1.0 / a)
    k = np.minimum(np.floor(x).astype(np.int64) - 1, n_items - 1)
    return k


def _scaled(shape: dict, scale: float) -> dict:
    if scale == 1.0:
        return dict(shape)
    s = dict(shape)
    s["N"] = max(int(shape["N"] * scale), 64)
    s["T"] = max(int(shape["T"] * scale), 128)
    s["labelled"] = max(min(shape["labelled"], s["N"] // 4), 1) if shape["labelled"] else 0
    return s


def make_triples(N, P, T, seed):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, P + 1)
    w /= w.sum()
    perm_s = rng.permutation(N)
    perm_o = rng.permutation(N)
    out = np.empty((0,), dtype=np.int64)
    need = T
    keys = np.empty((0,), dtype=np.int64)
    while need > 0:
        m = int(need * 1.15) + 1024
        p = rng.choice(P, size=m, p=w).astype(np.int64)
        s = perm_s[_powerlaw_ids(rng, N, m)]
        o = perm_o[_powerlaw_ids(rng, N, m)]
        k = (p * N + s) * N + o
        keys = np.unique(np.concatenate([keys, k]))
        if keys.shape[0] >= T:
            # drop a random surplus so that the kept set is not biased towards small keys
            keep = rng.permutation(keys.shape[0])[:T]
            keys = keys[np.sort(keep)]
            break
        need = T - keys.shape[0]
    o = keys % N
    s = (keys // N) % N
    p = keys // (N * N)
    return np.stack([s, p, o], axis=1)


def make_graph(name: str, seed: int = 0, scale: float = 1.0, value_mode: str = "norm_f32") -> SynthGraph:
    sh = _scaled(SHAPES[name], scale)
    tr = make_triples(sh["N"], sh["P"], sh["T"], seed)
    rows, cols, vals, R = stacked_coo(tr, sh["N"], sh["P"], value_mode)
    return SynthGraph(name, sh["N"], sh["P"], R, tr, rows, cols, vals, value_mode)


def make_labels(name: str, num_nodes: int, seed: int = 0, scale: float = 1.0):
    sh = _scaled(SHAPES[name], scale)
    rng = np.random.default_rng(seed + 7919)
    n = min(sh["labelled"], num_nodes)
    idx = np.sort(rng.choice(num_nodes, n, replace=False)).astype(np.int64)
    y = rng.integers(0, max(sh["classes"], 1), n).astype(np.int64)
    return idx, y


def layer_dims(name: str):
    """[(in, out), ...] of the node-classification model of the config (two layers)."""
    sh = SHAPES[name]
    if sh["classes"] == 0:  # link prediction: a single layer (configs/fb15k-237.toml:88-97)
        return [(sh["x_width"], sh["hidden"])]
    return [(sh["x_width"], sh["hidden"]), (sh["hidden"], sh["classes"])]
