"""Shared helpers for the test-suite (golden loading, numpy plan reference)."""
import os

import numpy as np
import scipy.sparse as sp
import torch

HOT_MIN_REFS = 16  # csrc/common.hpp: kHotMinRefs
NODE_BAND = 131072  # csrc/common.hpp: kNodeBand
STRADDLE_ROW_BYTES = (48,)  # plan.hip: the default operand row size of k_avoid_straddle
LEN_WINDOW_S, LEN_WINDOW_M = 0, 0  # csrc/common.hpp: kLenWindowS / kLenWindowM (0 = off)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_graph(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    A = sp.csr_matrix((g["csr_data"], g["csr_indices"], g["csr_indptr"]), shape=tuple(g["shape"]))
    return g, A


def graph_of_case(case_name):
    return "graph_smoke" if "_smoke_" in case_name else "graph_small"


def load_case(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rgcn_cases():
    import glob
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "rgcn_*.npz")))


def coo_tensor(A_csr, value_mode, device="cpu"):
    """What FullBatch.as_tensors_ produces (int8) or its float32 sibling."""
    idx = np.array(A_csr.nonzero())
    dtype = torch.int8 if value_mode == "ref_int8" else torch.float32
    t = torch.sparse_coo_tensor(torch.LongTensor(idx), torch.Tensor(A_csr.data), A_csr.shape, dtype=dtype)
    return t.to(device)


def ref_layout(t, name):
    """A tensor in the layout of the node-major `weight_I` parameter, (N, B, out), as the reference's (B*N, out);
    anything else (by parameter name) unchanged."""
    if t is not None and name.endswith("weight_I") and t.dim() == 3:
        N, B, F = t.shape
        return t.permute(1, 0, 2).reshape(B * N, F)
    return t


def build_rgcn_from_case(c, device, engine="fused"):
    from mrgcn_amd.models.rgcn import RGCN
    N, R, B = int(c["meta.num_nodes"]), int(c["meta.R"]), int(c["meta.num_bases"])
    bias, fl = bool(c["meta.bias"]), bool(c["meta.featureless"])
    lp = bool(c["meta.link_prediction"])
    dims = [tuple(int(x) for x in d) for d in c["dims"]]
    modules = []
    for li, (i, o) in enumerate(dims):
        act = torch.nn.ReLU() if (li < len(dims) - 1 or lp) else None
        modules.append((i, o, "mrgcn", act))
    torch.manual_seed(int(c["meta.seed"]))
    model = RGCN(modules, R, N, B, 0.0, fl, bias, lp)
    return model, dims


def load_state_from_case(model, c, prefix="init."):
    sd = {k[len(prefix):]: torch.from_numpy(np.array(c[k])) for k in c.files if k.startswith(prefix)}
    model.load_state_dict(sd, strict=True)


# ---------------------------------------------------------------------------------------
# numpy reference of the graph plan (index arrays must match the device plan bit for bit)
# ---------------------------------------------------------------------------------------
def numpy_plan(rows, cols, vals, num_rows, N, R, prune=False, row_bytes=STRADDLE_ROW_BYTES):
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    v = np.asarray(vals).astype(np.float32)
    if prune:
        keep = v != 0
        rows, cols, v = rows[keep], cols[keep], v[keep]
    RN = R * N
    key = rows * RN + cols
    order = np.argsort(key, kind="stable")
    key, v = key[order], v[order]
    rowidx = (key // RN).astype(np.int32)
    lcol = (key % RN).astype(np.int32)
    rowptr = np.searchsorted(key, np.arange(num_rows + 1, dtype=np.int64) * RN).astype(np.int32)
    rel, node = lcol // N, lcol % N
    key2 = node.astype(np.int64) * R + rel
    order2 = np.argsort(key2, kind="stable")
    key2s = key2[order2]
    head = np.ones(len(key2s), dtype=bool)
    head[1:] = key2s[1:] != key2s[:-1]
    cid = np.cumsum(head) - 1
    ncols = int(cid[-1] + 1) if len(cid) else 0
    ccol = np.empty(len(key), dtype=np.int32)
    ccol[order2] = cid
    crow = rowidx[order2]
    cval = v[order2]
    cptr = np.concatenate([np.nonzero(head)[0], [len(key)]]).astype(np.int32)
    uk = key2s[head]
    unode = (uk // R).astype(np.int32)
    urel = (uk % R).astype(np.int32)
    ulcol = (urel.astype(np.int64) * N + unode).astype(np.int32)
    nptr = np.searchsorted(unode, np.arange(N + 1)).astype(np.int32)
    band = min(NODE_BAND, N)
    nbands = (N + band - 1) // band
    k3 = ((unode.astype(np.int64) // band) * R + urel) * band + unode.astype(np.int64) % band
    rperm = np.argsort(k3, kind="stable").astype(np.int32)
    relptr = np.searchsorted(k3[rperm], np.arange(nbands * R + 1, dtype=np.int64) * band).astype(np.int32)
    # COMPACT view: rows in class-major order (<= 8 entries, <= 32, more; row order inside a class)
    lens = np.diff(rowptr.astype(np.int64))
    cls = np.where(lens <= 8, 0, np.where(lens <= 32, 1, 2))
    # inside the S / M classes: windows of LEN_WINDOW_S / _M consecutive row ids, a window's rows by length, then by id
    ids = np.arange(num_rows)
    sub = np.where(cls == 0, (ids // LEN_WINDOW_S) * 64 + lens if LEN_WINDOW_S else 0,
                   np.where(cls == 1, (ids // LEN_WINDOW_M) * 64 + lens if LEN_WINDOW_M else 0, 0))
    rowmap = np.lexsort((ids, sub, cls)).astype(np.int32)
    rank = np.empty(num_rows, dtype=np.int64)
    rank[rowmap] = np.arange(num_rows)
    ptr3 = np.concatenate([[0], np.cumsum(lens[rowmap])]).astype(np.int32)
    # storage order of the compact operand: hot columns (>= HOT_MIN_REFS entries) by falling count,
    # then every other column by the rank of the first row that reads it
    cnt = np.diff(cptr).astype(np.int64)
    if ncols:
        firstrow = rank[crow[cptr[:-1]].astype(np.int64)]
        max_count = int(cnt.max()) + 1
        hi = np.where(cnt >= HOT_MIN_REFS, max_count - cnt, max_count + 1 + firstrow)
        order = np.lexsort((np.arange(ncols), hi))
        # inside every aligned group of 32 positions, columns with several readers leave the slots whose rows straddle a
        # 128-byte line to single-reader columns (plan.hip::k_avoid_straddle; row sizes STRADDLE_ROW_BYTES)
        order = order.copy()
        n_hot = int((cnt >= HOT_MIN_REFS).sum())
        for g0 in range(0, ncols - 31, 32):
            if g0 < n_hot < g0 + 32:
                continue  # the group that holds the end of the hot region stays as it is
            grp = order[g0:g0 + 32]
            multi = cnt[grp] > 1
            pos = g0 + np.arange(32, dtype=np.int64)
            bad = np.zeros(32, dtype=bool)
            for sz in row_bytes:
                bad |= (pos * sz) // 128 != (pos * sz + sz - 1) // 128
            todo = [i for i in range(32) if bad[i] and multi[i]]
            free = [j for j in range(32) if not bad[j] and not multi[j]]
            for i, j in zip(todo, free):
                grp[i], grp[j] = grp[j], grp[i]
        mpos = np.empty(ncols, dtype=np.int32)
        mpos[order] = np.arange(ncols, dtype=np.int32)
    else:
        mpos = np.zeros(0, dtype=np.int32)
    if len(key):  # COMPACT view: ranks in order, entries of a row in rising operand position
        mc = mpos[ccol].astype(np.int64)
        o3 = np.argsort(rank[rowidx.astype(np.int64)] * max(ncols, 1) + mc, kind="stable")
        mcol, mval = mc[o3].astype(np.int32), v[o3]
    else:
        mcol, mval = np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32)
    return dict(rowptr=rowptr, lcol=lcol, ccol=ccol, val=v, rowidx=rowidx, cptr=cptr, crow=crow,
                cval=cval, urel=urel, unode=unode, ulcol=ulcol, nptr=nptr, rperm=rperm,
                relptr=relptr, mpos=mpos, mcol=mcol, mval=mval, rowmap=rowmap, ptr3=ptr3, ncols=ncols, nnz=len(key))
