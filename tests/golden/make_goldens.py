#!/usr/bin/env python
"""Golden-vector generator (AUTHORING CONTAINER ONLY).

Imports the *reference* implementation from /root/reference (with a stub for the
absent `rdflib`, which is only imported at module scope and never exercised on
the R-GCN path), drives it on small seeded graphs and stores inputs + outputs as
`.npz` fixtures next to this script.  Nothing from the reference tree is copied:
the fixtures are data (inputs, expected outputs).

    python tests/golden/make_goldens.py

The reference never travels to the GPU box; the tests only read the `.npz`.

Reference call sites driven here (file:line relative to /root/reference):
  mrgcn/encodings/graph_structure.py:162-169  normalize_adjacency_matrix
  mrgcn/encodings/graph_structure.py:33-38    identity block + hstack
  mrgcn/data/io/tarball.py:151-157            CSR stored/re-read as float32
  mrgcn/data/batch.py:144-149                 FullBatch.as_tensors_ (COO int8)
  mrgcn/layers/graph.py:62-102                GraphConvolution.forward
  mrgcn/models/rgcn.py:69-89                  RGCN._forward_full_batch
  mrgcn/models/mrgcn.py:189-214               MRGCN._forward_full_batch
  mrgcn/tasks/node_classification.py:166-193  train step (driven by hand, numpy
                                              label index: scipy rejects torch idx)
  mrgcn/tasks/node_classification.py:432-444  accuracy / cross-entropy
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _stub_rdflib():
    rdflib = types.ModuleType("rdflib")
    term = types.ModuleType("rdflib.term")
    namespace = types.ModuleType("rdflib.namespace")

    class URIRef(str):
        def neq(self, other):
            return str(self) != str(other)

    class Literal(str):
        pass

    class BNode(str):
        pass

    class Namespace(str):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return URIRef(str(self) + name)

    term.URIRef, term.Literal, term.BNode = URIRef, Literal, BNode
    rdflib.term = term
    rdflib.URIRef, rdflib.Literal, rdflib.BNode = URIRef, Literal, BNode
    rdflib.Namespace = Namespace
    rdflib.Graph = type("Graph", (), {})
    namespace.XSD = Namespace("http://www.w3.org/2001/XMLSchema#")
    namespace.Namespace = Namespace
    rdflib.namespace = namespace
    sys.modules["rdflib"] = rdflib
    sys.modules["rdflib.term"] = term
    sys.modules["rdflib.namespace"] = namespace


def import_reference():
    _stub_rdflib()
    # the repo root may hold an alias package called `mrgcn`; the reference wins here
    sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == "mrgcn" or k.startswith("mrgcn.")]:
        del sys.modules[k]
    import mrgcn.layers.graph as ref_graph
    import mrgcn.models.rgcn as ref_rgcn
    import mrgcn.models.mrgcn as ref_mrgcn
    import mrgcn.data.batch as ref_batch
    import mrgcn.data.utils as ref_dutils
    import mrgcn.encodings.graph_structure as ref_gs
    import mrgcn.tasks.node_classification as ref_nc
    import mrgcn.tasks.utils as ref_tutils
    return types.SimpleNamespace(graph=ref_graph, rgcn=ref_rgcn, mrgcn=ref_mrgcn,
                                 batch=ref_batch, dutils=ref_dutils, gs=ref_gs,
                                 nc=ref_nc, tutils=ref_tutils)


# --------------------------------------------------------------------------
# graph construction (integer triples -> reference adjacency builder)
# --------------------------------------------------------------------------
def random_triples(rng, num_nodes, num_pred, num_triples):
    """Unique (s, p, o) integer triples; predicate frequencies skewed."""
    w = 1.0 / np.arange(1, num_pred + 1)
    w /= w.sum()
    seen = set()
    out = []
    while len(out) < num_triples:
        p = int(rng.choice(num_pred, p=w))
        # skewed endpoints: square a uniform to favour low ids
        s = int(num_nodes * rng.random() ** 2)
        o = int(num_nodes * rng.random() ** 2)
        if (s, p, o) in seen:
            continue
        seen.add((s, p, o))
        out.append((s, p, o))
    return np.array(out, dtype=np.int64)


def reference_adjacency(ref, triples, num_nodes, num_pred):
    """graph_structure.py:70-108 driven with integer edges, :33-38 identity+hstack,
    tarball.py:151-157 float32 round trip."""
    shape = (num_nodes, num_nodes)
    adjacencies = []
    for p in range(num_pred):
        e = triples[triples[:, 1] == p]
        row, col = e[:, 0].astype(np.int32), e[:, 2].astype(np.int32)
        data = np.ones(len(row), dtype=np.int8)
        adj = sp.csr_matrix((data, (row, col)), shape=shape, dtype=np.int8)
        adjacencies.append(ref.gs.normalize_adjacency_matrix(adj))
        adj = sp.csr_matrix((data, (col, row)), shape=shape, dtype=np.int8)
        adjacencies.append(ref.gs.normalize_adjacency_matrix(adj))
    ident = sp.identity(num_nodes).tocsr()
    adjacencies.append(ref.gs.normalize_adjacency_matrix(ident))
    A = sp.hstack(adjacencies, format="csr")
    # what Tarball stores and returns
    A = sp.csr_matrix((A.data.astype(np.float32), A.indices, A.indptr),
                      shape=A.shape, dtype=np.float32)
    return A


def state_to_np(prefix, sd):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def grads_to_np(prefix, model):
    out = {}
    for n, p in model.named_parameters():
        if p.grad is not None:
            out[prefix + n] = p.grad.detach().cpu().numpy().copy()
    return out


# --------------------------------------------------------------------------
# case drivers
# --------------------------------------------------------------------------
def run_rgcn_case(ref, name, A_csr, num_nodes, R, dims, num_bases, bias,
                  featureless, value_mode, seed, labels_idx, labels_y, n_adam=5,
                  link_prediction=False):
    """RGCN-level golden: X fed directly (rgcn.py:63-89), then the hand-driven
    train step of node_classification.py:166-193."""
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)

    if value_mode == "ref_int8":
        A_t = ref.dutils.scipy_sparse_to_pytorch_sparse(A_csr, dtype=torch.int8)
    else:  # "norm_f32": layer-level contract is value-generic (A.float())
        A_t = ref.dutils.scipy_sparse_to_pytorch_sparse(A_csr, dtype=torch.float32)

    modules = []
    for li, (i, o) in enumerate(dims):
        act = torch.nn.ReLU() if (li < len(dims) - 1 or link_prediction) else None
        modules.append((i, o, "mrgcn", act))
    model = ref.rgcn.RGCN(modules, R, num_nodes, num_bases, 0.0, featureless,
                          bias, link_prediction)
    if bias:  # non-zero biases so that the bias path is visible
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith(".b"):
                    p.copy_(torch.from_numpy(
                        rng.standard_normal(p.shape).astype(np.float32) * 0.1))

    X = None
    if not featureless:
        X = torch.from_numpy(rng.standard_normal((num_nodes, dims[0][0])).astype(np.float32))
        X.requires_grad_(True)

    out = {}
    out.update(state_to_np("init.", model.state_dict()))
    if X is not None:
        out["X"] = X.detach().numpy().copy()

    # per-layer activations (same loop as rgcn.py:71-87)
    with torch.no_grad():
        H = X
        for li, (layer, act) in enumerate(zip(model.layers.values(),
                                              model.activations.values())):
            H = layer(H, A_t)
            out[f"act.pre_{li}"] = H.numpy().copy()
            if act is not None:
                H = act(H)
            out[f"act.post_{li}"] = H.numpy().copy()

    criterion = torch.nn.CrossEntropyLoss()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=0.0)
    targets = torch.as_tensor(labels_y, dtype=torch.long)
    for step in range(1, n_adam + 1):
        Y_hat = model(X, A_t)
        loss = criterion(Y_hat[labels_idx], targets)
        optimizer.zero_grad()
        if X is not None and X.grad is not None:
            X.grad = None
        loss.backward()
        if step == 1:
            out["logits"] = Y_hat.detach().numpy().copy()
            out["loss"] = np.float32(loss.item())
            out.update(grads_to_np("grad.", model))
            if X is not None:
                out["grad.X"] = X.grad.detach().numpy().copy()
        total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        if step == 1:
            out["grad_norm"] = np.float32(total_norm.item())
        optimizer.step()
        if step in (1, n_adam):
            out.update(state_to_np(f"adam{step}.", model.state_dict()))
        out[f"loss_step{step}"] = np.float32(loss.item())

    meta = dict(num_nodes=num_nodes, R=R, num_bases=num_bases, bias=int(bias),
                featureless=int(featureless), seed=seed, n_adam=n_adam,
                link_prediction=int(link_prediction))
    out["dims"] = np.array(dims, dtype=np.int64)
    out["value_mode"] = np.array(value_mode)
    out["labels_idx"] = np.asarray(labels_idx, dtype=np.int64)
    out["labels_y"] = np.asarray(labels_y, dtype=np.int64)
    for k, v in meta.items():
        out["meta." + k] = np.int64(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"  {name}: loss={out['loss']:.6f} |logits|={np.abs(out['logits']).max():.4f}")


def run_mrgcn_case(ref, name, A_csr, num_nodes, R, hidden, num_classes, num_bases,
                   seed, labels_idx, labels_y, with_encoders):
    """MRGCN-level golden through FullBatch (mrgcn.py:182-214, batch.py:135-149,
    node_classification.py:128-133)."""
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)

    X = [np.empty((num_nodes, 0), dtype=float)]
    modules_config = []
    X_width = 0
    enc_in = {}
    if with_encoders:
        # two encoding sets of xsd.numeric (4 -> 4) and one xsd.boolean (1 -> 2):
        # what graph_features.construct_feature_matrix emits for config 2
        idx_a = np.sort(rng.choice(num_nodes, num_nodes // 3, replace=False))
        idx_b = np.sort(rng.choice(num_nodes, num_nodes // 4, replace=False))
        enc_a = rng.uniform(-1, 1, (len(idx_a), 4)).astype(np.float32)
        enc_b = rng.choice([-1.0, 1.0], (len(idx_b), 1)).astype(np.float32)
        X.append(["xsd.numeric", [[enc_a, idx_a, np.ones(len(idx_a), dtype=int)]], False])
        X.append(["xsd.boolean", [[enc_b, idx_b, np.ones(len(idx_b), dtype=int)]], False])
        modules_config = [("xsd.numeric", (4, 4, 0.0), False),
                          ("xsd.boolean", (1, 2, 0.0), False)]
        modules_config.sort(key=lambda t: t[0])
        X_width = 6
        enc_in = {"enc.numeric": enc_a, "enc.numeric_idx": idx_a,
                  "enc.boolean": enc_b, "enc.boolean_idx": idx_b}
    featureless = X_width <= 0

    modules = [(X_width, hidden, "mrgcn", torch.nn.ReLU()),
               (hidden, num_classes, "mrgcn", None)]
    model = ref.mrgcn.MRGCN(modules, modules_config, R, num_nodes,
                            num_bases=num_bases, p_dropout=0.0,
                            featureless=featureless, bias=False,
                            gcn_gpu_acceleration=False)

    batch = ref.batch.FullBatch(A_csr, X, np.arange(num_nodes))
    batch.pad_(pad_symbols=dict())
    batch.to_dense_()
    batch.as_tensors_()
    batch.to(model.devices)

    out = {}
    out.update(state_to_np("init.", model.state_dict()))
    out.update(enc_in)
    out["param_names"] = np.array([n for n, _ in model.named_parameters()])
    opt_cfg = {"gate_weights": {}, "xsd.numeric": {}, "xsd.boolean": {}}
    groups = ref.tutils.optimizer_params(model, opt_cfg, featureless)
    out["optim_group_sizes"] = np.array([len(g["params"]) for g in groups], dtype=np.int64)

    criterion = torch.nn.CrossEntropyLoss()
    optimizer = torch.optim.Adam(groups, lr=0.01, weight_decay=0.0)
    targets = torch.as_tensor(labels_y, dtype=torch.long)
    for step in (1, 2, 3):
        model.train()
        Y_hat = model(batch).to("cpu")
        loss = criterion(Y_hat[labels_idx], targets)
        optimizer.zero_grad()
        loss.backward()
        if step == 1:
            out["logits"] = Y_hat.detach().numpy().copy()
            out["loss"] = np.float32(loss.item())
            out.update(grads_to_np("grad.", model))
            Yc = sp.csr_matrix((np.ones(len(labels_idx), dtype=np.int8),
                                (labels_idx, labels_y)),
                               shape=(num_nodes, num_classes), dtype=np.int8)
            acc = ref.nc.categorical_accuracy(Y_hat.detach(), Yc)[0]
            out["accuracy"] = np.float32(acc.item())
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        out.update(state_to_np(f"adam{step}.", model.state_dict()))

    out["labels_idx"] = np.asarray(labels_idx, dtype=np.int64)
    out["labels_y"] = np.asarray(labels_y, dtype=np.int64)
    for k, v in dict(num_nodes=num_nodes, R=R, hidden=hidden, num_classes=num_classes,
                     num_bases=num_bases, seed=seed,
                     with_encoders=int(with_encoders)).items():
        out["meta." + k] = np.int64(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"  {name}: loss={out['loss']:.6f}")


def save_graph(name, triples, num_nodes, num_pred, A_csr, ref):
    A_i8 = ref.dutils.scipy_sparse_to_pytorch_sparse(A_csr, dtype=torch.int8)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        triples=triples, num_nodes=np.int64(num_nodes), num_pred=np.int64(num_pred),
        csr_data=A_csr.data, csr_indices=A_csr.indices, csr_indptr=A_csr.indptr,
        shape=np.array(A_csr.shape, dtype=np.int64),
        has_sorted_indices=np.int64(bool(A_csr.has_sorted_indices)),
        coo_indices=A_i8._indices().numpy(), coo_values_i8=A_i8._values().numpy())
    print(f"  {name}: N={num_nodes} R={2*num_pred+1} nnz={A_csr.nnz} "
          f"int8-nonzero={(A_i8._values().numpy() != 0).sum()}")


def main():
    ref = import_reference()
    print("reference imported from", os.path.dirname(ref.graph.__file__))

    # ---------------- graphs ----------------
    rng = np.random.default_rng(1234)
    g_small = dict(N=50, P=2, T=120)           # R = 5
    g_smoke = dict(N=2329, P=14, T=2594)       # R = 29, nnz = 2T+N = 7517 (smoke-test shape)
    graphs = {}
    for gname, g in (("graph_small", g_small), ("graph_smoke", g_smoke)):
        tr = random_triples(rng, g["N"], g["P"], g["T"])
        A = reference_adjacency(ref, tr, g["N"], g["P"])
        save_graph(gname, tr, g["N"], g["P"], A, ref)
        graphs[gname] = (A, g["N"], 2 * g["P"] + 1)

    # ---------------- RGCN-level cases, small graph: full cross ----------------
    A, N, R = graphs["graph_small"]
    lab_idx = np.sort(rng.choice(N, 20, replace=False))
    lab_y = rng.integers(0, 3, 20)
    seed = 100
    for featureless in (True, False):
        for nb in (0, 3):
            for bias in (False, True):
                for vm in ("ref_int8", "norm_f32"):
                    nm = (f"rgcn_small_{'fl' if featureless else 'ft'}_b{nb}_"
                          f"{'bias' if bias else 'nobias'}_{vm}")
                    indim = 0 if featureless else 7
                    run_rgcn_case(ref, nm, A, N, R, [(indim, 6), (6, 3)], nb, bias,
                                  featureless, vm, seed, lab_idx, lab_y)
                    seed += 1
    # single-layer link-prediction style encoder (ReLU on the only layer, relations table)
    run_rgcn_case(ref, "rgcn_small_lp_b2", A, N, R, [(0, 8)], 2, False, True,
                  "ref_int8", 777, lab_idx, rng.integers(0, 8, 20), link_prediction=True)

    # ---------------- RGCN-level cases, smoke-shape graph ----------------
    A, N, R = graphs["graph_smoke"]
    lab_idx = np.sort(rng.choice(N, 163, replace=False))
    lab_y = rng.integers(0, 2, 163)
    run_rgcn_case(ref, "rgcn_smoke_fl_b5_norm_f32", A, N, R, [(0, 4), (4, 2)], 5, False,
                  True, "norm_f32", 201, lab_idx, lab_y)
    run_rgcn_case(ref, "rgcn_smoke_ft_b5_norm_f32", A, N, R, [(9, 4), (4, 2)], 5, False,
                  False, "norm_f32", 202, lab_idx, lab_y)
    run_rgcn_case(ref, "rgcn_smoke_ft_b5_ref_int8", A, N, R, [(9, 4), (4, 2)], 5, True,
                  False, "ref_int8", 203, lab_idx, lab_y)
    run_rgcn_case(ref, "rgcn_smoke_fl_b0_norm_f32", A, N, R, [(0, 4), (4, 2)], 0, False,
                  True, "norm_f32", 204, lab_idx, lab_y)

    # ---------------- MRGCN-level cases (FullBatch boundary) ----------------
    A, N, R = graphs["graph_small"]
    lab_idx = np.sort(rng.choice(N, 20, replace=False))
    lab_y = rng.integers(0, 3, 20)
    run_mrgcn_case(ref, "mrgcn_small_featureless_b0", A, N, R, 6, 3, 0, 301,
                   lab_idx, lab_y, with_encoders=False)
    run_mrgcn_case(ref, "mrgcn_small_featureless_b3", A, N, R, 6, 3, 3, 302,
                   lab_idx, lab_y, with_encoders=False)
    run_mrgcn_case(ref, "mrgcn_small_encoders_b3", A, N, R, 6, 3, 3, 303,
                   lab_idx, lab_y, with_encoders=True)


if __name__ == "__main__":
    main()
