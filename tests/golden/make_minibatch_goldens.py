#!/usr/bin/env python
"""Golden vectors for the mini-batch path, AUTHORING CONTAINER ONLY: imports the
reference and drives, on the 50-node golden graph,
  mrgcn/data/batch.py:150-263    MiniBatch / A_Batch._populate / getNeighboursSparse /
                                 getAdjacencyNodeColumnIdx / sliceSparseCOO
  mrgcn/models/rgcn.py:91-128    RGCN._forward_mini_batch
  mrgcn/layers/graph.py:62-102   GraphConvolution.forward(X, A, A_idx)
and records batch structure, logits of the batch nodes, loss and every gradient.
    python tests/golden/make_minibatch_goldens.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402


def main():
    ref = mg.import_reference()
    import mrgcn.data.batch as rb
    g = np.load(os.path.join(HERE, "graph_small.npz"))
    N, P = int(g["num_nodes"]), int(g["num_pred"])
    R = 2 * P + 1
    A_csr = mg.reference_adjacency(ref, g["triples"], N, P)
    out = {}
    rng = np.random.default_rng(7)
    batch_idx = np.sort(rng.choice(N, 9, replace=False)).astype(np.int64)
    out["batch_idx"] = batch_idx
    for tag, (featureless, B, bias, nlayers, hidden, classes, xw) in {
            "ft_b3": (False, 3, True, 2, 6, 4, 5), "fl_b0": (True, 0, False, 2, 6, 4, 0),
            "ft_b0_l3": (False, 0, True, 3, 5, 3, 4), "fl_b2_l1": (True, 2, False, 1, 4, 4, 0)}.items():
        torch.manual_seed(11)
        X0 = None if featureless else rng.standard_normal((N, xw)).astype(np.float32)
        Xlist = None if featureless else [X0]
        mb = rb.MiniBatch(A_csr, Xlist, batch_idx, nlayers)
        for i, (nb, row) in enumerate(zip(mb.A.neighbours, mb.A.row)):
            out[f"{tag}.neighbours_{i}"] = np.asarray(nb, dtype=np.int64)
            out[f"{tag}.row_{i}.shape"] = np.asarray(row.shape)
        mb.as_tensors_()
        for i, row in enumerate(mb.A.row):
            out[f"{tag}.row_{i}.indices"] = row._indices().numpy().copy()
            out[f"{tag}.row_{i}.values"] = row._values().numpy().copy()
        dims = [(xw if li == 0 else hidden, hidden if li < nlayers - 1 else classes) for li in range(nlayers)]
        modules = [(i, o, "mrgcn", torch.nn.ReLU() if li < nlayers - 1 else None) for li, (i, o) in enumerate(dims)]
        model = ref.rgcn.RGCN(modules, R, N, B, 0.0, featureless, bias, False)
        if bias:
            with torch.no_grad():
                for n, p in model.named_parameters():
                    if n.endswith(".b"):
                        p.copy_(torch.from_numpy(rng.standard_normal(p.shape).astype(np.float32) * 0.1))
        out.update(mg.state_to_np(f"{tag}.init.", model.state_dict()))
        X = None
        if not featureless:
            X = mb.X[0].float().clone().requires_grad_(True)   # rows of the outermost neighbours
            out[f"{tag}.X_full"] = X0
            out[f"{tag}.X_sub"] = X.detach().numpy().copy()
        logits = model(X, mb.A)
        y = torch.from_numpy(rng.integers(0, classes, len(batch_idx)))
        loss = torch.nn.CrossEntropyLoss()(logits, y)
        loss.backward()
        out[f"{tag}.logits"] = logits.detach().numpy().copy()
        out[f"{tag}.y"] = y.numpy()
        out[f"{tag}.loss"] = np.float32(loss.item())
        out.update(mg.grads_to_np(f"{tag}.grad.", model))
        if X is not None:
            out[f"{tag}.grad.X"] = X.grad.numpy().copy()
        out[f"{tag}.meta"] = np.asarray([int(featureless), B, int(bias), nlayers, hidden, classes, xw])
        # the slicing helpers on the innermost slice
        if not featureless:
            idx = rb.getAdjacencyNodeColumnIdx(mb.A.neighbours[0], N, R)
            sl = rb.sliceSparseCOO(mb.A.row[0], idx)
            out[f"{tag}.A_idx_0"] = idx.numpy().copy()
            out[f"{tag}.sliced_0.indices"] = sl._indices().numpy().copy()
            out[f"{tag}.sliced_0.values"] = sl._values().numpy().copy()
            out[f"{tag}.sliced_0.shape"] = np.asarray(sl.shape)
        print(tag, "neighbours", [len(n) for n in mb.A.neighbours], "loss", float(loss))
    np.savez_compressed(os.path.join(HERE, "minibatch_small.npz"), **out)


if __name__ == "__main__":
    main()
