#!/usr/bin/env python
"""Optimizer-checkpoint golden (AUTHORING CONTAINER ONLY): what run.py:230-236 saves and
node_classification.py:73-80 loads — `model.state_dict()` + `optimizer.state_dict()` of the REFERENCE model under
`torch.optim.Adam` (node_classification.py:35-37) — captured after two hand-driven epochs on the small golden graph,
together with the state after a third epoch that resumed from it.  The moments of `weight_I` are in the reference's
`(B*N, out)` shape: this package's models must load them through the reference's own unpatched lines.

    python tests/golden/make_optim_goldens.py      ->  tests/golden/optim_checkpoint.npz
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402


def main():
    ref = mg.import_reference()
    g = np.load(os.path.join(HERE, "graph_small.npz"))
    A_csr = sp.csr_matrix((g["csr_data"], g["csr_indices"], g["csr_indptr"]), shape=tuple(g["shape"]))
    N, P = int(g["num_nodes"]), int(g["num_pred"])
    R = 2 * P + 1
    seed, B, dims = 21, 3, [(6, 8), (8, 4)]
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    A_t = ref.dutils.scipy_sparse_to_pytorch_sparse(A_csr, dtype=torch.float32)
    modules = [(6, 8, "mrgcn", torch.nn.ReLU()), (8, 4, "mrgcn", None)]
    model = ref.rgcn.RGCN(modules, R, N, B, 0.0, False, True, False)
    X = torch.from_numpy(rng.standard_normal((N, 6)).astype(np.float32))
    idx = np.sort(rng.choice(N, 20, replace=False))
    y = rng.integers(0, 4, 20)
    criterion = torch.nn.CrossEntropyLoss()
    optimizer = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=0.0)   # node_classification.py:35-37
    targets = torch.as_tensor(y, dtype=torch.long)
    out = {"X": X.numpy().copy(), "labels_idx": idx.astype(np.int64), "labels_y": y.astype(np.int64),
           "dims": np.array(dims), "meta.num_nodes": np.int64(N), "meta.R": np.int64(R), "meta.num_bases": np.int64(B),
           "param_names": np.array([n for n, _ in model.named_parameters()])}
    out.update(mg.state_to_np("init.", model.state_dict()))

    def dump_optim(prefix):
        sd = optimizer.state_dict()
        for k, st in sd["state"].items():
            out[f"{prefix}{k}.exp_avg"] = st["exp_avg"].numpy().copy()
            out[f"{prefix}{k}.exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
            out[f"{prefix}{k}.step"] = np.float64(float(st["step"]))

    for step in (1, 2, 3):
        loss = criterion(model(X, A_t)[idx], targets)
        optimizer.zero_grad()
        loss.backward()                                                    # node_classification.py:190-193
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        out[f"loss_step{step}"] = np.float32(loss.item())
        if step >= 2:
            out.update(mg.state_to_np(f"state{step}.", model.state_dict()))
            dump_optim(f"optim{step}.")
    np.savez_compressed(os.path.join(HERE, "optim_checkpoint.npz"), **out)
    print("optim_checkpoint.npz:", {k: v.shape for k, v in out.items() if k.startswith("optim2.")})


if __name__ == "__main__":
    main()
