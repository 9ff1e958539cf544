#!/usr/bin/env python
"""Golden vectors for two reference behaviours no shipped config switches on (AUTHORING CONTAINER ONLY):

  * node dropout, `p_dropout > 0` (mrgcn/models/rgcn.py:78-84): one Bernoulli draw per node on the CPU
    generator, applied in train AND eval mode;
  * `shared_bases_weights=True` (mrgcn/layers/graph.py:42-44): weight_F_comp is weight_I_comp.

Drives the reference on the 50-node golden graph and stores inputs + outputs in `extras.npz`.

    python tests/golden/make_extra_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_goldens import import_reference  # noqa: E402


def main():
    ref = import_reference()
    g = np.load(os.path.join(HERE, "graph_small.npz"))
    import scipy.sparse as sp
    A_csr = sp.csr_matrix((g["csr_data"], g["csr_indices"], g["csr_indptr"]), shape=tuple(g["shape"]))
    N, R = int(g["num_nodes"]), 2 * int(g["num_pred"]) + 1
    A = ref.dutils.scipy_sparse_to_pytorch_sparse(A_csr, dtype=torch.float32)
    rng = np.random.default_rng(21)
    out = {"meta.N": N, "meta.R": R}

    # ---- node dropout ------------------------------------------------------------------------------
    K, H, C, B, p = 6, 8, 3, 3, 0.4
    torch.manual_seed(31)
    model = ref.rgcn.RGCN([(K, H, "mrgcn", torch.nn.ReLU()), (H, C, "mrgcn", None)], R, N, B, p, False, True, False)
    X = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32))
    out.update({"drop.init." + k: v.detach().numpy().copy() for k, v in model.state_dict().items()})
    out["drop.X"], out["drop.p"] = X.numpy().copy(), p
    for mode in ("train", "eval"):  # the reference's functional dropout ignores the mode
        getattr(model, mode)()
        torch.manual_seed(77)  # the masks come from the CPU generator: two draws of N, in layer order
        out[f"drop.logits_{mode}"] = model(X, A).detach().numpy().copy()
    torch.manual_seed(77)
    out["drop.mask0"] = torch.nn.functional.dropout(torch.ones(N), p=p).numpy().copy()
    out["drop.mask1"] = torch.nn.functional.dropout(torch.ones(N), p=p).numpy().copy()
    model.train()
    torch.manual_seed(77)
    Xg = X.clone().requires_grad_(True)
    y = model(Xg, A)
    w = torch.from_numpy(rng.standard_normal((N, C)).astype(np.float32))
    (y * w).sum().backward()
    out["drop.w"] = w.numpy().copy()
    out["drop.grad.X"] = Xg.grad.numpy().copy()
    out.update({"drop.grad." + n: q.grad.numpy().copy() for n, q in model.named_parameters()})

    # ---- shared basis coefficients -------------------------------------------------------------------
    torch.manual_seed(41)
    layer = ref.graph.GraphConvolution(K, H, R, N, num_bases=B, bias=True, input_layer=True, featureless=False,
                                       shared_bases_weights=True)
    assert layer.weight_F_comp is layer.weight_I_comp
    with torch.no_grad():
        layer.b.copy_(torch.from_numpy(rng.standard_normal(H).astype(np.float32) * 0.1))
    out.update({"shared.init." + k: v.detach().numpy().copy() for k, v in layer.state_dict().items()})
    out["shared.param_names"] = np.array([n for n, _ in layer.named_parameters()])
    Xs = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).requires_grad_(True)
    ys = layer(Xs, A)
    ws = torch.from_numpy(rng.standard_normal((N, H)).astype(np.float32))
    (ys * ws).sum().backward()
    out["shared.X"], out["shared.w"], out["shared.Y"] = Xs.detach().numpy().copy(), ws.numpy().copy(), ys.detach().numpy().copy()
    out["shared.grad.X"] = Xs.grad.numpy().copy()
    out.update({"shared.grad." + n: q.grad.numpy().copy() for n, q in layer.named_parameters()})
    np.savez_compressed(os.path.join(HERE, "extras.npz"), **out)
    print(sorted(k for k in out if "init" not in k))


if __name__ == "__main__":
    main()
