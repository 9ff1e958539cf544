"""Reference behaviours that no shipped config switches on (goldens: tests/golden/make_extra_goldens.py):
node dropout p > 0 (mrgcn/models/rgcn.py:78-84) and shared basis coefficients (mrgcn/layers/graph.py:42-44)."""
import os

import numpy as np
import pytest
import torch

from tests import util

G = np.load(os.path.join(util.GOLDEN, "extras.npz"))
N, R = int(G["meta.N"]), int(G["meta.R"])
K, H, C, B = 6, 8, 3, 3


def _adjacency():
    g, A = util.load_graph("graph_small")
    return util.coo_tensor(A, "norm_f32", "cuda")


def test_shared_bases_layer_has_the_reference_parameter_set():
    """weight_F_comp IS weight_I_comp: one parameter under two state-dict keys (CPU: names and init stream)."""
    from mrgcn_amd.layers.graph import GraphConvolution
    torch.manual_seed(41)
    layer = GraphConvolution(K, H, R, N, num_bases=B, bias=True, input_layer=True, featureless=False,
                             shared_bases_weights=True)
    assert layer.weight_F_comp is layer.weight_I_comp
    assert [n for n, _ in layer.named_parameters()] == list(G["shared.param_names"])
    sd = layer.state_dict()
    keys = sorted(k[len("shared.init."):] for k in G.files if k.startswith("shared.init."))
    assert sorted(sd) == keys
    for k in keys:
        if k != "b":
            np.testing.assert_array_equal(sd[k].numpy(), G["shared.init." + k], err_msg=k)


@pytest.mark.gpu
def test_shared_bases_layer_forward_and_gradients():
    from mrgcn_amd.layers.graph import GraphConvolution
    layer = GraphConvolution(K, H, R, N, num_bases=B, bias=True, input_layer=True, featureless=False,
                             shared_bases_weights=True)
    layer.load_state_dict({k[len("shared.init."):]: torch.from_numpy(np.array(G[k])) for k in G.files
                           if k.startswith("shared.init.")})
    layer = layer.cuda()
    A = _adjacency()
    X = torch.from_numpy(G["shared.X"]).cuda().requires_grad_(True)
    for engine in ("fused", "literal"):
        layer.engine = engine
        layer.zero_grad(set_to_none=True)
        X.grad = None
        Y = layer(X, A)
        np.testing.assert_allclose(Y.detach().cpu().numpy(), G["shared.Y"], rtol=1e-4, atol=1e-4)
        (Y * torch.from_numpy(G["shared.w"]).cuda()).sum().backward()
        np.testing.assert_allclose(X.grad.cpu().numpy(), G["shared.grad.X"], rtol=1e-3, atol=1e-5)
        for n, p in layer.named_parameters():  # the shared coefficients collect both terms' gradients
            np.testing.assert_allclose(util.ref_layout(p.grad, n).cpu().numpy(), G["shared.grad." + n], rtol=1e-3,
                                       atol=1e-5, err_msg=f"{n} ({engine})")


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["fused", "literal"])
def test_node_dropout_matches_the_reference(engine):
    """Same CPU seed, same masks (one Bernoulli draw per node and layer, kept nodes scaled by 1 / (1 - p)), in
    train and in eval mode (the reference's functional dropout ignores the mode); gradients
    flow through the masked rows only."""
    from mrgcn_amd.models.rgcn import RGCN
    p = float(G["drop.p"])
    model = RGCN([(K, H, "mrgcn", torch.nn.ReLU()), (H, C, "mrgcn", None)], R, N, B, p, False, True, False)
    model.load_state_dict({k[len("drop.init."):]: torch.from_numpy(np.array(G[k])) for k in G.files
                           if k.startswith("drop.init.")})
    model = model.cuda()
    model.set_engine(engine)
    A = _adjacency()
    X = torch.from_numpy(G["drop.X"]).cuda()
    for mode in ("train", "eval"):
        getattr(model, mode)()
        torch.manual_seed(77)
        with torch.no_grad():
            got = model(X, A).cpu().numpy()
        np.testing.assert_allclose(got, G[f"drop.logits_{mode}"], rtol=1e-4, atol=1e-4)
    assert not np.allclose(G["drop.logits_train"][G["drop.mask1"] == 0], 1e9)  # (dropped rows exist)
    assert (G["drop.mask1"] == 0).any() and np.abs(G["drop.logits_train"][G["drop.mask1"] == 0]).max() == 0.0
    model.train()
    torch.manual_seed(77)
    Xg = X.clone().requires_grad_(True)
    (model(Xg, A) * torch.from_numpy(G["drop.w"]).cuda()).sum().backward()
    np.testing.assert_allclose(Xg.grad.cpu().numpy(), G["drop.grad.X"], rtol=1e-3, atol=1e-5)
    for n, q in model.named_parameters():
        np.testing.assert_allclose(util.ref_layout(q.grad, n).cpu().numpy(), G["drop.grad." + n], rtol=1e-3, atol=1e-5,
                                   err_msg=n)
