"""BASELINE configs 1, 2 and 4 at (or near) full shape: the MI355X engine against the literal
ATen port of the reference's epoch (oracle/aten_literal.py, pinned to the reference's goldens)
running on the host — logits within 1e-4 (north_star), loss per epoch, with the reference's
int8 boundary cast of the adjacency (`ref_int8`)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_pair(name, scale, dims, B, featureless, value_mode, relu_last=False, epochs=3, labelled=None,
              l1=0.0, l2=0.0):
    from mrgcn_amd import synth
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.train import ClipAdam, train_step
    from oracle import aten_literal as AL
    g = synth.make_graph(name, seed=5, scale=scale, value_mode=value_mode)
    N, R = g.num_nodes, g.num_relations
    rng = np.random.default_rng(5)
    X = None if featureless else rng.standard_normal((N, dims[0][0])).astype(np.float32)
    n_lab = labelled or 200
    idx = np.sort(rng.choice(N, n_lab, replace=False)).astype(np.int64)
    y = rng.integers(0, dims[-1][1], n_lab).astype(np.int64)

    # host side: the reference's op sequence
    p = AL.make_params(dims, R, N, B, False, featureless, seed=11)
    init = {k: v.detach().clone() for k, v in p.items()}
    A_cpu = AL.coo_tensor(g.rows, g.cols, g.vals, (N, R * N))
    ep = AL.Epoch(p, len(dims), R, N, B, featureless, relu_last=relu_last, l1_lambda=l1, l2_lambda=l2)
    Xc = None if X is None else torch.from_numpy(X)
    ref = [ep.step(Xc, A_cpu, torch.from_numpy(idx), torch.from_numpy(y)) for _ in range(epochs)]

    # MI355X side, same initial parameters
    modules = [(i, o, "mrgcn", torch.nn.ReLU() if (li < len(dims) - 1 or relu_last) else None)
               for li, (i, o) in enumerate(dims)]
    model = RGCN(modules, R, N, B, 0.0, featureless, False, False)
    model.load_state_dict({k: v for k, v in init.items()})
    model = model.cuda()
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals),
                                (N, R * N)).cuda()
    assert A.dtype == (torch.int8 if value_mode == "ref_int8" else torch.float32)
    Xg = None if X is None else torch.from_numpy(X).cuda()
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0)
    ig, yg = torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda()
    for e in range(epochs):
        logits = model(Xg, A).detach().cpu().numpy()
        np.testing.assert_allclose(logits, ref[e][0].detach().numpy(), rtol=1e-4, atol=1e-4,
                                   err_msg=f"{name}: logits before epoch {e}")
        loss = train_step(model, lambda: model(Xg, A), ig, yg, opt, l1_lambda=l1, l2_lambda=l2)
        np.testing.assert_allclose(float(loss), float(ref[e][1]), rtol=2e-4, atol=2e-5)


def test_config1_aifb_structure_only_full_shape():
    # configs/aifb.toml: featureless, no bases, 2 layers -> 16 -> 4
    _run_pair("aifb", 1.0, [(0, 16), (16, 4)], 0, True, "ref_int8", labelled=176)


def test_config2_mutag_with_literal_features_full_shape():
    # configs/mutag.toml: 30 bases, 16 hidden, 2 classes; X = 8 encoder output columns
    _run_pair("mutag", 1.0, [(8, 16), (16, 2)], 30, False, "ref_int8", labelled=340)


def test_config2_mutag_normalised_values():
    _run_pair("mutag", 1.0, [(8, 16), (16, 2)], 30, False, "norm_f32", labelled=340)


def test_config4_fb15k_encoder_quarter_shape():
    # configs/fb15k-237.toml: single featureless layer, hidden 200, 2 bases, ReLU on it
    _run_pair("fb15k", 0.25, [(0, 200)], 2, True, "ref_int8", relu_last=True, epochs=2, labelled=300)


def test_l1_l2_regularisation_terms():
    # node_classification.py:172-188: penalties over parameters whose name contains 'weight'
    _run_pair("aifb", 0.5, [(5, 8), (8, 4)], 3, False, "norm_f32", labelled=100, l1=1e-4, l2=1e-3)
