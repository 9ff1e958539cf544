"""BASELINE configs 1, 2 and 4 at (or near) full shape: the MI355X engine against the literal
ATen port of the reference's epoch (oracle/aten_literal.py, pinned to the reference's goldens)
running on the host — logits within 1e-4 (north_star), loss per epoch, with the reference's
int8 boundary cast of the adjacency (`ref_int8`)."""
import numpy as np
import pytest
import torch

from tests import util

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


def test_config2_mutag_full_shape_with_the_encoders_in_front():
    """BASELINE config 2/3 with the modality encoders actually feeding `X` (not random columns): MRGCN(FullBatch) at
    the full MUTAG shape (23 644 nodes, 47 relations, 30 bases) with an xsd.numeric MLP over 6 000 literal nodes
    and an ogc.wktLiteral TCNN over 500, through `FullBatch.as_tensors_() / .to()`.  Reference on the host: the same
    encoder modules in float64 (torch CPU), the gate multiply + scatter of mrgcn.py:250-305 written out, then the
    float64 R-GCN oracle; logits at 1e-4, loss, and the gradients that travel back through X into the encoders
    (gates, first MLP layer, first TCNN convolution) via the oracle's dX."""
    import copy
    import scipy.sparse as sp
    from mrgcn_amd import synth
    from mrgcn_amd.data.batch import FullBatch
    from mrgcn_amd.models.mrgcn import MRGCN
    from oracle import rgcn_oracle as O
    g = synth.make_graph("mutag", seed=5, scale=1.0, value_mode="norm_f32")
    N, R, B = g.num_nodes, g.num_relations, 30
    rng = np.random.default_rng(8)
    num_idx = np.sort(rng.choice(N, 6000, replace=False))
    num = rng.standard_normal((6000, 4)).astype(np.float32)
    wkt_idx = np.sort(rng.choice(N, 500, replace=False))
    wkt = (rng.random((500, 9, 20)) < 0.15).astype(np.float32)
    torch.manual_seed(12)
    emb_cfg = sorted([("ogc.wktLiteral", (9, 5, "S", 0.0), False), ("xsd.numeric", (4, 3, 0.0), False)],
                     key=lambda t: t[0])
    modules = [(8, 16, "mrgcn", torch.nn.ReLU()), (16, 2, "mrgcn", None)]
    model = MRGCN(modules, emb_cfg, R, N, num_bases=B, p_dropout=0.0, featureless=False, bias=True,
                  gcn_gpu_acceleration=True)
    A = sp.csr_matrix((g.vals, (g.rows, g.cols)), shape=(N, R * N))
    X = [np.empty((N, 0), dtype=np.float32),
         ["ogc.wktLiteral", [[wkt, wkt_idx, np.full(500, 20)]], False],
         ["xsd.numeric", [[num, num_idx, np.ones(6000, dtype=int)]], False]]
    batch = FullBatch(A, X, np.arange(N), value_mode="norm_f32")
    batch.as_tensors_()
    batch.to(model.devices)
    model.train()   # batch-statistics BatchNorm in the TCNN, as in a training epoch
    logits = model(batch)
    idx = np.sort(rng.choice(N, 340, replace=False))
    y = rng.integers(0, 2, 340)
    from mrgcn_amd.train import categorical_crossentropy
    loss = categorical_crossentropy(logits, torch.from_numpy(idx).cuda(), torch.from_numpy(y).cuda())
    loss.backward()

    # ---- host reference -------------------------------------------------------------------------------------
    tcnn = copy.deepcopy(model.module_dict["ogc_wktLiteral_0"]).cpu().double().train()
    mlp = copy.deepcopy(model.module_dict["xsd_numeric_0"]).cpu().double().train()
    for m in tcnn.modules():
        if isinstance(m, torch.nn.BatchNorm1d):   # the GPU forward above already moved the running statistics
            m.momentum = 0.0
    gates = model.gate_weights.detach().cpu().double().requires_grad_(True)
    XF = torch.zeros((N, 8), dtype=torch.float64)
    off = 0
    for name, mod, enc, nidx in (("ogc.wktLiteral", tcnn, wkt, wkt_idx), ("xsd.numeric", mlp, num, num_idx)):
        _, _, out_dim, i_gate = model.modality_modules[name][0]
        out = mod(torch.from_numpy(enc).double())
        XF = XF.index_put((torch.from_numpy(nidx)[:, None], torch.arange(off, off + out_dim)[None, :]),
                          gates[i_gate] * out)
        off += out_dim
    cfgs = [O.LayerCfg(8, 16, R, N, B, bias=True, input_layer=True, featureless=False),
            O.LayerCfg(16, 2, R, N, B, bias=True, input_layer=False, featureless=False)]
    sd = {k: v.detach().cpu().numpy() for k, v in model.rgcn.state_dict().items()}
    params = [{k.split(".", 2)[2]: v for k, v in sd.items() if k.startswith(f"layers.layer_{i}.")} for i in range(2)]
    A64 = sp.csr_matrix((g.vals.astype(np.float64), (g.rows, g.cols)), shape=(N, R * N))
    Xn = XF.detach().numpy()
    pre0, c0 = O.layer_forward(cfgs[0], params[0], Xn, A64)
    H = np.maximum(pre0, 0)
    pre1, c1 = O.layer_forward(cfgs[1], params[1], H, A64)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), pre1, rtol=1e-4, atol=1e-4)
    z = pre1[idx] - pre1[idx].max(1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(1, keepdims=True))
    want_loss = -logp[np.arange(340), y].mean()
    np.testing.assert_allclose(float(loss), want_loss, rtol=2e-5, atol=1e-6)
    dlog = np.zeros_like(pre1)
    sm = np.exp(logp)
    sm[np.arange(340), y] -= 1.0
    np.add.at(dlog, idx, sm / 340)
    g1, dH = O.layer_backward(cfgs[1], params[1], H, A64, dlog, c1)
    g0, dX = O.layer_backward(cfgs[0], params[0], Xn, A64, dH * (pre0 > 0), c0)
    XF.backward(torch.from_numpy(dX))
    got_w = util.ref_layout(model.rgcn.layers["layer_0"].weight_F.grad, "weight_F").cpu().numpy()
    np.testing.assert_allclose(got_w, g0["weight_F"], rtol=2e-3, atol=2e-6 * max(1.0, np.abs(g0["weight_F"]).max()))
    for got, want, name in ((model.gate_weights.grad, gates.grad, "gates"),
                            (model.module_dict["xsd_numeric_0"].mlp[0].weight.grad, mlp.mlp[0].weight.grad, "mlp"),
                            (model.module_dict["ogc_wktLiteral_0"].conv[0].weight.grad, tcnn.conv[0].weight.grad,
                             "tcnn conv0")):
        want = want.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=5e-3, atol=5e-4 * float(np.abs(want).max()) + 1e-9,
                                   err_msg=name)


@pytest.mark.parametrize("value_mode", ["ref_int8", "norm_f32"])
def test_config4_fb15k_encoder_full_shape_at_sampled_rows(value_mode):
    """configs/fb15k-237.toml:85-97 at the FULL shape (N = 14 541, R = 475, one featureless layer -> 200 with ReLU,
    2 bases; the reference's `W_I` would be 5.5 GB): embeddings of 300 sampled rows (the largest hubs among them)
    against the float64 oracle on their receptive field, and the gradients of `weight_I_comp`, `b` and `relations`
    of a loss that lives on those rows against the oracle's backward restricted to them
    (oracle.input_term_comp_grad_at_rows, pinned to the reference's autograd in tests/test_oracle_golden.py).
    F = 200 takes the wide-row product (k_spmm<G >= 16>) and the scalar mix kernels."""
    import scipy.sparse as sp
    from mrgcn_amd import synth
    from mrgcn_amd.models.rgcn import RGCN
    from oracle import rgcn_oracle as O
    g = synth.make_graph("fb15k", seed=5, scale=1.0, value_mode=value_mode)
    N, R, B, F = g.num_nodes, g.num_relations, 2, 200
    assert (N, R) == (14541, 475)
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals),
                                (N, R * N)).cuda()
    torch.manual_seed(4)
    model = RGCN([(0, F, "mrgcn", torch.nn.ReLU())], R, N, B, 0.0, True, True, True).cuda()
    with torch.no_grad():
        model.layers["layer_0"].b.normal_(0.0, 0.1)      # (zeros at init: make the bias term visible)
    deg = np.bincount(g.rows, minlength=N)
    rng = np.random.default_rng(2)
    rows = np.unique(np.concatenate([np.argsort(deg)[-20:], rng.choice(N, 280, replace=False)]))
    sel = torch.from_numpy(rows).cuda()
    E = model(None, A)
    w = torch.randn((len(rows), F), device="cuda", generator=torch.Generator("cuda").manual_seed(7))
    (E[sel] * w).sum().backward()
    A_csr = sp.csr_matrix((g.vals.astype(np.float64), (g.rows, g.cols)), shape=(N, R * N))
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    cfgs = O.rgcn_cfgs([(0, F)], R, N, B, True, True)
    params = O.split_params(state, 1)
    ref = O.rgcn_forward_at_rows(cfgs, params, None, A_csr, rows, relu_last=True)
    got = E.detach()[sel].cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
    dpre = w.cpu().numpy().astype(np.float64) * (ref > 0)
    want_comp = O.input_term_comp_grad_at_rows(cfgs[0], params[0], A_csr, rows, dpre)
    got_comp = model.layers["layer_0"].weight_I_comp.grad.cpu().numpy()
    np.testing.assert_allclose(got_comp, want_comp, rtol=1e-3, atol=1e-4 * float(np.abs(want_comp).max()) + 1e-9)
    np.testing.assert_allclose(model.layers["layer_0"].b.grad.cpu().numpy(), dpre.sum(0), rtol=1e-3,
                               atol=1e-4 * float(np.abs(dpre.sum(0)).max()))
    assert model.relations.grad is None     # the decoder's table is not on this loss's path


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_config3_am_quarter_full_multimodal_with_the_encoders_in_front(compute):
    """BASELINE config 3 "full multimodal": MRGCN(FullBatch) on an AM/4-shaped graph (417 k nodes, R = 267, 40 bases)
    with every kind of encoder in front of the R-GCN instead of random feature columns — an image head on a
    (stand-in) CNN backbone with the pixel normaliser, a string head on a (stand-in) language model, a WKT TCNN and
    a numeric MLP, gated and scattered into X (mrgcn.py:250-305) through `FullBatch.as_tensors_() / .to()`.
    Reference on the host: the same encoder modules in float64, the gate multiply + scatter written out, then the
    float64 R-GCN oracle on the receptive field of 200 sampled rows (oracle.rgcn_forward_at_rows).  Logits 1e-4.
    `compute` = "bf16": the same model through `MRGCN.set_compute_dtype("bf16")` — the config's named precision: bf16
    activations in the R-GCN layers (X read as bf16 rows), the encoders' products and the backbones on the bf16 matrix
    cores, fp32 accumulation and parameters — against the SAME float64 reference: logits within 2e-2 of the largest
    (SURVEY 8d's stated tolerance; the reference itself has no reduced precision)."""
    import copy
    import mrgcn_amd
    import scipy.sparse as sp
    from mrgcn_amd import synth
    from mrgcn_amd.data.batch import FullBatch
    from mrgcn_amd.models.mrgcn import MRGCN
    from oracle import rgcn_oracle as O
    from tests.test_encoders import TinyImageNet, TinyLM
    g = synth.make_graph("am", seed=3, scale=0.25, value_mode="norm_f32")
    N, R, B = g.num_nodes, g.num_relations, 40
    rng = np.random.default_rng(8)

    def nodes(k):
        return np.sort(rng.choice(N, k, replace=False))
    img_idx, str_idx, wkt_idx, num_idx = nodes(3000), nodes(20000), nodes(800), nodes(60000)
    img = rng.integers(0, 256, (3000, 3, 12, 12)).astype(np.uint8)
    toks = rng.integers(1, 50, (20000, 16)).astype(np.int64)
    wkt = (rng.random((800, 9, 20)) < 0.15).astype(np.float32)
    num = rng.standard_normal((60000, 4)).astype(np.float32)
    torch.manual_seed(12)
    emb_cfg = sorted([("blob.image", (TinyImageNet(), {"mean": [0.485, 0.456, 0.406], "std": [0.229, 0.224, 0.225]}, 6, 0.0), False),
                      ("xsd.string", (TinyLM(), 4, 0.0), False),
                      ("ogc.wktLiteral", (9, 5, "S", 0.0), False),
                      ("xsd.numeric", (4, 3, 0.0), False)], key=lambda t: t[0])
    W = 6 + 5 + 3 + 4
    modules = [(W, 10, "mrgcn", torch.nn.ReLU()), (10, 11, "mrgcn", None)]
    model = MRGCN(modules, emb_cfg, R, N, num_bases=B, p_dropout=0.0, featureless=False, bias=True,
                  gcn_gpu_acceleration=True)
    model.set_compute_dtype(compute)
    A = sp.csr_matrix((g.vals, (g.rows, g.cols)), shape=(N, R * N))
    enc = {"blob.image": (img, img_idx, np.ones(3000, dtype=int)), "ogc.wktLiteral": (wkt, wkt_idx, np.full(800, 20)),
           "xsd.numeric": (num, num_idx, np.ones(60000, dtype=int)), "xsd.string": (toks, str_idx, np.full(20000, 16))}
    X = [np.empty((N, 0), dtype=np.float32)] + [[dt, [list(enc[dt])], False] for dt, _, _ in emb_cfg]
    batch = FullBatch(A, X, np.arange(N), value_mode="norm_f32")
    batch.as_tensors_()
    batch.to(model.devices)
    model.eval()        # (running statistics in the TCNN's BatchNorm: a deterministic forward on both sides)
    mrgcn_amd.reset_stats()
    with torch.no_grad():
        logits = model(batch)
    assert (mrgcn_amd.stats().get("bf16.xform_xbf16") == 1) == (compute == "bf16"), mrgcn_amd.stats()
    # ---- host reference: the same encoders in float64, gate * output scattered into X --------------------------
    XF = np.zeros((N, W))
    gates = model.gate_weights.detach().cpu().double()
    off = 0
    for dt, _, _ in emb_cfg:
        module, _, out_dim, i_gate = model.modality_modules[dt][0]
        ref_mod = copy.deepcopy(module).cpu().double().eval()
        data, nidx, _ = enc[dt]
        with torch.no_grad():
            if dt == "blob.image":
                x = model.im_norm.normalize_(torch.from_numpy(data)).double()
            elif dt == "xsd.string":
                x = torch.from_numpy(data).int()
            else:
                x = torch.from_numpy(data).double()
            out = ref_mod(x)
        XF[nidx, off:off + out_dim] = (gates[i_gate] * out).numpy()
        off += out_dim
    deg = np.bincount(g.rows, minlength=N)
    rows = np.unique(np.concatenate([np.argsort(deg)[-20:], rng.choice(N, 180, replace=False),
                                     rng.choice(img_idx, 20), rng.choice(wkt_idx, 20)]))
    A64 = sp.csr_matrix((g.vals.astype(np.float64), (g.rows, g.cols)), shape=(N, R * N))
    state = {k: v.detach().cpu().numpy() for k, v in model.rgcn.state_dict().items()}
    cfgs = O.rgcn_cfgs([(W, 10), (10, 11)], R, N, B, True, False)
    ref = O.rgcn_forward_at_rows(cfgs, O.split_params(state, 2), XF, A64, rows)
    got = logits[torch.from_numpy(rows).cuda()].cpu().numpy()
    if compute == "bf16":
        err = float(np.abs(got - ref).max())
        assert 1e-6 < err <= 2e-2 * float(np.abs(ref).max()), (err, float(np.abs(ref).max()))
    else:
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
    assert float(np.abs(XF).max()) > 0 and len(np.unique(np.nonzero(XF)[1])) == W   # every encoder contributed
