"""Modality encoders feeding X against goldens captured from the reference
(tests/golden/make_encoder_goldens.py).  Weights are rebuilt from the same torch seed — which pins the
init stream and the state-dict layout — and checked against per-tensor checksums."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from tests import util

GOLD = os.path.join(os.path.dirname(__file__), "golden", "encoders.npz")


def _check_state(sd, g, prefix):
    keys = sorted(sd)
    assert keys == list(g[prefix + "keys"])
    np.testing.assert_array_equal([sd[k].numel() for k in keys], g[prefix + "numel"])
    np.testing.assert_allclose([float(sd[k].double().sum()) for k in keys], g[prefix + "sum"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose([float(sd[k].double().abs().sum()) for k in keys], g[prefix + "abs"], rtol=1e-9)


@pytest.mark.parametrize("size", ["S", "M", "L"])
def test_tcnn_matches_reference(size):
    from mrgcn_amd.models.temporal_cnn import TCNN
    g = np.load(GOLD)
    torch.manual_seed(3)
    m = TCNN(features_in=9, features_out=7, p_dropout=0.0, size=size)
    _check_state(m.state_dict(), g, f"tcnn{size}.sd.")          # same keys, same init stream
    assert m.minimal_length == int(g[f"tcnn{size}.minimal_length"])
    x = torch.from_numpy(g[f"tcnn{size}.x"])
    m.train()
    y = m(x)
    y.square().sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), g[f"tcnn{size}.y_train"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.conv[0].weight.grad.numpy(), g[f"tcnn{size}.grad_conv0"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(m.conv[1].running_mean.numpy(), g[f"tcnn{size}.sd_after.running_mean0"], rtol=1e-5, atol=1e-6)
    m.eval()
    np.testing.assert_allclose(m(x).detach().numpy(), g[f"tcnn{size}.y_eval"], rtol=1e-4, atol=1e-5)


class TinyImageNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.features = nn.Sequential(nn.Conv2d(3, 6, 3, padding=1), nn.ReLU(), nn.Conv2d(6, 8, 3, padding=1))
        self.classifier = nn.Linear(8, 5)


class TinyLM(nn.Module):
    def __init__(self):
        super().__init__()
        self.emb = nn.Embedding(50, 12)
        self.lin = nn.Linear(12, 12)

    def forward(self, ids):
        return (self.lin(self.emb(ids)),)


def _load(module, g, prefix):
    module.load_state_dict({k[len(prefix):]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith(prefix)})


def test_backbone_heads_and_normalizer_match_reference():
    from mrgcn_amd.models.heads import ImageCNN, Normalizer, Transformer
    g = np.load(GOLD, allow_pickle=True)
    base = TinyImageNet()
    _load(base, g, "img.base.")
    head = ImageCNN(base, output_dim=6, p_dropout=0.0)
    assert sorted(head.state_dict()) == sorted(k[len("img.sd."):] for k in g.files if k.startswith("img.sd."))
    _load(head, g, "img.sd.")
    np.testing.assert_allclose(head(torch.from_numpy(g["img.x"])).detach().numpy(), g["img.y"], rtol=1e-5, atol=1e-6)
    # the backbone is frozen, the head trains (imagecnn.py:18-20)
    assert sorted(n for n, p in head.named_parameters() if p.requires_grad) == list(g["img.trainable"])
    lm = TinyLM()
    _load(lm, g, "lm.base.")
    th = Transformer(lm, output_dim=4, p_dropout=0.0)
    _load(th, g, "lm.sd.")
    np.testing.assert_allclose(th(torch.from_numpy(g["lm.x"])).detach().numpy(), g["lm.y"], rtol=1e-5, atol=1e-6)
    nz = Normalizer([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    x = torch.from_numpy(g["norm.x"])
    np.testing.assert_allclose(nz.normalize_(x).numpy(), g["norm.y"], rtol=1e-6)
    np.testing.assert_allclose(nz.normalize_(x[0]).numpy(), g["norm.single"], rtol=1e-6)


def test_mrgcn_takes_the_reference_hub_configs(monkeypatch):
    """embedding_modules as graph_features.py:184-236 produces them for configs/am.toml: hub config lists for the
    string and image backbones.  loadFromHub (models/utils.py:32-44) splits positional from key=value entries; ONE
    language model and ONE image model are loaded and shared by all encoding sets (mrgcn.py:83-105)."""
    from mrgcn_amd.models import mrgcn as M
    calls = []

    def fake_hub_load(*args, **kwargs):
        calls.append((args, kwargs))
        return TinyLM() if args[1] == "model" else TinyImageNet()
    monkeypatch.setattr(torch.hub, "load", fake_hub_load)
    lm_cfg = ["huggingface/pytorch-transformers", "model", "distilbert-base-multilingual-cased", "force_reload = False"]
    im_cfg = ["pytorch/vision:v0.10.0", "mobilenet_v2", "MobileNet_V2_Weights.IMAGENET1K_V1"]
    tr_cfg = {"mode": "RGB", "mean": [0.485, 0.456, 0.406], "std": [0.229, 0.224, 0.225]}
    modules = [(5, 4, "mrgcn", nn.ReLU()), (4, 2, "mrgcn", None)]
    emb = [("blob.image", (im_cfg, tr_cfg, 6, 0.0), False), ("blob.image", (im_cfg, tr_cfg, 6, 0.0), False),
           ("xsd.numeric", (4, 4, 0.0), False),
           ("xsd.anyURI", (lm_cfg, 3, 0.0), False), ("xsd.string", (lm_cfg, 5, 0.0), False)]
    m = M.MRGCN(modules, emb, 3, 10)
    assert calls == [(tuple(im_cfg), {}), (tuple(lm_cfg[:3]), {"force_reload": "False"})]  # one load per modality
    assert set(m.gate_map) == {"blob_image_0", "blob_image_1", "xsd_numeric_0", "xsd_anyURI_0", "xsd_string_1"}
    assert m.module_dict["xsd_anyURI_0"].base_model is m.module_dict["xsd_string_1"].base_model
    assert m.im_norm is not None and m.modality_out_dim == 24
    # the reference's parameter names (tasks/utils.py:20-43 splits them on '.')
    names = [n for n, _ in m.named_parameters()]   # (the shared backbone is listed once, under its first head)
    assert any(n.startswith("module_dict.xsd_anyURI_0.base_model.") for n in names)
    assert "module_dict.xsd_string_1.fc.weight" in names


def test_mrgcn_reports_a_failed_hub_load_and_builds_from_modules(monkeypatch):
    from mrgcn_amd.models.mrgcn import MRGCN

    def no_network(*a, **k):
        raise OSError("no route to host")
    monkeypatch.setattr(torch.hub, "load", no_network)
    modules = [(5, 4, "mrgcn", nn.ReLU()), (4, 2, "mrgcn", None)]
    with pytest.raises(RuntimeError, match="torch.hub.load.*pass the backbone nn.Module"):
        MRGCN(modules, [("xsd.string", (["huggingface/pytorch-transformers", "model", "distilbert"], 5, 0.0), False)], 3, 10)
    with pytest.raises(TypeError):
        MRGCN(modules, [("xsd.string", (42, 5, 0.0), False)], 3, 10)
    with pytest.raises(Exception, match="Datatype not supported"):
        MRGCN(modules, [("xsd.unknown", (1, 1, 0.0), False)], 3, 10)
    m = MRGCN(modules, [("blob.image", (TinyImageNet(), {"mean": [0.5] * 3, "std": [0.2] * 3}, 3, 0.0), False),
                        ("xsd.anyURI", (TinyLM(), 2, 0.0), False)], 3, 10)
    assert set(m.gate_map) == {"blob_image_0", "xsd_anyURI_0"} and m.im_norm is not None and m.modality_out_dim == 5


@pytest.mark.skipif(torch.cuda.is_available(), reason="host-side logic on CPU tensors: the constructor places the "
                    "encoders on the GPU when one is present (covered there by the gpu-marked MRGCN tests)")
def test_modality_embeddings_resolution_full_batch_cache_and_mini_batch():
    """`MRGCN._compute_modality_embeddings` (mrgcn.py:250-305) on the host: which batch rows carry an encoding,
    the gate multiply, zero gates skipped; a full batch resolves its sets once and notices new tensors; a mini-batch
    (partial overlap with the set's nodes) matches a row-by-row restatement of the reference's masks."""
    from mrgcn_amd.models.mrgcn import MRGCN
    N, R = 40, 3
    torch.manual_seed(1)
    emb_cfg = sorted([("xsd.boolean", (2, 2, 0.0), False), ("xsd.numeric", (4, 3, 0.0), False)], key=lambda t: t[0])
    model = MRGCN([(5, 4, "mrgcn", None)], emb_cfg, R, N, num_bases=0, p_dropout=0.0, featureless=False, bias=False,
                  gcn_gpu_acceleration=False)
    model.devices["relational"] = torch.device("cpu")   # (the embeddings alone run anywhere; the R-GCN needs the GPU)
    model.gate_weights.data = model.gate_weights.data.cpu()
    rng = np.random.default_rng(3)
    num_idx = torch.from_numpy(np.sort(rng.choice(N, 12, replace=False)))
    boo_idx = torch.from_numpy(np.sort(rng.choice(N, 7, replace=False)))
    num, boo = torch.randn((12, 4)), torch.randn((7, 2))
    F = [["xsd.boolean", [[boo, boo_idx, None]], False], ["xsd.numeric", [[num, num_idx, None]], False]]

    def restated(batch_idx):
        X = torch.zeros((len(batch_idx), model.modality_out_dim))
        off = 0
        for dt, sets, _ in F:
            for i, (enc, nidx, _) in enumerate(sets):
                module, _, dim, ig = model.modality_modules[dt][i]
                if not torch.isclose(model.gate_weights[ig], torch.tensor(0.0)):
                    for pos, node in enumerate(batch_idx.tolist()):
                        hit = (nidx == node).nonzero()
                        if hit.numel():
                            X[pos, off:off + dim] = module(enc[hit[0, 0]][None].float())[0] * model.gate_weights[ig]
                off += dim
        return X

    full = torch.arange(N)
    with torch.no_grad():
        a = model._compute_modality_embeddings(F, full, full_batch=True)
        torch.testing.assert_close(a, restated(full))
        cached = dict(model._full_batch_sets)
        b = model._compute_modality_embeddings(F, full, full_batch=True)
        assert torch.equal(a, b) and all(model._full_batch_sets[k][3] is cached[k][3] for k in cached)   # resolved once
        num2 = torch.randn((12, 4))
        F[1][1][0][0] = num2                                                                      # new encodings
        c = model._compute_modality_embeddings(F, full, full_batch=True)
        torch.testing.assert_close(c, restated(full))
        assert model._full_batch_sets[("xsd.numeric", 0)][0] is num2
        # a mini-batch: ascending node ids, as the neighbour lists are (the reference pairs the k-th selected encoding
        # with the k-th selected batch position, mrgcn.py:276-303, which is the same node only in that order)
        mb = torch.from_numpy(np.sort(rng.permutation(N)[:17]))
        torch.testing.assert_close(model._compute_modality_embeddings(F, mb), restated(mb))
        none = torch.tensor([n for n in range(N) if n not in set(num_idx.tolist()) | set(boo_idx.tolist())][:5])
        assert float(model._compute_modality_embeddings(F, none).abs().sum()) == 0.0
        model.gate_weights.data[model.modality_modules["xsd.numeric"][0][3]] = 0.0                 # a closed gate
        d = model._compute_modality_embeddings(F, full, full_batch=True)
        torch.testing.assert_close(d, restated(full))


@pytest.mark.gpu
def test_mrgcn_with_tcnn_and_mlp_encoders_vs_reference():
    """MRGCN(FullBatch) with an ogc.wktLiteral (TCNN) and an xsd.numeric (MLP) encoder: state-dict
    layout + init stream, logits, gradients of the gates and of the TCNN's first convolution."""
    from mrgcn_amd.data.batch import FullBatch
    from mrgcn_amd.models.mrgcn import MRGCN
    g = np.load(GOLD, allow_pickle=True)
    _, A = util.load_graph("graph_small")
    N = A.shape[0]
    R = A.shape[1] // N
    torch.manual_seed(6)
    emb_cfg = sorted([("ogc.wktLiteral", (9, 5, "S", 0.0), False), ("xsd.numeric", (4, 3, 0.0), False)],
                     key=lambda t: t[0])
    modules = [(8, 6, "mrgcn", nn.ReLU()), (6, 4, "mrgcn", None)]
    model = MRGCN(modules, emb_cfg, R, N, num_bases=3, p_dropout=0.0, featureless=False, bias=False,
                  gcn_gpu_acceleration=True)
    _check_state({k: v.cpu() for k, v in model.state_dict().items()}, g, "mrgcn.sd.")
    X = [np.empty((N, 0), dtype=np.float32),
         ["ogc.wktLiteral", [[g["mrgcn.wkt"], g["mrgcn.wkt_idx"], np.full(14, 20)]], False],
         ["xsd.numeric", [[g["mrgcn.num"], g["mrgcn.num_idx"], np.ones(25, dtype=int)]], False]]
    batch = FullBatch(A, X, np.arange(N))
    batch.as_tensors_()
    batch.to(model.devices)
    logits = model(batch)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["mrgcn.logits"], rtol=1e-4, atol=1e-4)
    logits.square().mean().backward()
    np.testing.assert_allclose(model.gate_weights.grad.cpu().numpy(), g["mrgcn.grad.gate_weights"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(model.module_dict["ogc_wktLiteral_0"].conv[0].weight.grad.cpu().numpy(),
                               g["mrgcn.grad.tcnn_conv0"], rtol=2e-3, atol=1e-6)


@pytest.mark.gpu
def test_mrgcn_epoch_with_encoders_replays_from_a_hipgraph():
    """A full-batch MRGCN epoch (TCNN + MLP encoders, gates, two R-GCN layers, CE, clip, Adam) captured by
    GraphedTrainStep follows the eager epochs of an identically initialised model: the encoder sets of a full batch
    are resolved once and the gates are not read back inside the capture."""
    from mrgcn_amd.data.batch import FullBatch
    from mrgcn_amd.models.mrgcn import MRGCN
    from mrgcn_amd.train import ClipAdam, GraphedTrainStep, train_step
    g = np.load(GOLD, allow_pickle=True)
    _, A = util.load_graph("graph_small")
    N = A.shape[0]
    R = A.shape[1] // N
    emb_cfg = sorted([("ogc.wktLiteral", (9, 5, "S", 0.0), False), ("xsd.numeric", (4, 3, 0.0), False)],
                     key=lambda t: t[0])
    modules = [(8, 6, "mrgcn", nn.ReLU()), (6, 4, "mrgcn", None)]
    X = [np.empty((N, 0), dtype=np.float32),
         ["ogc.wktLiteral", [[g["mrgcn.wkt"], g["mrgcn.wkt_idx"], np.full(14, 20)]], False],
         ["xsd.numeric", [[g["mrgcn.num"], g["mrgcn.num_idx"], np.ones(25, dtype=int)]], False]]
    idx = torch.arange(0, N, 3, device="cuda")
    y = (idx % 4).to(torch.int64)
    losses, gates = [], []
    for graphed in (False, True):
        torch.manual_seed(6)
        model = MRGCN(modules, emb_cfg, R, N, num_bases=3, p_dropout=0.0, featureless=False, bias=True,
                      gcn_gpu_acceleration=True)
        batch = FullBatch(A, X, np.arange(N))
        batch.as_tensors_()
        batch.to(model.devices)
        model.train()
        opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0, capturable=graphed)
        if graphed:
            step = GraphedTrainStep(model, lambda: model(batch), idx, y, opt, warmup=2)
            ls = [float(step()) for _ in range(3)]
        else:
            ls = [float(train_step(model, lambda: model(batch), idx, y, opt)) for _ in range(5)][2:]
        losses.append(ls)
        gates.append(model.gate_weights.detach().cpu().numpy().copy())
        assert ("ogc.wktLiteral", 0) in model._full_batch_sets and ("xsd.numeric", 0) in model._full_batch_sets
    np.testing.assert_allclose(losses[1], losses[0], rtol=2e-3)
    np.testing.assert_allclose(gates[1], gates[0], rtol=2e-3, atol=1e-5)
    assert losses[0][-1] < losses[0][0]


# ---- the same encoders on the HIP kernels (csrc/encoders.hip through mrgcn_amd.dense) ---------------------
@pytest.mark.gpu
@pytest.mark.parametrize("size", ["S", "M", "L"])
def test_tcnn_on_the_matrix_cores_matches_reference(size):
    """Every Conv1d as an implicit-im2col f32-MFMA product, the fully connected tail likewise: forward in train
    and eval mode, the gradient of the first convolution and BatchNorm's running mean vs the reference's values."""
    from mrgcn_amd.models.temporal_cnn import TCNN
    g = np.load(GOLD)
    torch.manual_seed(3)
    m = TCNN(features_in=9, features_out=7, p_dropout=0.0, size=size).cuda()
    x = torch.from_numpy(g[f"tcnn{size}.x"]).cuda()
    m.train()
    y = m(x)
    assert type(y.grad_fn).__name__ == "_LinearBackward"       # the HIP path ran
    y.square().sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"tcnn{size}.y_train"], rtol=1e-4, atol=1e-5)
    gref = g[f"tcnn{size}.grad_conv0"]  # through up to 15 convolutions and batch-statistics BatchNorms: fp32 sums in
    np.testing.assert_allclose(m.conv[0].weight.grad.cpu().numpy(), gref, rtol=2e-3,  # another order than the CPU's
                               atol=2e-4 * float(np.abs(gref).max()))
    np.testing.assert_allclose(m.conv[1].running_mean.cpu().numpy(), g[f"tcnn{size}.sd_after.running_mean0"],
                               rtol=1e-5, atol=1e-6)
    m.eval()
    np.testing.assert_allclose(m(x).detach().cpu().numpy(), g[f"tcnn{size}.y_eval"], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_backbone_heads_on_the_matrix_cores_match_reference():
    from mrgcn_amd.models.heads import ImageCNN, Transformer
    g = np.load(GOLD, allow_pickle=True)
    base = TinyImageNet()
    _load(base, g, "img.base.")
    head = ImageCNN(base, output_dim=6, p_dropout=0.0)
    _load(head, g, "img.sd.")
    head = head.cuda()
    y = head(torch.from_numpy(g["img.x"]).cuda())
    assert type(y.grad_fn).__name__ == "_LinearBackward"
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["img.y"], rtol=1e-5, atol=1e-6)
    lm = TinyLM()
    _load(lm, g, "lm.base.")
    th = Transformer(lm, output_dim=4, p_dropout=0.0)
    _load(th, g, "lm.sd.")
    th = th.cuda()
    np.testing.assert_allclose(th(torch.from_numpy(g["lm.x"]).cuda()).detach().cpu().numpy(), g["lm.y"], rtol=1e-5,
                               atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1, 1), (5, 3, 2), (70, 130, 65), (257, 64, 7), (33, 768, 16),
                                   # tiled form: K split + the ReLU pass behind it, ragged 64- and 128-row tiles
                                   (300, 1100, 200), (1000, 96, 520), (130, 2050, 70)])
@pytest.mark.parametrize("relu", [False, True])
def test_linear_on_the_matrix_cores_vs_float64(shape, relu):
    """dense.linear (f32 MFMA 16x16x4: exact fp32 arithmetic) forward and all three gradients, edge tiles included."""
    from mrgcn_amd import dense
    n, K, N = shape
    gen = torch.Generator("cuda").manual_seed(n + K)
    x = torch.randn((n, K), device="cuda", generator=gen, requires_grad=True)
    W = torch.randn((N, K), device="cuda", generator=gen, requires_grad=True)
    b = torch.randn((N,), device="cuda", generator=gen, requires_grad=True)
    w = torch.randn((n, N), device="cuda", generator=gen)
    y = dense.linear(x, W, b, relu=relu)
    (y * w).sum().backward()
    x64, W64, b64 = (t.detach().double().requires_grad_(True) for t in (x, W, b))
    r = x64 @ W64.t() + b64
    r = torch.relu(r) if relu else r
    (r * w.double()).sum().backward()
    tol = dict(rtol=1e-5, atol=1e-5 * max(1.0, K ** 0.5))
    torch.testing.assert_close(y.double(), r, **tol)
    for a, c in ((x, x64), (W, W64), (b, b64)):
        torch.testing.assert_close(a.grad.double(), c.grad, rtol=1e-5, atol=1e-5 * max(1.0, (n * K) ** 0.5))


@pytest.mark.gpu
@pytest.mark.parametrize("B,Cin,T,Cout,KW,pad", [(2, 3, 20, 5, 3, 1), (7, 9, 33, 64, 7, 3), (3, 64, 10, 128, 3, 0),
                                                 (1, 1, 2, 1, 2, 0), (5, 16, 3, 32, 3, 1),
                                                 # the TCNN S shapes of a 500-literal set: a narrow dW with a long
                                                 # reduction (split K), sequences shorter than a loader's run of 8
                                                 # positions, one-position outputs
                                                 (500, 9, 20, 64, 3, 1), (300, 64, 5, 128, 3, 1),
                                                 (600, 128, 2, 256, 2, 0),
                                                 # the TCNN M shapes on a smaller batch: 128-row tiles with ragged last
                                                 # tiles (M = 37 * 7 = 259 rows of dW), sequences that end inside a
                                                 # tile, the closing valid convolution (one output position)
                                                 (24, 37, 300, 64, 7, 3), (24, 64, 100, 128, 3, 1),
                                                 (40, 128, 33, 256, 3, 1), (90, 256, 3, 512, 3, 1),
                                                 (150, 512, 3, 1024, 3, 0),
                                                 # one position in, one out, padded: the transposed loaders with
                                                 # runs of one element
                                                 (64, 32, 1, 48, 3, 1)])
def test_conv1d_on_the_matrix_cores_vs_torch(B, Cin, T, Cout, KW, pad):
    from mrgcn_amd import dense
    gen = torch.Generator("cuda").manual_seed(B + T)
    x = torch.randn((B, Cin, T), device="cuda", generator=gen, requires_grad=True)
    W = torch.randn((Cout, Cin, KW), device="cuda", generator=gen, requires_grad=True)
    b = torch.randn((Cout,), device="cuda", generator=gen, requires_grad=True)
    y = dense.conv1d(x, W, b, padding=pad)
    w = torch.randn(y.shape, device="cuda", generator=gen)
    (y * w).sum().backward()
    x64, W64, b64 = (t.detach().double().cpu().requires_grad_(True) for t in (x, W, b))
    r = torch.nn.functional.conv1d(x64, W64, b64, padding=pad)
    (r * w.double().cpu()).sum().backward()
    torch.testing.assert_close(y.double().cpu(), r, rtol=1e-5, atol=1e-4)
    for a, c in ((x, x64), (W, W64), (b, b64)):   # (dW / db sum over batch x positions terms of unit scale)
        torch.testing.assert_close(a.grad.double().cpu(), c.grad, rtol=1e-5, atol=2e-4 * max(1.0, (B * T / 100) ** 0.5))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(70, 130, 65), (300, 1100, 200), (1000, 96, 520), (130, 2050, 70), (33, 768, 16),
                                   (5, 3, 2)])
def test_linear_on_the_bf16_matrix_cores(shape):
    """The bf16 pipeline's products (`dense.set_matmul_dtype("bf16")`, mrgcn_gemm_bf16mm_f32): fp32 operands rounded to
    bf16 as tiles are staged, v_mfma_f32_16x16x32_bf16, fp32 accumulation.  A tiled shape equals the float64 product of
    the ROUNDED operands to fp32 accuracy; every shape (the small ones run the exact fp32 kernel) stays within 1e-2 of
    the largest element of the unrounded float64 result — forward and all three gradients, whose products run with
    what the forward ran with even after the switch is back at "f32"."""
    from mrgcn_amd import dense
    n, K, N = shape
    gen = torch.Generator("cuda").manual_seed(n + K)
    x = torch.randn((n, K), device="cuda", generator=gen, requires_grad=True)
    W = torch.randn((N, K), device="cuda", generator=gen, requires_grad=True)
    b = torch.randn((N,), device="cuda", generator=gen, requires_grad=True)
    w = torch.randn((n, N), device="cuda", generator=gen)
    prev = dense.set_matmul_dtype("bf16")
    try:
        y = dense.linear(x, W, b, relu=False)   # (no ReLU in front of the gradient check: a unit at the kink flips whole)
        yr = dense.linear(x.detach(), W.detach(), b.detach(), relu=True)
    finally:
        dense.set_matmul_dtype(prev)
    assert dense.matmul_dtype() == "f32"
    (y * w).sum().backward()
    tiled = (n >= 48 and N >= 48) or K >= 1024
    xr, Wr = (t.detach().to(torch.bfloat16).double() for t in (x, W))
    if tiled:
        exact = xr @ Wr.t() + b.detach().double()
        tol = dict(rtol=1e-5, atol=1e-5 * max(1.0, K ** 0.5))
        torch.testing.assert_close(y.detach().double(), exact, **tol)
        torch.testing.assert_close(yr.double(), torch.relu(exact), **tol)
        assert float((y.detach().double() - (x.detach().double() @ W.detach().double().t() + b.detach().double())).abs().max()) > 0
    x64, W64, b64 = (t.detach().double().requires_grad_(True) for t in (x, W, b))
    r = x64 @ W64.t() + b64
    (r * w.double()).sum().backward()
    for name, got, ref in (("y", y, r), ("dx", x.grad, x64.grad), ("dW", W.grad, W64.grad), ("db", b.grad, b64.grad)):
        err = float((got.double() - ref).abs().max())
        assert err <= 1e-2 * float(ref.abs().max()) + 1e-6, (name, err, float(ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,Cin,T,Cout,KW,pad", [(500, 9, 20, 64, 3, 1), (300, 64, 5, 128, 3, 1), (600, 128, 2, 256, 2, 0),
                                                 (24, 37, 300, 64, 7, 3), (40, 128, 33, 256, 3, 1), (150, 512, 3, 1024, 3, 0),
                                                 (64, 32, 1, 48, 3, 1), (2, 3, 20, 5, 3, 1)])
def test_conv1d_on_the_bf16_matrix_cores(B, Cin, T, Cout, KW, pad):
    """The TCNN's convolutions with bf16 matrix-core arithmetic (every loader mode of the tiled product: im2col rows,
    its transpose, the y[b][n][t] operand and output) against torch's float64 convolution: forward and gradients within
    1e-2 of the result's largest element."""
    from mrgcn_amd import dense
    gen = torch.Generator("cuda").manual_seed(B + T)
    x = torch.randn((B, Cin, T), device="cuda", generator=gen, requires_grad=True)
    W = torch.randn((Cout, Cin, KW), device="cuda", generator=gen, requires_grad=True)
    b = torch.randn((Cout,), device="cuda", generator=gen, requires_grad=True)
    prev = dense.set_matmul_dtype("bf16")
    try:
        y = dense.conv1d(x, W, b, padding=pad)
    finally:
        dense.set_matmul_dtype(prev)
    w = torch.randn(y.shape, device="cuda", generator=gen)
    (y * w).sum().backward()
    x64, W64, b64 = (t.detach().double().cpu().requires_grad_(True) for t in (x, W, b))
    r = torch.nn.functional.conv1d(x64, W64, b64, padding=pad)
    (r * w.double().cpu()).sum().backward()
    for name, got, ref in (("y", y, r), ("dx", x.grad, x64.grad), ("dW", W.grad, W64.grad), ("db", b.grad, b64.grad)):
        err = float((got.double().cpu() - ref).abs().max())
        assert err <= 1e-2 * float(ref.abs().max()) + 1e-6, (name, err, float(ref.abs().max()))


@pytest.mark.gpu
def test_products_cut_along_the_batch_equal_the_single_launch(monkeypatch):
    """Operands beyond the tiled product's 2^29-byte reach are multiplied piece by piece along the batch
    (dense._batch_pieces): forced here with a small limit, the results equal the single launch."""
    from mrgcn_amd import dense
    gen = torch.Generator("cuda").manual_seed(4)

    def run():
        x = torch.randn((37, 16, 40), device="cuda", generator=gen, requires_grad=True)
        W = torch.randn((24, 16, 3), device="cuda", generator=gen, requires_grad=True)
        b = torch.randn((24,), device="cuda", generator=gen, requires_grad=True)
        y = dense.conv1d(x, W, b, padding=1)
        z = dense.linear(y.transpose(1, 2).reshape(-1, 24), W.reshape(24, 48)[:, :24].contiguous(), None, relu=True)
        (y.square().sum() + z.sum()).backward()
        return [t.detach().clone() for t in (y, z, x.grad, W.grad, b.grad)]

    gen.manual_seed(4)
    ref = run()
    monkeypatch.setattr(dense, "_MM_MAX_BYTES", 16 * 40 * 4 * 6)   # ~5 samples per piece
    assert len(dense._batch_pieces(37, 16 * 40, 24 * 40)) > 5
    gen.manual_seed(4)
    got = run()
    # outputs: the same sums in the same order.  Gradients of W / b: sums over 37 x 40 (sample, position) terms whose
    # split-K partial tiles meet through float atomics — the order of the pieces differs from launch to launch, so
    # the slack is the sibling test's rounding bound, not 1e-4 (seen: one element of 1 152 off by 4.9e-4 at 30)
    for i, (a, c) in enumerate(zip(got, ref)):
        if i < 2:
            torch.testing.assert_close(a, c, rtol=1e-5, atol=1e-4)
        else:
            torch.testing.assert_close(a, c, rtol=2e-5, atol=2e-4 * (37 * 40 / 100) ** 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,with_rows", [([1, 4], True), ([6, 4, 3], True), ([16, 16, 16, 16, 16], False),
                                            ([3, 1], False)])
def test_fused_mlp_gate_scatter_vs_torch(dims, with_rows):
    """The literal encoder as ONE kernel (every Linear + ReLU, the gate, the scatter into XF) against the nn.Linear
    path: output block, untouched remainder of XF, and the gradients of weights, biases and the gate."""
    from mrgcn_amd import dense
    gen = torch.Generator("cuda").manual_seed(sum(dims))
    n, NX, off = 1000, 1500, 3
    enc = torch.rand((n, dims[0]), device="cuda", generator=gen) * 2 - 1
    rows = torch.randperm(NX, device="cuda", generator=gen)[:n] if with_rows else None
    Ws = [torch.rand((o, i), device="cuda", generator=gen, requires_grad=True) for i, o in zip(dims, dims[1:])]
    bs = [torch.rand((o,), device="cuda", generator=gen, requires_grad=True) for o in dims[1:]]
    gates = torch.tensor([0.3, 0.1, 0.7], device="cuda", requires_grad=True)
    width = off + dims[-1] + 2
    XF = torch.full((NX if with_rows else n, width), 5.0, device="cuda")
    out = dense.mlp_gate_scatter(XF, enc, rows, gates, 1, off, Ws, bs)
    w = torch.randn(out.shape, device="cuda", generator=gen)
    (out * w).sum().backward()
    got = [t.grad.clone() for t in Ws + bs + [gates]]
    for t in Ws + bs + [gates]:
        t.grad = None
    h = enc
    for W, b in zip(Ws, bs):
        h = torch.relu(h @ W.t() + b)
    ref = torch.full_like(XF, 5.0)
    idx = rows if rows is not None else torch.arange(n, device="cuda")
    ref[idx, off:off + dims[-1]] = h * gates[1]
    (ref * w).sum().backward()
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    for a, t in zip(got, Ws + bs + [gates]):
        torch.testing.assert_close(a, t.grad, rtol=2e-4, atol=2e-4 * float(t.grad.abs().max()) + 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,T,kind,arg", [(5, 7, 20, 0, 0), (4, 64, 300, 1, 3), (3, 33, 21, 1, 2), (6, 16, 11, 2, 3),
                                            (2, 5, 9, 2, 2), (1, 3, 4, 2, 4)])
def test_batchnorm_relu_pool_block_vs_torch(B, C, T, kind, arg):
    """csrc/tcnn.hip through the C ABI against nn.BatchNorm1d -> ReLU -> MaxPool1d / AdaptiveMaxPool1d (float64 on
    the CPU): output, d x / d gamma / d beta, the running statistics after the step, and the eval-mode pass.
    (6, 16, 11, adaptive 3): overlapping adaptive windows; (3, 33, 21, max 2): a tail position no window covers."""
    from mrgcn_amd import dense
    torch.manual_seed(B * 100 + T)
    x = torch.randn(B, C, T) * 1.5 + 0.3
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C))          # negative scales too: ReLU and max do not commute with them
        bn.bias.copy_(torch.randn(C) * 0.5)
    ref_bn = torch.nn.BatchNorm1d(C).double()
    ref_bn.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    pool = {0: torch.nn.Identity(), 1: torch.nn.MaxPool1d(arg, arg) if kind == 1 else None,
            2: torch.nn.AdaptiveMaxPool1d(arg) if kind == 2 else None}[kind]
    xr = x.double().requires_grad_(True)
    yr = pool(torch.relu(ref_bn(xr)))
    w = torch.randn_like(yr)
    (yr * w).sum().backward()

    bn = bn.cuda()
    xg = x.cuda().requires_grad_(True)
    y = dense.bn_relu_pool(xg, bn, kind, arg)
    assert type(y.grad_fn).__name__ == "_BnReluPoolBackward"
    (y * w.float().cuda()).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=1e-4, atol=1e-5)
    for got, want, name in ((xg.grad, xr.grad, "dx"), (bn.weight.grad, ref_bn.weight.grad, "dgamma"),
                            (bn.bias.grad, ref_bn.bias.grad, "dbeta")):
        want = want.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(want).max()),
                                   err_msg=name)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), ref_bn.running_mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), ref_bn.running_var.numpy(), rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    bn.eval(); ref_bn.eval()
    xe = x.cuda().requires_grad_(True)
    ye = dense.bn_relu_pool(xe, bn, kind, arg)
    xre = x.double().requires_grad_(True)
    yre = pool(torch.relu(ref_bn(xre)))
    np.testing.assert_allclose(ye.detach().cpu().numpy(), yre.detach().numpy(), rtol=1e-4, atol=1e-5)
    (ye * w.float().cuda()).sum().backward()
    (yre * w).sum().backward()
    np.testing.assert_allclose(xe.grad.cpu().numpy(), xre.grad.numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.gpu
def test_mlp_beyond_the_fused_kernel_runs_on_hip_kernels_too():
    """perceptron.py:6-46 outside the limits of the one-kernel form (layer widths > 16; dropout active in training):
    on the GPU every Linear + ReLU is the matrix-core GEMM of csrc/encoders.hip — never torch.nn's — and equals the
    same module in float64 on the host: values, input gradient, weight gradients.  With dropout the HIP products
    still run (the autograd graph shows them), a fraction p of the outputs is zeroed and the rest scaled by
    1 / (1 - p), like Linear -> Dropout -> ReLU."""
    import copy
    from mrgcn_amd import dense
    from mrgcn_amd.models.perceptron import MLP
    torch.manual_seed(5)
    m = MLP(input_dim=40, output_dim=22, num_layers=3, p_dropout=0.0).cuda()
    assert not m.fused_ok(torch.zeros((3, 40), device="cuda"))           # too wide for the fused kernel
    with torch.no_grad():
        for l in m.linears():                                             # (U(0,1) weights blow up 40 -> 22: rescale)
            l.weight.mul_(0.1)
    x = (torch.randn((500, 40), device="cuda") * 0.5).requires_grad_(True)
    y = m(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()

    def nodes(fn, seen=None):
        seen = set() if seen is None else seen
        if fn is None or fn in seen:
            return seen
        seen.add(fn)
        for nxt, _ in fn.next_functions:
            nodes(nxt, seen)
        return seen
    assert sum(type(f).__name__.startswith("_Linear") for f in nodes(y.grad_fn)) == 3
    ref = copy.deepcopy(m).cpu().double()
    x64 = x.detach().cpu().double().requires_grad_(True)
    y64 = ref.mlp(x64)
    (y64 * w.cpu().double()).sum().backward()
    torch.testing.assert_close(y.detach().cpu().double(), y64.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(x.grad.cpu().double(), x64.grad, rtol=1e-3, atol=1e-4)
    for a, b in zip(m.linears(), [l for l in ref.mlp if isinstance(l, nn.Linear)]):
        torch.testing.assert_close(a.weight.grad.cpu().double(), b.weight.grad, rtol=1e-3, atol=1e-3)
    # dropout active: HIP products + torch's mask
    md = MLP(input_dim=8, output_dim=4, num_layers=2, p_dropout=0.5).cuda().train()
    assert not md.fused_ok(torch.zeros((3, 8), device="cuda"))
    xd = torch.rand((20000, 8), device="cuda") + 0.5                      # positive inputs and weights: no ReLU zeros
    yd = md(xd)
    assert sum(type(f).__name__.startswith("_Linear") for f in nodes(yd.grad_fn)) == 2
    assert 0.45 < float((yd == 0).float().mean()) < 0.55
    md.eval()
    assert md.fused_ok(xd) and float((md(xd) == 0).float().mean()) == 0.0


@pytest.mark.gpu
def test_a_cpu_encoder_next_to_a_gpu_rgcn_is_refused():
    """No silent host detour on a GPU box: MRGCN refuses an encoder whose parameters live on the CPU."""
    from mrgcn_amd import _lib
    from mrgcn_amd.models.mrgcn import MRGCN
    model = MRGCN([(3, 4, "mrgcn", None)], [("xsd.numeric", (4, 3, 0.0), False)], 3, 20, num_bases=0, featureless=False)
    model.module_dict["xsd_numeric_0"].cpu()
    F = [["xsd.numeric", [[torch.randn((5, 4), device="cuda"), torch.arange(5, device="cuda"), None]], False]]
    with pytest.raises(_lib.MrgcnError, match="CPU"):
        model._compute_modality_embeddings(F, torch.arange(20), full_batch=True)


@pytest.mark.gpu
def test_conv_bias_gradient_handed_over_by_the_batchnorm_block(monkeypatch):
    """Conv1d -> BatchNorm1d/ReLU/pool: the block's backward leaves the per-channel sums of its dx on the gradient
    tensor and the convolution's backward takes them as its bias gradient (no second pass over dx) — equal to the sum
    computed from dx, and not used when the gradient tensor is another one."""
    from mrgcn_amd import dense
    gen = torch.Generator("cuda").manual_seed(9)
    taken = []
    real = dense._chan_sum_of

    def spy(dy, C):
        r = real(dy, C)
        taken.append(r is not None)
        return r

    monkeypatch.setattr(dense, "_chan_sum_of", spy)
    for (B, Cin, T, Cout, kind, arg) in [(6, 5, 40, 7, 1, 3), (9, 4, 3, 130, 0, 0), (3, 3, 70, 64, 2, 3)]:
        x = torch.randn((B, Cin, T), device="cuda", generator=gen, requires_grad=True)
        W = torch.randn((Cout, Cin, 3), device="cuda", generator=gen, requires_grad=True)
        b = torch.randn((Cout,), device="cuda", generator=gen, requires_grad=True)
        bn = torch.nn.BatchNorm1d(Cout).cuda().train()
        y = dense.conv1d(x, W, b, padding=1)
        y.retain_grad()
        z = dense.bn_relu_pool(y, bn, kind, arg)
        (z * torch.randn(z.shape, device="cuda", generator=gen)).sum().backward()
        assert taken and taken[-1]                                     # the hand-over happened
        torch.testing.assert_close(b.grad, y.grad.sum(dim=(0, 2)), rtol=1e-4, atol=1e-4)
        assert real(y.grad.clone(), Cout) is None                      # another tensor carries nothing


@pytest.mark.gpu
def test_zero_gates_are_masked_on_the_device_without_a_host_read():
    """mrgcn.py:263-266 skips a set whose gate is zero.  On the GPU the gate vector is masked on the device instead
    (MRGCN._gate_decisions): the set's block of X is exactly zero, the gate and the set's encoder get zero gradient,
    the other sets are untouched — with the fused MLP path and with the module path — and nothing synchronises."""
    from mrgcn_amd.models.mrgcn import MRGCN
    N, R = 300, 3
    torch.manual_seed(2)
    emb_cfg = sorted([("xsd.boolean", (2, 2, 0.0), False), ("xsd.numeric", (4, 3, 0.0), False)], key=lambda t: t[0])
    model = MRGCN([(5, 4, "mrgcn", None)], emb_cfg, R, N, num_bases=0, p_dropout=0.0, featureless=False, bias=False,
                  gcn_gpu_acceleration=True)
    assert model.gate_weights.is_cuda and model.gate_weights.requires_grad
    rng = np.random.default_rng(3)
    num_idx = torch.from_numpy(np.sort(rng.choice(N, 120, replace=False)))
    boo_idx = torch.from_numpy(np.sort(rng.choice(N, 70, replace=False)))
    num, boo = torch.randn((120, 4)).cuda(), torch.randn((70, 2)).cuda()
    F = [["xsd.boolean", [[boo, boo_idx, None]], False], ["xsd.numeric", [[num, num_idx, None]], False]]
    full = torch.arange(N)
    ig_num = model.modality_modules["xsd.numeric"][0][3]
    ig_boo = model.modality_modules["xsd.boolean"][0][3]
    off_num = model.modality_modules["xsd.boolean"][0][2]
    open_X = model._compute_modality_embeddings(F, full, full_batch=True).detach().clone()
    assert float(open_X[:, off_num:].abs().sum()) > 0
    with torch.no_grad():
        model.gate_weights[ig_num] = 0.0
    decided, gates = model._gate_decisions()
    assert decided is None and float(gates[ig_num]) == 0.0 and float(gates[ig_boo]) == float(model.gate_weights[ig_boo])
    X = model._compute_modality_embeddings(F, full, full_batch=True)
    assert float(X[:, off_num:].abs().sum()) == 0.0
    assert torch.equal(X[:, :off_num], open_X[:, :off_num])
    X.square().sum().backward()
    assert float(model.gate_weights.grad[ig_num]) == 0.0 and float(model.gate_weights.grad[ig_boo]) != 0.0
    for q in model.modality_modules["xsd.numeric"][0][0].parameters():
        assert q.grad is None or float(q.grad.abs().sum()) == 0.0
    assert any(float(q.grad.abs().sum()) > 0 for q in model.modality_modules["xsd.boolean"][0][0].parameters())
