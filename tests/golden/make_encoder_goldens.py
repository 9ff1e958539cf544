#!/usr/bin/env python
"""Golden vectors for the modality encoders feeding X, AUTHORING CONTAINER ONLY.
Imports the reference and records, for seeded random weights and inputs,
  mrgcn/models/temporal_cnn.py:6-156   TCNN sizes S / M / L: state dict, forward (train-mode BatchNorm
                                       and eval mode), gradient of the first conv weight
  mrgcn/models/imagecnn.py:9-41        ImageCNN head on a tiny stand-in backbone (with `.classifier`)
  mrgcn/models/transformer.py:8-38     Transformer head on a tiny stand-in backbone (returns a tuple)
  mrgcn/encodings/blob/image.py:139-166 image Normalizer
  mrgcn/models/mrgcn.py:250-305        MRGCN with an ogc.wktLiteral (TCNN) + xsd.numeric encoder, full batch
    python tests/golden/make_encoder_goldens.py"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402


class TinyImageNet(nn.Module):  # stand-in for a torchvision backbone: children up to `.classifier`
    def __init__(self):
        super().__init__()
        self.features = nn.Sequential(nn.Conv2d(3, 6, 3, padding=1), nn.ReLU(), nn.Conv2d(6, 8, 3, padding=1))
        self.classifier = nn.Linear(8, 5)

    def forward(self, x):
        return self.classifier(self.features(x).mean((2, 3)))


class TinyLM(nn.Module):  # stand-in for a HF encoder: forward(ids) -> (hidden_states,)
    def __init__(self):
        super().__init__()
        self.emb = nn.Embedding(50, 12)
        self.lin = nn.Linear(12, 12)

    def forward(self, ids):
        return (self.lin(self.emb(ids)),)


def checksums(prefix, sd):
    """Weights are NOT stored (tens of MB): the tests rebuild them from the same torch seed — which
    also pins the init stream — and compare per-tensor checksums."""
    keys = sorted(sd)
    return {prefix + "keys": np.asarray(keys),
            prefix + "sum": np.asarray([float(sd[k].double().sum()) for k in keys]),
            prefix + "abs": np.asarray([float(sd[k].double().abs().sum()) for k in keys]),
            prefix + "numel": np.asarray([sd[k].numel() for k in keys])}


def main():
    ref = mg.import_reference()
    from mrgcn.models.temporal_cnn import TCNN
    from mrgcn.models.imagecnn import ImageCNN
    from mrgcn.models.transformer import Transformer
    from mrgcn.encodings.blob.image import Normalizer
    out = {}
    rng = np.random.default_rng(5)
    for size, L in (("S", 20), ("M", 300), ("L", 300)):
        torch.manual_seed(3)
        m = TCNN(features_in=9, features_out=7, p_dropout=0.0, size=size)
        x = torch.from_numpy(rng.standard_normal((4, 9, L)).astype(np.float32))
        out.update(checksums(f"tcnn{size}.sd.", m.state_dict()))
        m.train()
        y = m(x)
        y.square().sum().backward()
        out[f"tcnn{size}.x"] = x.numpy()
        out[f"tcnn{size}.y_train"] = y.detach().numpy().copy()
        out[f"tcnn{size}.grad_conv0"] = m.conv[0].weight.grad.numpy().copy()
        out[f"tcnn{size}.sd_after.running_mean0"] = m.conv[1].running_mean.numpy().copy()
        m.eval()
        out[f"tcnn{size}.y_eval"] = m(x).detach().numpy().copy()
        out[f"tcnn{size}.minimal_length"] = np.int64(m.minimal_length)
        print("tcnn", size, y.shape, len(m.state_dict()))
    torch.manual_seed(4)
    base = TinyImageNet()
    out.update(mg.state_to_np("img.base.", base.state_dict()))
    head = ImageCNN(base, output_dim=6, p_dropout=0.0)
    out.update(mg.state_to_np("img.sd.", head.state_dict()))
    xi = torch.from_numpy(rng.standard_normal((3, 3, 10, 10)).astype(np.float32))
    out["img.x"], out["img.y"] = xi.numpy(), head(xi).detach().numpy().copy()
    out["img.trainable"] = np.asarray(sorted(n for n, p in head.named_parameters() if p.requires_grad))
    torch.manual_seed(5)
    lm = TinyLM()
    out.update(mg.state_to_np("lm.base.", lm.state_dict()))
    th = Transformer(lm, output_dim=4, p_dropout=0.0)
    out.update(mg.state_to_np("lm.sd.", th.state_dict()))
    ids = torch.from_numpy(rng.integers(0, 50, (5, 7)))
    out["lm.x"], out["lm.y"] = ids.numpy(), th(ids).detach().numpy().copy()
    nz = Normalizer([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    im = torch.from_numpy(rng.integers(0, 256, (2, 3, 4, 4)).astype(np.float32))
    out["norm.x"], out["norm.y"] = im.numpy(), nz.normalize_(im).numpy().copy()
    out["norm.single"] = nz.normalize_(im[0]).numpy().copy()

    # MRGCN with a TCNN (wkt) and an MLP (numeric) encoder, full batch, 50-node golden graph
    g = np.load(os.path.join(HERE, "graph_small.npz"))
    N, P = int(g["num_nodes"]), int(g["num_pred"])
    R = 2 * P + 1
    A_csr = mg.reference_adjacency(ref, g["triples"], N, P)
    torch.manual_seed(6)
    wkt_idx = np.sort(rng.choice(N, 14, replace=False))
    num_idx = np.sort(rng.choice(N, 25, replace=False))
    wkt = rng.standard_normal((14, 9, 20)).astype(np.float32)
    num = rng.standard_normal((25, 4)).astype(np.float32)
    emb_cfg = sorted([("ogc.wktLiteral", (9, 5, "S", 0.0), False), ("xsd.numeric", (4, 3, 0.0), False)],
                     key=lambda t: t[0])
    modules = [(8, 6, "mrgcn", nn.ReLU()), (6, 4, "mrgcn", None)]
    model = ref.mrgcn.MRGCN(modules, emb_cfg, R, N, num_bases=3, p_dropout=0.0, featureless=False, bias=False)
    X = [np.empty((N, 0), dtype=np.float32),
         ["ogc.wktLiteral", [[wkt, wkt_idx, np.full(14, 20)]], False],
         ["xsd.numeric", [[num, num_idx, np.ones(25, dtype=int)]], False]]
    batch = ref.batch.FullBatch(A_csr, X, np.arange(N))
    batch.as_tensors_()
    out.update(checksums("mrgcn.sd.", model.state_dict()))
    logits = model(batch)
    logits.square().mean().backward()
    out["mrgcn.wkt"], out["mrgcn.wkt_idx"], out["mrgcn.num"], out["mrgcn.num_idx"] = wkt, wkt_idx, num, num_idx
    out["mrgcn.logits"] = logits.detach().numpy().copy()
    out["mrgcn.grad.gate_weights"] = model.gate_weights.grad.numpy().copy()
    out["mrgcn.grad.tcnn_conv0"] = model.module_dict["ogc_wktLiteral_0"].conv[0].weight.grad.numpy().copy()
    print("mrgcn keys", len(model.state_dict()), "gate_map", model.gate_map)
    np.savez_compressed(os.path.join(HERE, "encoders.npz"), **out)
    print(os.path.getsize(os.path.join(HERE, "encoders.npz")))


if __name__ == "__main__":
    main()
