"""Heads that sit on a pretrained backbone (reference: mrgcn/models/imagecnn.py:9-41,
mrgcn/models/transformer.py:8-38, mrgcn/models/utils.py:10-58) and the image normaliser
(mrgcn/encodings/blob/image.py:139-166).  The reference fetches the backbones with `torch.hub.load`;
this package has no network path, so the backbone is an `nn.Module` the caller supplies (parity of
pretrained weights is unpinned) — the heads, freezing rule and pooling are the reference's."""
import numpy as np
import torch
import torch.nn as nn


def freeze_(model, layer="", _grad=False):
    for name, param in model.named_parameters():
        if layer in name:
            param.requires_grad_(_grad)


def unfreeze_(model, layer=""):
    freeze_(model, layer, _grad=True)


def stripClassifier(model):
    """The children of `model` up to (excluding) its `.classifier`, as a Sequential."""
    kept = []
    for module in model.children():
        if module is model.classifier:
            break
        kept.append(module)
    return nn.Sequential(*kept)


def inferOutputDim(model):
    """Width of the last Linear / Conv2d of the model."""
    for module in reversed(list(model.modules())):
        if isinstance(module, nn.Linear):
            return module.out_features
        if isinstance(module, nn.Conv2d):
            return module.out_channels
    return -1


def _widen(t):
    """a backbone output that came out of the bf16 autocast goes on in fp32 (the heads' parameters are fp32)"""
    return t.float() if t.dtype in (torch.bfloat16, torch.float16) else t


class _Head(nn.Module):
    def __init__(self, base_model, inter_dim, output_dim, p_dropout, bias, finetune):
        super().__init__()
        self.module_dict = nn.ModuleDict()
        self.finetune = finetune
        self.base_model = base_model
        if self.finetune:  # (sic) the reference freezes the backbone when `finetune` is set
            freeze_(self.base_model)
        self.pre_fc = nn.Linear(inter_dim, inter_dim, bias=bias)
        self.fc = nn.Linear(inter_dim, output_dim, bias=bias)
        self.f_activation = nn.ReLU()

    def _backbone(self, X):
        """The (frozen or fine-tuned) backbone.  With the bf16 pipeline on (`dense.matmul_dtype() == "bf16"`: BASELINE
        config 3) it runs under torch's bf16 autocast — its convolutions and products on the bf16 matrix cores, fp32
        parameters; the reference has no reduced precision (imagecnn.py:9-41, transformer.py:8-38)."""
        from .. import dense
        if dense.matmul_dtype() == "bf16" and torch.is_tensor(X) and X.is_cuda:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return self.base_model(X)
        return self.base_model(X)

    def _project(self, pooled):
        """pre_fc -> ReLU -> dropout -> fc (imagecnn.py:31-41, transformer.py:29-38); on the GPU the two products
        run on the matrix cores with bias / ReLU in their epilogues (dense.linear, csrc/encoders.hip)."""
        from .. import dense
        if dense.usable(pooled, self.pre_fc.weight):
            out = dense.linear(pooled, self.pre_fc.weight, self.pre_fc.bias, relu=True)
            if self.dropout is not None:
                out = self.dropout(out)
            return dense.linear(out, self.fc.weight, self.fc.bias)
        out = self.f_activation(self.pre_fc(pooled))
        if self.dropout is not None:
            out = self.dropout(out)
        return self.fc(out)


class ImageCNN(_Head):
    def __init__(self, model, output_dim, p_dropout=0.2, bias=True, finetune=True):
        base = stripClassifier(model)
        super().__init__(base, inferOutputDim(base), output_dim, p_dropout, bias, finetune)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.dropout = nn.Dropout(p=p_dropout) if p_dropout > 0 else None

    def forward(self, X):
        return self._project(_widen(torch.flatten(self.avgpool(self._backbone(X)), 1)))


class Transformer(_Head):
    def __init__(self, model, output_dim, p_dropout=0.2, bias=True, finetune=True):
        super().__init__(model, inferOutputDim(model), output_dim, p_dropout, bias, finetune)
        self.dropout = nn.Dropout(p=p_dropout) if p_dropout > 0 else None

    def forward(self, X):
        hidden_state = self._backbone(X)[0]      # (batch, seq_len, dim)
        return self._project(_widen(hidden_state[:, 0]))  # first token


class Normalizer:
    """Per-channel (x - mean) / std on pixel values; means / stds given in [0, 1] are scaled by 255."""

    def __init__(self, mean_values, std_values, convert_float_to_pixel=True):
        self.mean_values = np.array(mean_values, dtype=np.float64)
        self.std_values = np.array(std_values, dtype=np.float64)
        if convert_float_to_pixel:
            self.mean_values *= 255
            self.std_values *= 255

    def _stats(self, device):
        # one upload per device, kept: a host -> device copy per call would also be illegal inside a hipGraph capture
        cache = self.__dict__.setdefault("_dev", {})
        ent = cache.get(device)
        if ent is None:
            ent = cache[device] = (torch.as_tensor(self.mean_values, device=device)[:, None, None],
                                   torch.as_tensor(self.std_values, device=device)[:, None, None])
        return ent

    def normalize_(self, im):
        mean, std = self._stats(im.device)
        return ((im - mean) / std).float()  # broadcasts over a leading batch dimension
