"""Top-level model behind the reference's `MRGCN` interface (mrgcn/models/mrgcn.py:26-305):
same constructor, `module_dict` / `gate_weights` / `gate_map` / `devices` attributes and
state-dict keys; full-batch forward = gated literal encoders -> concatenation with X0 ->
R-GCN on the MI355X kernels.

Encoders: MLPs for the numeric / boolean / temporal datatypes, the TCNN for
`ogc.wktLiteral`, and the string / image heads on a backbone: either the reference's own hub config
list (`embedding_modules` as graph_features.py produces them: the backbone is then fetched with
`torch.hub.load` exactly as models/utils.py:32-44 does, ONE language model and ONE image model shared by
all encoding sets, mrgcn.py:83-105) or an `nn.Module` the caller supplies (a box without network).
In a mini-batch the encoders run for the outermost neighbours only."""
from __future__ import annotations

import logging
import os
import warnings

import torch
import torch.nn as nn

from .heads import ImageCNN, Normalizer, Transformer
from .perceptron import MLP
from .rgcn import RGCN
from .temporal_cnn import TCNN

logger = logging.getLogger(__name__)

_GATE_READBACK = os.environ.get("MRGCN_GATE_READBACK", "0") == "1"
_MLP_LAYERS = {"xsd.boolean": 1, "xsd.numeric": 1, "xsd.date": 2, "xsd.dateTime": 2, "xsd.gYear": 2}
_COUNTER_GROUP = {"xsd.boolean": "num", "xsd.numeric": "num", "xsd.date": "temp",
                  "xsd.dateTime": "temp", "xsd.gYear": "temp", "xsd.string": "llm", "xsd.anyURI": "llm",
                  "blob.image": "img", "ogc.wktLiteral": "geo"}


def loadFromHub(config):
    """models/utils.py:32-44: the entries of a hub config without '=' are torch.hub.load's positional arguments,
    `key=value` entries its keyword arguments (values stay strings, as in the reference)."""
    parameters, named_parameters = [], {}
    for param in config:
        if "=" not in param:
            parameters.append(param)
            continue
        key, value = param.split("=")
        named_parameters[key.strip()] = value.strip()
    return torch.hub.load(*parameters, **named_parameters)


def _backbone(config_or_module, shared, datatype):
    """The backbone behind a string / image head: a module as it is, or the hub config of the reference loaded once
    and shared (`shared`: the language / image model already loaded for an earlier encoding set, mrgcn.py:83-84, :95-96)."""
    if isinstance(config_or_module, nn.Module):
        return config_or_module
    if shared is not None:
        return shared
    if isinstance(config_or_module, (list, tuple)) and all(isinstance(x, str) for x in config_or_module):
        try:
            return loadFromHub(config_or_module)
        except Exception as e:  # noqa: BLE001  (no network, unknown repository, ...)
            raise RuntimeError(
                f"{datatype}: torch.hub.load{tuple(config_or_module)} failed ({type(e).__name__}: {e}); on a box without "
                "network pass the backbone nn.Module in place of the hub config (models/utils.py:32-44)") from e
    raise TypeError(f"{datatype}: expected a hub config (list of str) or an nn.Module, got {type(config_or_module).__name__}")


def _pick_device(want_gpu: bool):
    if want_gpu:
        if torch.cuda.is_available():
            return torch.device("cuda")
        warnings.warn("CUDA Resource not available", ResourceWarning)
    return torch.device("cpu")


class MRGCN(nn.Module):
    def __init__(self, modules, embedding_modules, num_relations, num_nodes, num_bases=-1,
                 p_dropout=0.0, featureless=False, bias=False, link_prediction=False,
                 gcn_gpu_acceleration=False, gated=True):
        super().__init__()
        assert len(modules) > 0

        self.num_nodes = num_nodes
        self.p_dropout = p_dropout
        self.module_dict = nn.ModuleDict()
        self.devices = dict()
        self.gate_map = dict()
        self.modality_modules = dict()
        self.modality_out_dim = 0
        self.compute_modality_embeddings = False
        self.im_norm = None

        counters = {"num": 0, "temp": 0, "llm": 0, "img": 0, "geo": 0}
        i_gate = 0
        language_model = image_model = None  # one of each, shared by every encoding set (mrgcn.py:43-44)
        for datatype, args, gpu_acceleration in embedding_modules:
            seq_length = -1
            if datatype in _MLP_LAYERS:
                ncols, dim_out, p_drop = args
                module = MLP(input_dim=ncols, output_dim=dim_out, num_layers=_MLP_LAYERS[datatype],
                             p_dropout=p_drop)
            elif datatype == "ogc.wktLiteral":  # mrgcn.py:108-118
                nrows, dim_out, model_size, p_drop = args
                module = TCNN(features_in=nrows, features_out=dim_out, p_dropout=p_drop, size=model_size)
                seq_length = module.minimal_length
            elif datatype in ("xsd.string", "xsd.anyURI", "blob.image"):  # mrgcn.py:80-107
                if datatype == "blob.image":
                    model_config, transform_config, dim_out, p_drop = args
                    backbone = _backbone(model_config, image_model, datatype)
                    if not isinstance(model_config, nn.Module):
                        image_model = backbone
                    module = ImageCNN(backbone, output_dim=dim_out, p_dropout=p_drop)
                    if "mean" in transform_config and "std" in transform_config:
                        self.im_norm = Normalizer(transform_config["mean"], transform_config["std"])
                else:
                    model_config, dim_out, p_drop = args
                    backbone = _backbone(model_config, language_model, datatype)
                    if not isinstance(model_config, nn.Module):
                        language_model = backbone
                    module = Transformer(backbone, output_dim=dim_out, p_dropout=p_drop)
            else:
                raise Exception("Datatype not supported: " + datatype)
            grp = _COUNTER_GROUP[datatype]  # booleans+numerics, temporal types, strings+URIs share counters
            mod_name = datatype.replace(".", "_") + "_" + str(counters[grp])
            counters[grp] += 1
            self.module_dict[mod_name] = module
            self.modality_modules.setdefault(datatype, []).append((module, seq_length, dim_out, i_gate))
            self.modality_out_dim += dim_out
            self.compute_modality_embeddings = True
            self.gate_map[mod_name] = i_gate
            i_gate += 1
            # MI355X-first: the (tiny) encoders live next to the R-GCN in HBM whenever a GPU is
            # present; the per-datatype flag only matters on a GPU-less host
            device = _pick_device(gpu_acceleration or torch.cuda.is_available())
            self.devices[datatype] = device
            module.to(device)

        # gates start at 0.1 so that the encoders' signal is damped at first (mrgcn.py:148-156)
        self.gate_weights = torch.ones(i_gate)
        if gated and i_gate > 0:
            self.gate_weights = nn.Parameter(self.gate_weights * 0.1)
        else:
            self.gate_weights.requires_grad = False

        self.rgcn = RGCN(modules, num_relations, num_nodes, num_bases, p_dropout, featureless, bias,
                         link_prediction)
        # The R-GCN of this package only computes on the GPU.  With the flag off the module is
        # still built (state dicts, CPU-side tooling) but moved to the GPU lazily at first use.
        device = _pick_device(gcn_gpu_acceleration or torch.cuda.is_available())
        self.devices["relational"] = device
        self.rgcn.to(device)
        self.X_device = device
        if isinstance(self.gate_weights, nn.Parameter):
            self.gate_weights.data = self.gate_weights.data.to(device)
        else:
            self.gate_weights = self.gate_weights.to(device)

    # ------------------------------------------------------------------------------
    def set_compute_dtype(self, dtype: str):
        """"f32" (the reference's arithmetic: the parity default) or "bf16" — BASELINE config 3's pipeline: the R-GCN
        layers store their activations in bf16 (`RGCN.set_operand_dtype`), the encoders' products (MLP / TCNN / heads)
        and the backbones run on the bf16 matrix cores with fp32 accumulation; parameters, BatchNorm statistics,
        gradients and the optimizer stay fp32.  The reference has no reduced precision anywhere (mrgcn.py:250-305)."""
        assert dtype in ("f32", "bf16")
        self.compute_dtype = dtype
        self.rgcn.set_operand_dtype(dtype)

    def forward(self, batch):
        from .. import dense
        prev = dense.set_matmul_dtype(getattr(self, "compute_dtype", "f32"))
        try:   # (the backward of every product runs with what its forward ran with: dense._Linear / _Conv1d)
            if type(batch).__name__ == "MiniBatch":
                return self._forward_mini_batch(batch)
            return self._forward_full_batch(batch)
        finally:
            dense.set_matmul_dtype(prev)

    def _forward_mini_batch(self, batch):
        """mrgcn.py:216-248: modality embeddings only for the outermost neighbours, then the
        mini-batch R-GCN."""
        dev = self.devices["relational"]
        X = None
        if self.compute_modality_embeddings:
            X0, F = batch.X[0], batch.X[1:]
            batch_idx = batch.A.neighbours[-1]
            XF = self._compute_modality_embeddings(F, batch_idx)
            X = XF.float() if X0.shape[1] == 0 else torch.cat([X0.to(dev), XF], dim=1).float()
        return self.rgcn(X, batch.A)

    def _forward_full_batch(self, batch):
        X0, F = batch.X[0], batch.X[1:]
        dev = self.devices["relational"]
        X = None
        if self.compute_modality_embeddings:
            # (every node, in order — the same tensor every epoch: a fresh 1.67 M-element arange per forward is a
            # parallel CPU op that wakes torch's whole intra-op pool each step)
            batch_idx = self.__dict__.get("_full_batch_idx")
            if batch_idx is None or batch_idx.numel() != self.num_nodes:
                batch_idx = self.__dict__["_full_batch_idx"] = torch.arange(self.num_nodes)
            XF = self._compute_modality_embeddings(F, batch_idx, full_batch=True)
            # (no given feature columns: the encoders' matrix is X — a concatenation would copy it once more)
            X = XF.float() if X0.shape[1] == 0 else torch.cat([X0.to(dev), XF], dim=1).float()
        return self.rgcn(X, batch.A)

    def _gate_decisions(self):
        """(which gates are zero as far as the host knows — a list of bools, or None: compute every set —,
        the gate vector with its zero entries masked on the device).

        mrgcn.py:263-266 skips a set whose gate `isclose` to zero; reading the gates back for that costs a host wait
        per forward (the reference: one per set) and cannot happen inside a captured epoch.  The test runs on the
        device instead and masks the gate vector: a zero gate's block of X is exactly zero and no gradient reaches the
        gate or its encoder — what the skip produces (a skipped encoder's parameters keep `.grad = None`, a masked
        one's get zeros: the same Adam step from zero moments) — without the host ever looking.  What is lost is the
        saving: a zero-gated set's encoder still runs (and an encoder output that is not finite would turn its block
        NaN where the skip would not have looked).  Gates start at 0.1 and are trained (mrgcn.py:150-154): one
        that is exactly zero was put there by hand; MRGCN_GATE_READBACK=1 then buys the skip back for one host wait per
        forward (outside captures).  Gates on the CPU are read directly."""
        g = self.gate_weights
        zero = torch.isclose(g.detach(), torch.zeros((), device=g.device))
        gates = torch.where(zero, torch.zeros_like(g), g)
        if not g.is_cuda:
            return zero.tolist(), gates                      # (no device to wait for)
        if _GATE_READBACK and not torch.cuda.is_current_stream_capturing():
            return zero.cpu().tolist(), gates
        return None, gates

    def _compute_modality_embeddings(self, F, batch_idx, full_batch=False):
        """XF[node, off:off+dim] = gate * encoder(encodings[node]) per encoding set
        (mrgcn.py:250-305).

        The reference resolves, per call and per set, which rows of the batch carry the encoding
        (`torch_intersect1d` + two `isin` masks) and reads every gate back to test it against zero
        (mrgcn.py:263-266: a zero gate's set is skipped).  Here no forward waits for the device:
        the zero test runs there and masks the gate vector (`_gate_decisions`).  A full batch resolves its sets once (the row positions
        are the node ids, the encodings do not change between epochs) and a mini-batch matches
        node ids on the device the encodings live on."""
        dev = self.devices["relational"]
        X = torch.zeros((len(batch_idx), self.modality_out_dim), dtype=torch.float32, device=dev)
        gates = self.gate_weights
        if isinstance(self.gate_weights, nn.Parameter):
            gate_is_zero, gates = self._gate_decisions()
        else:
            gate_is_zero = None             # ungated: constant ones (mrgcn.py:150-156)
        cache = self.__dict__.setdefault("_full_batch_sets", {}) if full_batch else None
        bidx = None
        offset = 0
        for datatype, encoding_sets, _ in F:
            if datatype not in self.modality_modules:
                continue
            for i, encoding_set in enumerate(encoding_sets):
                module, _, out_dim, i_gate = self.modality_modules[datatype][i]
                if gate_is_zero is not None and gate_is_zero[i_gate]:
                    offset += out_dim
                    continue
                encodings, node_idx, _ = encoding_set
                hit = cache.get((datatype, i)) if cache is not None else None
                if hit is not None and hit[0] is encodings and hit[1] is node_idx and hit[2] == len(batch_idx):
                    rows, data = hit[3], hit[4]
                else:
                    # rows of the batch that carry this encoding (mrgcn.py:276-277, :303); in a full
                    # batch the position equals the node id
                    nidx = torch.as_tensor(node_idx)
                    if bidx is None or bidx.device != nidx.device:
                        bidx = torch.as_tensor(batch_idx).to(nidx.device)
                    keep = torch.isin(nidx, bidx)
                    rows = torch.isin(bidx, nidx[keep]).nonzero().squeeze(1).to(dev)
                    if rows.numel() == 0:
                        rows = data = None
                    else:
                        data = encodings[keep.to(encodings.device)]
                        if datatype in ("xsd.string", "xsd.anyURI"):   # token ids (mrgcn.py:287-288)
                            data = data.int()
                        elif datatype != "blob.image":
                            data = data.float()
                    if cache is not None:
                        cache[(datatype, i)] = (encodings, node_idx, len(batch_idx), rows, data)
                if rows is None:            # no node of this batch has the datatype
                    offset += out_dim
                    continue
                gate = gates[i_gate]
                if datatype == "blob.image" and self.im_norm is not None:
                    data = self.im_norm.normalize_(data)
                elif datatype == "blob.image":
                    data = data.float()
                if X.is_cuda and any(not q.is_cuda for q in module.parameters()):
                    from .. import _lib
                    raise _lib.MrgcnError(f"{datatype}: the encoder lives on the CPU while the R-GCN runs on the GPU — "
                                          "mrgcn_amd has no CPU encoder path on a GPU box (move the module: "
                                          "batch.to(model.devices) / model.to(device))")
                if isinstance(module, MLP) and X.is_cuda and data.device == X.device and module.fused_ok(data):
                    # every Linear + ReLU, the gate and the scatter in one kernel (csrc/encoders.hip)
                    from .. import dense
                    lin = module.linears()
                    X = dense.mlp_gate_scatter(X, data, rows, gates, i_gate, offset,
                                               [l.weight for l in lin], [l.bias for l in lin], fresh=True)
                else:
                    out = module(data).to(dev) * gate.to(dev)
                    if X.is_cuda and out.dtype == X.dtype:
                        from .. import dense
                        X = dense.scatter_block(X, out, rows, offset)   # (X: this call's zero buffer)
                    else:
                        X[rows, offset:offset + out_dim] = out
                offset += out_dim
        return X
