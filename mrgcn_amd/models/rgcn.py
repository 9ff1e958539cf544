"""Stack of R-GCN layers behind the reference's `RGCN` interface (mrgcn/models/rgcn.py:12-132):
same constructor, `layers` / `activations` ModuleDicts keyed `layer_<i>`, `relations` table for
link prediction, `num_layers`.  Full-batch only: the ReLU between layers is folded into the
epilogue of the layer's sparse product instead of running as its own pass."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn.functional import dropout

from ..layers.graph import GraphConvolution
from ..plan import plan_of


class RGCN(nn.Module):
    def __init__(self, modules, num_relations, num_nodes, num_bases, p_dropout, featureless, bias,
                 link_prediction):
        super().__init__()
        assert len(modules) > 0

        self.num_nodes = num_nodes
        self.p_dropout = p_dropout
        self.layers = nn.ModuleDict()
        self.activations = nn.ModuleDict()
        for i, (indim, outdim, _ltype, f_activation) in enumerate(modules):
            first = i == 0  # rgcn.py:30-37: only layer 0 is an input layer / may be featureless
            self.layers[f"layer_{i}"] = GraphConvolution(
                indim=indim, outdim=outdim, num_relations=num_relations, num_nodes=num_nodes,
                num_bases=num_bases, featureless=featureless if first else False,
                input_layer=first, bias=bias)
            self.activations[f"layer_{i}"] = f_activation
        self.num_layers = len(self.layers)

        if link_prediction:
            # DistMult diagonal relation embeddings (rgcn.py:55-61)
            self.relations = nn.Parameter(torch.empty((num_relations, modules[-1][1])))
            self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.relations)

    def set_engine(self, engine: str):
        for layer in self.layers.values():
            layer.engine = engine

    def set_operand_dtype(self, dtype: str):
        assert dtype in ("f32", "bf16")
        for layer in self.layers.values():
            layer.operand_dtype = dtype

    def forward(self, X, A):
        if not isinstance(A, torch.Tensor):
            raise NotImplementedError("mini-batch A_Batch input (rgcn.py:91-128) is outside the "
                                      "full-batch path of mrgcn_amd")
        return self._forward_full_batch(X, A)

    def _forward_full_batch(self, X, A):
        for key, layer in self.layers.items():
            f_activation = self.activations[key] if key in self.activations else None
            fuse_relu = (isinstance(f_activation, nn.ReLU) and self.p_dropout <= 0.0
                         and layer.engine == "fused")
            if fuse_relu:
                plan = plan_of(A, layer.num_nodes, layer.num_relations)
                X = layer._forward_fused(X, plan, relu=True)
                continue
            X = layer(X, A)
            if self.p_dropout > 0.0:
                # node dropout: one Bernoulli draw per node, applied regardless of train/eval
                # mode and drawn on the CPU, as rgcn.py:78-84 does
                ones = dropout(torch.ones(self.num_nodes), p=self.p_dropout).to(X.device)
                X = X * ones.unsqueeze(1)
            if f_activation is not None:
                X = f_activation(X)
        return X
