"""Stack of R-GCN layers behind the reference's `RGCN` interface (mrgcn/models/rgcn.py:12-132):
same constructor, `layers` / `activations` ModuleDicts keyed `layer_<i>`, `relations` table for
link prediction, `num_layers`.  In full-batch mode the ReLU between layers is folded into the
epilogue of the layer's sparse product; mini-batch mode (`A_Batch` input) walks the row slices."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn.functional import dropout

from ..layers.graph import GraphConvolution
from ..plan import build_plans_parallel, plan_of


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
        # hidden layers keep their output in rows padded to whole 16-byte pieces (plan.spmm: the product stores whole
        # pieces); the model's own output is dense like the reference's
        for key in list(self.layers)[:-1]:
            self.layers[key].padded_output = True

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
        if not isinstance(A, torch.Tensor):  # A_Batch (rgcn.py:63-67)
            return self._forward_mini_batch(X, A)
        return self._forward_full_batch(X, A)

    def _forward_mini_batch(self, X, A):
        """rgcn.py:91-128: layer l computes the embeddings of the nodes (L-1-l) hops from the batch
        nodes out of those one hop further, on the matching row slice of A."""
        from ..data.batch import A_BatchMasked, getAdjacencyNodeColumnIdx
        if isinstance(A, A_BatchMasked):
            return self._forward_masked(X, A)
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            from .. import _lib
            raise _lib.MrgcnError("the mini-batch forward cannot be captured into a hipGraph (its backward decides by "
                                  "counts read back from the device); GraphedTrainStep is for full-batch steps")
        self.prepare_batch(A)
        for layer_idx, (key, layer) in enumerate(self.layers.items()):
            f_activation = self.activations[key] if key in self.activations else None
            i = self.num_layers - (layer_idx + 1)
            A_slices = A.row[i]
            if layer.input_layer and layer.featureless:
                X = layer(None, A_slices)
            else:
                A_idx = A._a_idx.get(i) if hasattr(A, "_a_idx") else None
                if A_idx is None:
                    A_idx = getAdjacencyNodeColumnIdx(A.neighbours[i], layer.num_nodes,
                                                      layer.num_relations).to(A_slices.device)
                    if hasattr(A, "_a_idx"):
                        A._a_idx[i] = A_idx
                X = layer(X, A_slices, A_idx)
            if self.p_dropout > 0.0:
                ones = dropout(torch.ones(X.shape[0]), p=self.p_dropout).to(X.device)
                X = X * ones.unsqueeze(1)
            if f_activation is not None:
                X = f_activation(X)
        return X

    def _forward_masked(self, X, A):
        """The same walk on a masked batch (data.batch.A_BatchMasked): every layer is a masked pass over the full
        graph's plan (functional.masked_layer) on compact arrays — X: one row per node of A.neighbours[-1], hidden
        activations one row per node of the sample they belong to, the result one row per batch node."""
        from .. import _lib
        from .. import functional as Fn
        for layer_idx, (key, layer) in enumerate(self.layers.items()):
            f_activation = self.activations[key] if key in self.activations else None
            sup = A.row[self.num_layers - (layer_idx + 1)]
            K = 0 if (layer.input_layer and layer.featureless) else int(X.shape[1])
            need_dX = K > 0 and bool(X.requires_grad) and torch.is_grad_enabled()
            if layer.engine != "fused" or not Fn.masked_layer_supported(sup, layer, K, need_dX):
                raise _lib.MrgcnError(
                    f"{key}: outside the masked mini-batch pass (fused engine, out <= 16, f32 operand, matrix-core "
                    "transform shapes: an input that wants its gradient has at most 64 columns); use "
                    "data.batch.A_BatchDevice / MiniBatch for it")
            fuse_relu = isinstance(f_activation, nn.ReLU) and self.p_dropout <= 0.0
            X = Fn.masked_layer(sup, layer, None if K == 0 else X, relu=fuse_relu)
            if self.p_dropout > 0.0:
                ones = dropout(torch.ones(X.shape[0]), p=self.p_dropout).to(X.device)
                X = X * ones.unsqueeze(1)
            if f_activation is not None and not fuse_relu:
                X = f_activation(X)
        return X.index_select(0, A.out_rank)

    def prepare_batch(self, A):
        """Builds the slice plans a (fresh) batch still lacks — two per layer at most — side by side.  Called by the
        forward; a prefetcher calls it ahead of time on its own stream (data/batch.py BatchPrefetcher)."""
        jobs = []
        for layer_idx, layer in enumerate(self.layers.values()):
            A_sl = A.row[self.num_layers - (layer_idx + 1)]
            if not (isinstance(A_sl, torch.Tensor) and A_sl.is_cuda):
                continue
            rb = [layer.operand_row_bytes()]
            if layer.input_layer:
                jobs.append((A_sl, layer.num_nodes, layer.num_relations, rb))
            sl = getattr(A_sl, "_mrgcn_slice", None)
            if sl is not None and not (layer.input_layer and layer.featureless):
                jobs.append((sl[1], int(A.neighbours[self.num_layers - (layer_idx + 1)].numel()), layer.num_relations, rb))
        if len(jobs) > 1:
            build_plans_parallel(jobs)
        return A

    def operand_row_bytes(self):
        """Row sizes of the layers' compact operands: the layout hint of the adjacency's graph plan."""
        return sorted({layer.operand_row_bytes() for layer in self.layers.values()})

    def _forward_full_batch(self, X, A):
        # the plan all layers share is built (on first use) for every layer's operand layout
        plan_of(A, self.num_nodes, self.layers["layer_0"].num_relations, operand_row_bytes=self.operand_row_bytes())
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
