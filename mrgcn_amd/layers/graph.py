"""R-GCN layer on MI355X behind the reference's `GraphConvolution` interface
(mrgcn/layers/graph.py:8-116: same constructor, parameter names/shapes/initialisation and
forward semantics), computing with the HIP kernels of libmrgcn_hip.so.

    Y = A . [ W_I  (+ X . W_F) ]  (+ b)        A: N x (R*N), column r*N + j

Two engines share the parameters:

  "fused"   (default) never builds the (R*N) x out operands of graph.py:70-75,:93-95.
            A dense operand M with one row per *touched* column of A is produced directly
            from the basis tables (M[c] = comp_I[r_c] . V_I[:, j_c, :] + X[j_c] . W_F[r_c])
            and multiplied with the compact view of A in one product.
  "literal" materialises W_I / FW_F exactly as the reference does and multiplies with the
            literal view; kept as the op-for-op counterpart (tests, roofline of the plain
            stacked-CSR SpMM).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as Fn
from ..plan import plan_of

_ENGINES = ("fused", "literal")
DEFAULT_ENGINE = "fused"


class GraphConvolution(nn.Module):
    def __init__(self, indim, outdim, num_relations, num_nodes, num_bases=-1, bias=False,
                 input_layer=False, featureless=False, shared_bases_weights=False):
        super().__init__()
        self.indim, self.outdim = indim, outdim
        self.num_relations, self.num_nodes = num_relations, num_nodes
        self.num_bases = num_bases
        self.input_layer, self.featureless, self.bias = input_layer, featureless, bias
        self.engine = DEFAULT_ENGINE
        # storage type of the fused engine's compact operand: "f32" (parity default) or "bf16"
        # (SURVEY §8d's additional run: bf16 dense operand, fp32 accumulation, tolerance 2e-2)
        self.operand_dtype = "f32"

        use_bases = num_bases > 0
        S = num_bases if use_bases else num_relations  # graph.py:33-36
        wants_F = not featureless

        # registration order fixes the RNG stream of reset_parameters (graph.py:38-57):
        # weight_I_comp, weight_F_comp, weight_I, weight_F, b
        shapes = [
            ("weight_I_comp", (num_relations, num_bases) if use_bases and input_layer else None),
            ("weight_F_comp", (num_relations, num_bases) if use_bases and wants_F else None),
            ("weight_I", (S * num_nodes, outdim) if input_layer else None),
            ("weight_F", (S, indim, outdim) if wants_F else None),
            ("b", (outdim,) if bias else None),
        ]
        for name, shape in shapes:
            if shape is None:
                setattr(self, name, None)
            elif name == "weight_F_comp" and shared_bases_weights:
                # graph.py:42-44: alias of weight_I_comp (None on non-input layers)
                self.weight_F_comp = self.weight_I_comp
            else:
                setattr(self, name, nn.Parameter(torch.empty(shape)))
        self.reset_parameters()

    def reset_parameters(self):
        """Glorot-uniform on every tensor but the bias, zeros on the bias (graph.py:104-116)."""
        for name, param in self.named_parameters():
            if name == "b":
                nn.init.zeros_(param)
            else:
                nn.init.xavier_uniform_(param)

    # ------------------------------------------------------------------------------
    def forward(self, X, A, A_idx=None):
        if A_idx is not None:
            return self._forward_mini_batch(X, A, A_idx)
        plan = plan_of(A, self.num_nodes, self.num_relations)
        if self.engine == "literal":
            return self._forward_literal(X, plan)
        return self._forward_fused(X, plan)

    # -- op-for-op counterpart of graph.py:62-102 ---------------------------------------
    def _forward_literal(self, X, plan):
        R, N, B, out = self.num_relations, self.num_nodes, self.num_bases, self.outdim
        Y = None
        if self.input_layer:
            W_I = self.weight_I
            if B > 0:
                W_I = (self.weight_I_comp @ W_I.view(B, N * out)).view(R * N, out)
            last = self.featureless
            Y = Fn.spmm_literal(plan, W_I, bias=self.b if (last and self.bias) else None)
            if last:
                return Y
        W_F = self.weight_F
        if B > 0:
            W_F = (self.weight_F_comp @ W_F.view(B, -1)).view(R, self.indim, out)
        FW = torch.matmul(X.unsqueeze(0), W_F).reshape(R * X.shape[0], out)
        AFW = Fn.spmm_literal(plan, FW, bias=self.b if self.bias else None)
        return AFW if Y is None else Y + AFW

    # -- mini-batch mode (graph.py:62-102 with A_idx) --------------------------------------
    def _forward_mini_batch(self, X, A, A_idx):
        """`A`: the row slice of the sample nodes (|sample| x R*N, global columns), `X`: features /
        embeddings of the n_b neighbour nodes, `A_idx`: their columns for every relation.  The
        input term keeps the global column space and the stored values; the feature term runs on
        `sliceSparseCOO(A, A_idx)` (|sample| x R*n_b, all-ones values) — two graph plans, both
        cached on the slice tensor."""
        from ..data.batch import sliceSparseCOO
        R, B, out = self.num_relations, self.num_bases, self.outdim
        n_b = X.shape[0]
        cached = getattr(A, "_mrgcn_slice", None)
        if cached is None or cached[0] is not A_idx:
            cached = (A_idx, sliceSparseCOO(A, A_idx))
            A._mrgcn_slice = cached
        plan_F = plan_of(cached[1], n_b, R)
        Y = None
        if self.input_layer:
            plan_I = plan_of(A, self.num_nodes, R)
            if self.engine == "literal":
                W_I = self.weight_I
                if B > 0:
                    W_I = (self.weight_I_comp @ W_I.view(B, self.num_nodes * out)).view(R * self.num_nodes, out)
                Y = Fn.spmm_literal(plan_I, W_I)
            else:
                Y = Fn.rgcn_layer(plan_I, self, None, feature_term=False, use_bias=False)
        if self.engine == "literal":
            W_F = self.weight_F
            if B > 0:
                W_F = (self.weight_F_comp @ W_F.view(B, -1)).view(R, self.indim, out)
            FW = torch.matmul(X.unsqueeze(0), W_F).reshape(R * n_b, out)
            YF = Fn.spmm_literal(plan_F, FW, bias=self.b if self.bias else None)
        else:
            YF = Fn.rgcn_layer(plan_F, self, X, input_term=False)
        return YF if Y is None else Y + YF

    # -- fused engine ----------------------------------------------------------------------
    def _forward_fused(self, X, plan, relu=False):
        return Fn.rgcn_layer(plan, self, X, relu=relu)
