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

Internal layout of `weight_I` with bases (num_bases > 0): the parameter is kept NODE-MAJOR,
`(N, B, out)` — the B basis rows of a node are one contiguous block — instead of the reference's
`(B*N, out)` (graph.py:50-51, rows b*N + j).  It is the same numbers transposed: `state_dict()`
hands out and `load_state_dict()` takes the reference's shape (checkpoints are interchangeable,
`weight_I_reference()` gives the reference view), `reset_parameters()` draws the Glorot values in
the reference's shape and order (same seed, same initial values).  In this layout the forward, the
backward and Adam touch a node's weights as one 4*B*out-byte run, and a node without gradient is
skipped as a whole.  Without bases `weight_I` is `(R*N, out)` exactly as in the reference.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as Fn
from ..plan import plan_of

_ENGINES = ("fused", "literal")
DEFAULT_ENGINE = "fused"


# Library GEMMs of the literal engine are issued in pieces whose outputs stay below this many elements: with
# (R*N) x out results beyond 4 GiB (AM/4 and up) the ROCm BLAS path was measured to return a handful of wrong
# rows from the second call on (the fused engine never forms these operands).
_MAX_GEMM_OUT = 1 << 28


def _basis_contract(comp, W2d):
    """einsum('rb,bx->rx') (graph.py:69-72 / :83-85 after the views), output columns in pieces."""
    R, X = comp.shape[0], W2d.shape[1]
    step = max(_MAX_GEMM_OUT // max(R, 1), 1)
    if X <= step:
        return comp @ W2d
    return torch.cat([comp @ W2d[:, x0:x0 + step] for x0 in range(0, X, step)], dim=1)


def _relation_transform(X, W_F):
    """einsum('ij,bjk->bik') (graph.py:93-94): [R, N, out], relations in pieces."""
    R, n, out = W_F.shape[0], X.shape[0], W_F.shape[2]
    step = max(_MAX_GEMM_OUT // max(n * out, 1), 1)
    if R <= step:
        return torch.matmul(X.unsqueeze(0), W_F)
    return torch.cat([torch.matmul(X.unsqueeze(0), W_F[r0:r0 + step]) for r0 in range(0, R, step)], dim=0)


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
        self.weight_I_node_major = bool(use_bases and input_layer)
        for name, shape in shapes:
            if shape is None:
                setattr(self, name, None)
            elif name == "weight_F_comp" and shared_bases_weights:
                # graph.py:42-44: alias of weight_I_comp (None on non-input layers)
                self.weight_F_comp = self.weight_I_comp
            elif name == "weight_I" and self.weight_I_node_major:
                self.weight_I = nn.Parameter(torch.empty((num_nodes, num_bases, outdim)))
            else:
                setattr(self, name, nn.Parameter(torch.empty(shape)))
        self._register_state_dict_hook(_weight_I_to_reference)
        self._register_load_state_dict_pre_hook(_weight_I_from_reference, with_module=True)
        self._tag_parameters()
        self.reset_parameters()

    def _tag_parameters(self):
        """Marks the node-major `weight_I` Parameter: optimizers (mrgcn_amd.train / mrgcn_amd.optim) hand its state
        out in the reference's layout.  The mark is a Python attribute of the Parameter object, which copies of the
        module do not inherit — `copy.deepcopy(model)`, unpickling and parameter-replacing conversions re-create the
        Parameter — so it is re-applied wherever a module can come by a new one (`__setstate__`, `_apply`); the
        module's own `weight_I_node_major` is the source of truth."""
        w = self._parameters.get("weight_I")
        if w is not None and self.weight_I_node_major:
            w._mrgcn_node_major = True

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tag_parameters()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._tag_parameters()
        return out

    def reset_parameters(self):
        """Glorot-uniform on every tensor but the bias, zeros on the bias (graph.py:104-116); the values
        are drawn in the reference's shapes and order, so the same seed gives the same parameters."""
        for name, param in self.named_parameters():
            if name == "b":
                nn.init.zeros_(param)
            elif name == "weight_I" and self.weight_I_node_major:
                N, B, out = param.shape
                ref = torch.empty((B * N, out), dtype=param.dtype, device=param.device)
                nn.init.xavier_uniform_(ref)
                with torch.no_grad():
                    param.copy_(ref.view(B, N, out).permute(1, 0, 2))
            else:
                nn.init.xavier_uniform_(param)

    def weight_I_reference(self):
        """`weight_I` in the reference's shape `(S*N, out)` (graph.py:50-51); differentiable."""
        W = self.weight_I
        if W is not None and self.weight_I_node_major:
            N, B, out = W.shape
            W = W.permute(1, 0, 2).reshape(B * N, out)
        return W

    # ------------------------------------------------------------------------------
    def forward(self, X, A, A_idx=None):
        if A_idx is not None:
            return self._forward_mini_batch(X, A, A_idx)
        plan = plan_of(A, self.num_nodes, self.num_relations, operand_row_bytes=[self.operand_row_bytes()])
        if self.engine == "literal":
            return self._forward_literal(X, plan)
        return self._forward_fused(X, plan)

    # -- op-for-op counterpart of graph.py:62-102 ---------------------------------------
    def _forward_literal(self, X, plan):
        R, N, B, out = self.num_relations, self.num_nodes, self.num_bases, self.outdim
        Y = None
        if self.input_layer:
            W_I = self.weight_I_reference()
            if B > 0:
                W_I = _basis_contract(self.weight_I_comp, W_I.view(B, N * out)).view(R * N, out)
            last = self.featureless
            Y = Fn.spmm_literal(plan, W_I, bias=self.b if (last and self.bias) else None)
            if last:
                return Y
        W_F = self.weight_F
        if B > 0:
            W_F = (self.weight_F_comp @ W_F.view(B, -1)).view(R, self.indim, out)
        FW = _relation_transform(X, W_F).reshape(R * X.shape[0], out)
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
        plan_F = plan_of(cached[1], n_b, R, operand_row_bytes=[self.operand_row_bytes()])
        Y = None
        if self.input_layer:
            plan_I = plan_of(A, self.num_nodes, R, operand_row_bytes=[self.operand_row_bytes()])
            if self.engine == "literal":
                W_I = self.weight_I_reference()
                if B > 0:
                    W_I = _basis_contract(self.weight_I_comp, W_I.view(B, self.num_nodes * out)
                                          ).view(R * self.num_nodes, out)
                Y = Fn.spmm_literal(plan_I, W_I, bias=self.b if (self.featureless and self.bias) else None)
            else:
                Y = Fn.rgcn_layer(plan_I, self, None, feature_term=False, use_bias=self.featureless)
            if self.featureless:  # graph.py:77-81: the input term (+ bias) is the whole layer
                return Y
        if self.engine == "literal":
            W_F = self.weight_F
            if B > 0:
                W_F = (self.weight_F_comp @ W_F.view(B, -1)).view(R, self.indim, out)
            FW = _relation_transform(X, W_F).reshape(R * n_b, out)
            YF = Fn.spmm_literal(plan_F, FW, bias=self.b if self.bias else None)
        else:
            YF = Fn.rgcn_layer(plan_F, self, X, input_term=False)
        return YF if Y is None else Y + YF

    def operand_row_bytes(self) -> int:
        """Row size of this layer's compact operand: the layout hint its graph plan takes (plan.plan_of)."""
        return Fn.operand_row_bytes(self.outdim, getattr(self, "operand_dtype", "f32") == "bf16")

    # -- fused engine ----------------------------------------------------------------------
    def _forward_fused(self, X, plan, relu=False):
        return Fn.rgcn_layer(plan, self, X, relu=relu)


def _weight_I_to_reference(module, state_dict, prefix, local_metadata):
    """state_dict hook: the node-major `(N, B, out)` parameter leaves as the reference's `(B*N, out)`."""
    key = prefix + "weight_I"
    if getattr(module, "weight_I_node_major", False) and key in state_dict:
        W = state_dict[key]
        N, B, out = W.shape
        state_dict[key] = W.permute(1, 0, 2).reshape(B * N, out)
    return state_dict


def _weight_I_from_reference(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                             error_msgs):
    """load_state_dict pre-hook: a reference-shaped `weight_I` is transposed into the node-major layout."""
    key = prefix + "weight_I"
    if getattr(module, "weight_I_node_major", False) and key in state_dict:
        W = state_dict[key]
        N, B, out = module.weight_I.shape
        if W.dim() == 2 and tuple(W.shape) == (B * N, out):
            state_dict[key] = W.view(B, N, out).permute(1, 0, 2).contiguous()
