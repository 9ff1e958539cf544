"""MLP literal encoder (numeric / boolean / temporal datatypes).  Structure and initialisation
follow mrgcn/models/perceptron.py:6-46: `num_layers` blocks of Linear -> Dropout(inplace) -> ReLU
whose widths step linearly from `input_dim` down to `output_dim`, every weight and bias drawn
from U(0, 1) in parameter order (so the same seed gives the reference's values; state-dict keys
`mlp.<3k>.weight|bias`).  Inside `MRGCN` the whole encoder — every Linear + ReLU, the gate multiply and
the scatter into the feature matrix — is ONE HIP kernel (`dense.mlp_gate_scatter`, csrc/encoders.hip) when
the literals live on the GPU, the widths are <= 16 and dropout is inactive.  Otherwise on the GPU (wider layers,
p_dropout > 0 in training, a stand-alone call) `forward` runs every Linear + ReLU on the matrix-core GEMM of
csrc/encoders.hip (`dense.linear`), with torch's dropout mask between layers — ReLU and inverted dropout commute
(the mask zeroes, the scale is positive), so Linear -> Dropout -> ReLU is computed as Linear+ReLU -> Dropout.
CPU tensors (host-side tooling, a GPU-less test box) take nn.Linear; inside `MRGCN` on a GPU box that never happens
(the encoders are placed next to the R-GCN, `MRGCN._compute_modality_embeddings` refuses a CPU encoder there)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def layer_widths(input_dim: int, output_dim: int, num_layers: int):
    """Output width of every block: output_dim + k * ((input_dim - output_dim) // num_layers) for
    k = num_layers - 1 ... 0."""
    stride = (input_dim - output_dim) // num_layers
    return [output_dim + k * stride for k in range(num_layers - 1, -1, -1)]


class MLP(nn.Module):
    def __init__(self, input_dim, output_dim, num_layers=3, p_dropout=0.0, bias=True):
        super().__init__()
        self.input_dim, self.output_dim, self.p_dropout = input_dim, output_dim, p_dropout
        dims = [input_dim] + layer_widths(input_dim, output_dim, num_layers)
        self.mlp = nn.Sequential(*[
            module for fan_in, fan_out in zip(dims, dims[1:])
            for module in (nn.Linear(fan_in, fan_out, bias), nn.Dropout(p=p_dropout, inplace=True), nn.ReLU())])
        self.init()

    @torch.no_grad()
    def init(self):
        for tensor in self.parameters():
            tensor.uniform_(0.0, 1.0)

    def forward(self, X):
        from .. import dense
        lin = self.linears()
        if not (X.is_cuda and all(l.weight.is_cuda for l in lin)):
            return self.mlp(X)                      # host-side tooling only (see the module docstring)
        X = X.float()
        for l in lin:
            X = dense.linear(X, l.weight, l.bias, relu=True)
            if self.training and self.p_dropout > 0:
                X = F.dropout(X, self.p_dropout, True)
        return X

    def linears(self):
        return [m for m in self.mlp if isinstance(m, nn.Linear)]

    def fused_ok(self, X) -> bool:
        """Can `dense.mlp_gate_scatter` take this call?"""
        from .. import dense
        lin = self.linears()
        return (dense.usable(X) and all(l.weight.is_cuda for l in lin) and not (self.training and self.p_dropout > 0)
                and dense.mlp_fused_supported([l.weight for l in lin]))
