"""MLP literal encoder with the reference's structure and initialisation
(mrgcn/models/perceptron.py:6-46): `num_layers` blocks of Linear -> Dropout(inplace) -> ReLU
whose widths step linearly from input_dim down to output_dim; every weight and bias is
drawn from U(0, 1).  Dense and tiny: runs on the library GEMM (rocBLAS via nn.Linear)."""
import torch.nn as nn


class MLP(nn.Module):
    def __init__(self, input_dim, output_dim, num_layers=3, p_dropout=0.0, bias=True):
        super().__init__()
        self.input_dim, self.output_dim, self.p_dropout = input_dim, output_dim, p_dropout
        step = (input_dim - output_dim) // num_layers
        widths = [output_dim + k * step for k in range(num_layers - 1, -1, -1)]
        blocks, fan_in = [], input_dim
        for width in widths:
            blocks += [nn.Linear(fan_in, width, bias), nn.Dropout(p=p_dropout, inplace=True), nn.ReLU()]
            fan_in = width
        self.mlp = nn.Sequential(*blocks)
        self.init()

    def forward(self, X):
        return self.mlp(X)

    def init(self):
        for param in self.parameters():
            nn.init.uniform_(param)
