"""Character-level temporal CNN encoder for `ogc.wktLiteral` values (reference:
mrgcn/models/temporal_cnn.py:6-160): Conv1d -> BatchNorm1d -> ReLU blocks with max-pools in
between, an adaptive max-pool, a closing valid convolution that leaves one position, then
Linear -> ReLU -> Dropout -> Linear.  Built from a per-size table so that the layer order — hence
the state-dict keys (`conv.<i>.*`, `fc.<i>.*`, both also under `module_dict.`) and the order in
which the default initialisers consume the RNG — equals the reference's.  On the GPU every block runs on the
HIP kernels: the Conv1d as an implicit-im2col product on the matrix cores, BatchNorm1d + ReLU (+ the pooling layer
that follows) as one pass (dense.conv1d / dense.bn_relu_pool / dense.linear; csrc/encoders.hip, csrc/tcnn.hip);
the torch.nn modules below hold the parameters / buffers and serve CPU-side tooling."""
import torch.nn as nn

# (kind, *args): "c" = Conv1d(out_channels, kernel, padding) + BatchNorm1d + ReLU, "p" = MaxPool1d(k, stride k),
# "a" = AdaptiveMaxPool1d(n)
_SPECS = {
    "S": (20, [("c", 64, 3, 1), ("c", 64, 3, 1), ("p", 2),
               ("c", 128, 3, 1), ("c", 128, 3, 1), ("p", 2),
               ("c", 256, 3, 1), ("c", 256, 3, 1), ("a", 2),
               ("c", 512, 2, 0)]),
    "M": (300, [("c", 64, 7, 3), ("c", 64, 7, 3), ("p", 3),
                ("c", 128, 3, 1), ("c", 128, 3, 1), ("p", 3),
                ("c", 256, 3, 1), ("c", 256, 3, 1), ("a", 3),
                ("c", 512, 3, 1), ("c", 512, 3, 1), ("c", 1024, 3, 0)]),
    "L": (300, [("c", 64, 7, 3), ("c", 64, 7, 3), ("p", 3),
                ("c", 128, 7, 3), ("c", 128, 7, 3), ("p", 3),
                ("c", 256, 3, 1), ("c", 256, 3, 1), ("p", 3),
                ("c", 512, 3, 1), ("c", 512, 3, 1), ("a", 3),
                ("c", 1024, 3, 1), ("c", 1024, 3, 1), ("c", 2048, 3, 0)]),
}


class TCNN(nn.Module):
    LENGTH_S, LENGTH_M, LENGTH_L = 20, 100, 300

    def __init__(self, features_in, features_out, p_dropout=0.0, size="M"):
        super().__init__()
        self.module_dict = nn.ModuleDict()
        self.minimal_length, spec = _SPECS[size]
        layers, channels = [], features_in
        for kind, *args in spec:
            if kind == "c":
                out_ch, k, pad = args
                layers += [nn.Conv1d(channels, out_ch, kernel_size=k, padding=pad), nn.BatchNorm1d(out_ch), nn.ReLU()]
                channels = out_ch
            elif kind == "p":
                layers.append(nn.MaxPool1d(kernel_size=args[0], stride=args[0]))
            else:
                layers.append(nn.AdaptiveMaxPool1d(args[0]))
        self.conv = nn.Sequential(*layers)
        self.module_dict["conv"] = self.conv
        self.fc = nn.Sequential(nn.Linear(channels, channels), nn.ReLU(), nn.Dropout(p=p_dropout),
                                nn.Linear(channels, features_out))
        self.module_dict["fc"] = self.fc

    def forward(self, X):
        from .. import dense
        if not dense.usable(X, self.fc[0].weight):
            X = self.conv(X)
            return self.fc(X.view(X.size(0), -1))
        # GPU: Conv1d = implicit-im2col product on the matrix cores; BatchNorm1d + ReLU + the pooling layer behind
        # them (if any) = one elementwise pass
        mods = list(self.conv)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv1d):
                X = dense.conv1d(X, m.weight, m.bias, m.padding[0])
                i += 1
            elif isinstance(m, nn.BatchNorm1d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                kind, arg, used = dense.POOL_NONE, 0, 2
                nxt = mods[i + 2] if i + 2 < len(mods) else None
                if isinstance(nxt, nn.MaxPool1d) and nxt.stride == nxt.kernel_size and nxt.padding == 0 \
                        and nxt.dilation == 1 and not nxt.ceil_mode:
                    kind, arg, used = dense.POOL_MAX, int(nxt.kernel_size), 3
                elif isinstance(nxt, nn.AdaptiveMaxPool1d):
                    kind, arg, used = dense.POOL_ADAPTIVE, int(nxt.output_size), 3
                X = dense.bn_relu_pool(X, m, kind, arg)
                i += used
            else:
                from .. import _lib
                raise _lib.MrgcnError(f"TCNN: no HIP kernel for {type(m).__name__} at position {i} of the "
                                      "convolution stack (expected Conv1d, BatchNorm1d + ReLU [+ pooling])")
        X = dense.linear(X.view(X.size(0), -1), self.fc[0].weight, self.fc[0].bias, relu=True)
        return dense.linear(self.fc[2](X), self.fc[3].weight, self.fc[3].bias)


def out_dim(seq_length, kernel_size, padding=0, stride=1, dilation=1):
    return (seq_length + 2 * padding - dilation * (kernel_size - 1) - 1) // stride + 1
