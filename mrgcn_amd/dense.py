"""autograd wiring of the encoder kernels (csrc/encoders.hip): the dense step in front of the graph path
(SURVEY §8f next-2).  Like the graph ops these call the C ABI only — no PyTorch or CPU fallback inside;
the modules in `models/` decide per call whether their input qualifies (GPU tensors, supported widths)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# Arithmetic of the encoders' products: "f32" (v_mfma_f32_16x16x4_f32, exact fp32 — the parity default) or "bf16" (the
# bf16 pipeline of BASELINE config 3: operands rounded to bf16 as tiles are staged, v_mfma_f32_16x16x32_bf16, fp32
# accumulation; activations, parameters, BatchNorm statistics and every gradient stay fp32 in memory).  Process-wide,
# like torch's matmul precision switches; MRGCN.set_compute_dtype sets it.
_MATMUL_DTYPE = "f32"


def set_matmul_dtype(dtype: str) -> str:
    """Returns the previous setting."""
    global _MATMUL_DTYPE
    assert dtype in ("f32", "bf16")
    prev, _MATMUL_DTYPE = _MATMUL_DTYPE, dtype
    return prev


def matmul_dtype() -> str:
    return _MATMUL_DTYPE


def _gemm(amode, bmode, cmode, M, N, K, A, lda, B, ldb, Cout, ldc, bias=None, relu=False, mask=None, alpha=1.0,
          geom=None, mm=None):
    """`mm`: the arithmetic ("f32" / "bf16"; None = the process-wide setting) — a backward passes what its forward ran
    with."""
    g = (C.c_int32 * 6)(*geom) if geom is not None else None
    lib = L.load()
    fn, name = ((lib.mrgcn_gemm_bf16mm_f32, "mrgcn_gemm_bf16mm_f32") if (mm or _MATMUL_DTYPE) == "bf16"
                else (lib.mrgcn_gemm_f32, "mrgcn_gemm_f32"))
    with torch.cuda.device(Cout.device):
        L.check(fn(amode, bmode, cmode, M, N, K, A.data_ptr(), lda, B.data_ptr(), ldb, Cout.data_ptr(), ldc,
                   bias.data_ptr() if bias is not None else 0, 1 if relu else 0, mask.data_ptr() if mask is not None else 0,
                   float(alpha), g, _stream(Cout.device)), name)
    return Cout


def _colsum(X2d):
    M, N = X2d.shape
    out = torch.empty(N, dtype=torch.float32, device=X2d.device)
    with torch.cuda.device(X2d.device):
        L.check(L.load().mrgcn_colsum_f32(X2d.data_ptr(), X2d.stride(0), M, N, out.data_ptr(), _stream(X2d.device)),
                "mrgcn_colsum_f32")
    return out


def usable(*tensors) -> bool:
    """The HIP encoder path takes float32 GPU tensors."""
    return all(t is not None and t.is_cuda and t.dtype == torch.float32 for t in tensors)


# The tiled product addresses its operands through 32-bit byte offsets: an operand of a single launch holds at most
# 2^29 bytes (csrc/encoders.hip kMmMaxBytes; larger ones would fall to the element-loader kernel, ~40x slower).  A
# convolution over a large batch (a linear layer over many rows) is therefore cut along the batch: samples are
# independent in the forward and dX products, and dW is the sum of the per-piece products.
_MM_MAX_BYTES = 1 << 29


def _batch_pieces(Bn: int, *elems_per_sample: int):
    """[(b0, b1), ...]: batch ranges whose operands (elements per sample given) each stay within _MM_MAX_BYTES."""
    per = max(1, max(elems_per_sample)) * 4
    step = max(1, _MM_MAX_BYTES // per - 1)   # (the launcher's size estimate counts one sample more)
    return [(b0, min(Bn, b0 + step)) for b0 in range(0, Bn, step)]


# ---------------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = relu?(x W^T + b) on the matrix cores (nn.Linear semantics: W is [out, in])."""

    @staticmethod
    def forward(ctx, x, W, b, relu: bool):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        Wc = W.contiguous()
        n, K = x2.shape
        N = Wc.shape[0]
        y = torch.empty((n, N), dtype=torch.float32, device=x.device)
        bc = b.contiguous() if b is not None else None
        for r0, r1 in _batch_pieces(n, K, N):
            _gemm(0, 1, 0, r1 - r0, N, K, x2[r0:r1], K, Wc, K, y[r0:r1], N, bias=bc, relu=relu)
        ctx.relu, ctx.shape, ctx.has_b, ctx.mm = relu, x.shape, b is not None, _MATMUL_DTYPE
        ctx.save_for_backward(x2, Wc, y if relu else None)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W, y = ctx.saved_tensors
        n, K = x2.shape
        N = W.shape[0]
        dy2 = dy.reshape(n, N).contiguous()
        if ctx.relu:  # gradient at the pre-activation: dy * (y > 0)
            from .functional import relu_bwd
            dy2 = relu_bwd(dy2, y)
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((n, K), dtype=torch.float32, device=dy.device)
            for r0, r1 in _batch_pieces(n, K, N):
                _gemm(0, 0, 0, r1 - r0, K, N, dy2[r0:r1], N, W, K, dx[r0:r1], K, mm=ctx.mm)   # dx = dy . W
            dx = dx.view(ctx.shape)
        if ctx.needs_input_grad[1]:
            for r0, r1 in _batch_pieces(n, K, N):
                part = torch.empty((N, K), dtype=torch.float32, device=dy.device)
                _gemm(1, 0, 0, N, K, r1 - r0, dy2[r0:r1], N, x2[r0:r1], K, part, K, mm=ctx.mm)   # dW = dy^T . x
                dW = part if dW is None else dW.add_(part)
            if dW is None:
                dW = torch.zeros((N, K), dtype=torch.float32, device=dy.device)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = _colsum(dy2)
        return dx, dW, db, None


def linear(x, W, b=None, relu: bool = False):
    return _Linear.apply(x, W, b, relu)


def _chan_sum_of(dy, C):
    """The per-channel sums the producer of this gradient tensor left on it (the batch-norm block's backward writes
    them beside dx), or None: valid only for the very tensor they were computed for, unmodified since."""
    m = getattr(dy, "_mrgcn_chan_sum", None)
    if m is not None and m[0] == dy._version and m[1].numel() == C and m[1].device == dy.device:
        return m[1]
    return None


class _Conv1d(torch.autograd.Function):
    """nn.Conv1d (stride 1, dilation 1, zero padding) as an implicit-im2col product on the matrix cores:
    y[b, co, t] = bias[co] + sum_{ci, kw} W[co, ci, kw] x[b, ci, t + kw - pad]."""

    @staticmethod
    def forward(ctx, x, W, b, pad: int):
        x = x.contiguous()
        Wc = W.contiguous()
        Bn, Cin, Tin = x.shape
        Cout, _, KW = Wc.shape
        Tout = Tin + 2 * pad - KW + 1
        y = torch.empty((Bn, Cout, Tout), dtype=torch.float32, device=x.device)
        geom = (Cin, Tin, KW, pad, Tout, Cout)
        bc = b.contiguous() if b is not None else None
        for b0, b1 in _batch_pieces(Bn, Cin * Tin, Cout * Tout):
            _gemm(2, 1, 2, (b1 - b0) * Tout, Cout, Cin * KW, x[b0:b1], 0, Wc.view(Cout, Cin * KW), Cin * KW, y[b0:b1], 0,
                  bias=bc, geom=geom)
        ctx.geom, ctx.has_b, ctx.mm = geom, b is not None, _MATMUL_DTYPE
        ctx.save_for_backward(x, Wc)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        Cin, Tin, KW, pad, Tout, Cout = ctx.geom
        Bn = x.shape[0]
        dy = dy.contiguous()
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            # dx[b, ci, t] = sum_{co, kw'} dy[b, co, t + kw' - (KW-1-pad)] W[co, ci, KW-1-kw']: a convolution of dy
            Wf = W.flip(2).permute(0, 2, 1).reshape(Cout * KW, Cin).contiguous()   # [(co, kw')][ci]
            dx = torch.empty_like(x)
            for b0, b1 in _batch_pieces(Bn, Cin * Tin, Cout * Tout):
                _gemm(2, 0, 2, (b1 - b0) * Tin, Cin, Cout * KW, dy[b0:b1], 0, Wf, Cin, dx[b0:b1], 0,
                      geom=(Cout, Tout, KW, KW - 1 - pad, Tin, Cin), mm=ctx.mm)
        if ctx.needs_input_grad[1]:
            # dW^T[(ci, kw)][co] = sum_{b, t} x[b, ci, t + kw - pad] dy[b, co, t]
            dWt = None
            for b0, b1 in _batch_pieces(Bn, Cin * Tin, Cout * Tout):
                part = torch.empty((Cin * KW, Cout), dtype=torch.float32, device=dy.device)
                _gemm(3, 2, 0, Cin * KW, Cout, (b1 - b0) * Tout, x[b0:b1], 0, dy[b0:b1], 0, part, Cout, geom=ctx.geom,
                      mm=ctx.mm)
                dWt = part if dWt is None else dWt.add_(part)
            if dWt is None:
                dWt = torch.zeros((Cin * KW, Cout), dtype=torch.float32, device=dy.device)
            dW = dWt.t().reshape(Cout, Cin, KW).contiguous()
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = _chan_sum_of(dy, Cout)
        if ctx.has_b and ctx.needs_input_grad[2] and db is None:
            db = torch.empty(Cout, dtype=torch.float32, device=dy.device)
            with torch.cuda.device(dy.device):
                L.check(L.load().mrgcn_channel_sum_f32(dy.data_ptr(), Bn, Cout, Tout, db.data_ptr(), _stream(dy.device)),
                        "mrgcn_channel_sum_f32")
        return dx, dW, db, None


def conv1d(x, W, b=None, padding: int = 0):
    return _Conv1d.apply(x, W, b, int(padding))


# ---------------------------------------------------------------------------------------------------------
POOL_NONE, POOL_MAX, POOL_ADAPTIVE = 0, 1, 2


class _BnReluPool(torch.autograd.Function):
    """BatchNorm1d -> ReLU [-> MaxPool1d(k, k) | AdaptiveMaxPool1d(n)] of a [B, C, T] tensor in one pass
    (csrc/tcnn.hip).  `mean` / `var` are written by the forward in training mode (batch statistics, biased
    variance) and read in eval mode (running statistics)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps: float, training: bool, kind: int, arg: int):
        x = x.contiguous()
        Bn, Cn, T = x.shape
        lib = L.load()
        Tout = int(lib.mrgcn_pool_out_len(kind, arg, T))
        if Tout <= 0:
            raise L.MrgcnError("bn_relu_pool: the pooling window is longer than the sequence")
        y = torch.empty((Bn, Cn, Tout), dtype=torch.float32, device=x.device)
        am = torch.empty((Bn, Cn, Tout), dtype=torch.int32, device=x.device) if kind != POOL_NONE else None
        ws = torch.empty(int(lib.mrgcn_bn_workspace_bytes(Cn)), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            L.check(lib.mrgcn_bn_relu_pool_fwd_f32(
                x.data_ptr(), Bn, Cn, T, gamma.data_ptr() if gamma is not None else 0,
                beta.data_ptr() if beta is not None else 0, float(eps), int(training), mean.data_ptr(),
                var.data_ptr(), kind, arg, y.data_ptr(), am.data_ptr() if am is not None else 0, ws.data_ptr(),
                _stream(x.device)), "mrgcn_bn_relu_pool_fwd_f32")
        ctx.meta = (float(eps), bool(training), kind, arg)
        ctx.save_for_backward(x, y, am, gamma, mean, var)
        ctx.mark_non_differentiable(mean, var)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, am, gamma, mean, var = ctx.saved_tensors
        eps, training, kind, arg = ctx.meta
        Bn, Cn, T = x.shape
        dy = dy.contiguous()
        dz = torch.empty_like(x) if kind == POOL_ADAPTIVE else None   # scatter target of overlapping windows only
        dx = torch.empty_like(x)
        dgamma = torch.empty(Cn, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(Cn, dtype=torch.float32, device=x.device)
        lib = L.load()
        ws = torch.empty(int(lib.mrgcn_bn_workspace_bytes(Cn)), dtype=torch.uint8, device=x.device)
        # per-channel sums of dx, taken in the pass that writes it: what the convolution in front of this block needs as
        # its bias gradient (handed over on the gradient tensor, _chan_sum_of)
        cs = torch.empty(Cn, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.check(lib.mrgcn_bn_relu_pool_bwd_sum_f32(
                x.data_ptr(), y.data_ptr(), dy.data_ptr(), am.data_ptr() if am is not None else 0, Bn, Cn, T,
                gamma.data_ptr() if gamma is not None else 0, mean.data_ptr(), var.data_ptr(), eps, int(training),
                kind, arg, dz.data_ptr() if dz is not None else 0, dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                cs.data_ptr(), ws.data_ptr(), _stream(x.device)), "mrgcn_bn_relu_pool_bwd_sum_f32")
        dx._mrgcn_chan_sum = (dx._version, cs)
        return dx, (dgamma if gamma is not None else None), (dbeta if gamma is not None else None), None, None, \
            None, None, None, None


def bn_relu_pool(x, bn, pool_kind: int = POOL_NONE, pool_arg: int = 0):
    """`bn`: an nn.BatchNorm1d; its running statistics are updated as nn.BatchNorm1d.forward does
    (momentum, unbiased variance, num_batches_tracked) when it is in training mode."""
    training = bn.training or bn.running_mean is None
    if training:
        mean = torch.empty(bn.num_features, dtype=torch.float32, device=x.device)
        var = torch.empty_like(mean)
    else:
        mean, var = bn.running_mean, bn.running_var
    y = _BnReluPool.apply(x, bn.weight, bn.bias, mean, var, bn.eps, training, int(pool_kind), int(pool_arg))
    if bn.training and bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            bn.num_batches_tracked += 1
            n = x.shape[0] * x.shape[2]
            if (bn.momentum is not None and bn.running_mean.is_cuda and bn.running_mean.dtype == torch.float32
                    and bn.running_mean.is_contiguous() and bn.running_var.is_contiguous()):
                with torch.cuda.device(x.device):
                    L.check(L.load().mrgcn_bn_running_stats_f32(
                        mean.data_ptr(), var.data_ptr(), bn.num_features, n, float(bn.momentum),
                        bn.running_mean.data_ptr(), bn.running_var.data_ptr(), _stream(x.device)),
                        "mrgcn_bn_running_stats_f32")
            else:  # cumulative average (momentum None): the factor depends on the counter
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                bn.running_var.mul_(1 - mom).add_(var * (n / max(n - 1, 1)), alpha=mom)
    return y


# ---------------------------------------------------------------------------------------------------------
def mlp_dims(weights):
    return [int(weights[0].shape[1])] + [int(w.shape[0]) for w in weights]


def mlp_fused_supported(weights) -> bool:
    dims = mlp_dims(weights)
    arr = (C.c_int32 * len(dims))(*dims)
    return bool(L.load().mrgcn_mlp_fused_supported(len(weights), arr))


def _ptr_array(tensors, n):
    return (C.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in tensors])


class _MlpGateScatter(torch.autograd.Function):
    """XF[rows, off : off + d_out] = gate_weights[i_gate] * MLP(enc) in one kernel (mrgcn.py:285-303 with
    perceptron.py:6-46 inside).  XF is written in place and handed on."""

    @staticmethod
    def forward(ctx, XF, enc, rows, gate_weights, i_gate: int, offset: int, n_layers: int, fresh: bool, *params):
        ctx.fresh = bool(fresh)
        weights = [p.contiguous() for p in params[:n_layers]]
        biases = [(p.contiguous() if p is not None else None) for p in params[n_layers:]]
        dims = mlp_dims(weights)
        enc = enc.contiguous()
        lib = L.load()
        dims_arr = (C.c_int32 * len(dims))(*dims)
        gate = gate_weights[i_gate:i_gate + 1].contiguous()
        with torch.cuda.device(XF.device):
            L.check(lib.mrgcn_mlp_gate_scatter_fwd_f32(
                n_layers, dims_arr, _ptr_array(weights, n_layers), _ptr_array(biases, n_layers), enc.data_ptr(),
                enc.stride(0), enc.shape[0], gate.data_ptr(), rows.data_ptr() if rows is not None else 0,
                XF.data_ptr(), XF.stride(0), offset, _stream(XF.device)), "mrgcn_mlp_gate_scatter_fwd_f32")
        ctx.meta = (i_gate, offset, n_layers, dims, [b is not None for b in biases])
        ctx.save_for_backward(enc, rows, gate_weights, *weights, *[b for b in biases if b is not None])
        ctx.mark_dirty(XF)
        return XF

    @staticmethod
    def backward(ctx, dXF):
        i_gate, offset, n_layers, dims, has_b = ctx.meta
        enc, rows, gate_weights = ctx.saved_tensors[:3]
        weights = list(ctx.saved_tensors[3:3 + n_layers])
        bs = list(ctx.saved_tensors[3 + n_layers:])
        biases = [bs.pop(0) if h else None for h in has_b]
        dXF = dXF.contiguous()
        dW = [torch.zeros_like(w) for w in weights]
        db = [torch.zeros_like(b) if b is not None else None for b in biases]
        dgate = torch.zeros_like(gate_weights)
        gate = gate_weights[i_gate:i_gate + 1].contiguous()
        dims_arr = (C.c_int32 * len(dims))(*dims)
        dg_view = dgate[i_gate:i_gate + 1]
        with torch.cuda.device(dXF.device):
            L.check(L.load().mrgcn_mlp_gate_scatter_bwd_f32(
                n_layers, dims_arr, _ptr_array(weights, n_layers), _ptr_array(biases, n_layers), enc.data_ptr(),
                enc.stride(0), enc.shape[0], gate.data_ptr(), rows.data_ptr() if rows is not None else 0,
                dXF.data_ptr(), dXF.stride(0), offset, _ptr_array(dW, n_layers), _ptr_array(db, n_layers),
                dg_view.data_ptr(), _stream(dXF.device)), "mrgcn_mlp_gate_scatter_bwd_f32")
        # the block this op wrote does not depend on what XF held before (index_fill_ / fill_ take the zero as a
        # kernel argument: indexed assignment of a Python scalar stages it through a host copy, which a stream
        # capture refuses)
        dXF_in = None
        if ctx.needs_input_grad[0] and ctx.fresh:
            # the block held constants before this op (a fresh zero buffer filled block by block): whoever made XF's
            # other blocks reads only those — the gradient goes on as it is, no copy of the whole matrix per encoder
            dXF_in = dXF
        elif ctx.needs_input_grad[0]:
            dXF_in = dXF.clone()
            blk = dXF_in[:, offset:offset + dims[-1]]
            if rows is not None:
                blk.index_fill_(0, rows, 0.0)
            else:
                blk.fill_(0.0)
        return (dXF_in, None, None, dgate, None, None, None, None, *dW, *db)


def mlp_gate_scatter(XF, enc, rows, gate_weights, i_gate, offset, weights, biases, fresh=False):
    """`fresh`: XF is a zero buffer that is being filled block by block (every block written once, nothing else reads
    it in between): the backward then hands XF's gradient on without copying it."""
    return _MlpGateScatter.apply(XF, enc, rows, gate_weights, int(i_gate), int(offset), len(weights), bool(fresh),
                                 *weights, *biases)


class _ScatterBlock(torch.autograd.Function):
    """XF[rows, off : off + d] = out, in place, for a zero buffer that is filled block by block (mrgcn.py:303): the
    backward gathers the block's rows and hands XF's gradient on untouched — autograd's own indexed assignment copies
    the whole gradient matrix (1.07 GB at the AM shape) once per encoder."""

    @staticmethod
    def forward(ctx, XF, out, rows, offset: int):
        d = out.shape[1]
        XF[rows, offset:offset + d] = out
        ctx.mark_dirty(XF)
        ctx.save_for_backward(rows)
        ctx.blk = (int(offset), int(d))
        return XF

    @staticmethod
    def backward(ctx, dXF):
        (rows,) = ctx.saved_tensors
        off, d = ctx.blk
        d_out = dXF[:, off:off + d].index_select(0, rows) if ctx.needs_input_grad[1] else None
        return dXF, d_out, None, None


def scatter_block(XF, out, rows, offset):
    return _ScatterBlock.apply(XF, out, rows, int(offset))
