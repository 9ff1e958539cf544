#!/usr/bin/env python
"""Measurements for the SURVEY §8(f) rows around the hot path (one JSON object on stdout):
  mini-batch   device-built batch structure, slice plans and a train step on a re-sampled batch (AM/4 shape)
  encoders     fused MLP + gate + scatter; TCNN (implicit-im2col MFMA convolutions + fused BatchNorm / ReLU / pool)
               forward + backward, next to the same modules on torch.nn (MIOpen / rocBLAS) on the same GPU
  ingestion    CSR of the dataset archive -> graph plan on the device (AM shape)
    python tools/next_rows_probe.py > profiles/rNN_next_rows.json"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrgcn_amd import synth  # noqa: E402
from mrgcn_amd.host import fit_cpu_pool_to_quota  # noqa: E402

fit_cpu_pool_to_quota()   # (a container CPU quota: mrgcn_amd/host.py)


def timed(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def minibatch():
    from mrgcn_amd.data import batch as mb
    from mrgcn_amd.models.rgcn import RGCN
    from mrgcn_amd.train import ClipAdam, categorical_crossentropy
    g = synth.make_graph("am", seed=0, scale=0.25)
    N, R = g.num_nodes, g.num_relations
    A = sp.csr_matrix((g.vals, (g.rows, g.cols)), shape=(N, R * N))
    dcsr = mb.DeviceCSR(A)
    rng = np.random.default_rng(0)
    dims = synth.layer_dims("am")
    mods = [(dims[0][0], dims[0][1], "mrgcn", torch.nn.ReLU()), (dims[1][0], dims[1][1], "mrgcn", None)]
    torch.manual_seed(0)
    model = RGCN(mods, R, N, synth.SHAPES["am"]["bases"], 0.0, False, True, False).cuda()
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0)
    X = torch.randn((N, dims[0][0]), device="cuda")
    out = {"graph": f"am x 0.25 (N={N}, R={R}, nnz={A.nnz})", "batch_nodes": 1024, "layers": 2}
    t_build, t_step, sizes = [], [], None
    for it in range(6):
        idx = np.sort(rng.choice(N, 1024, replace=False))
        y = torch.from_numpy(rng.integers(0, dims[-1][1], 1024)).cuda()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ab = mb.A_BatchDevice(dcsr, idx, 2, short_lived=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        logits = model(X[ab.neighbours[-1]], ab)           # builds the slice plans on first use
        loss = categorical_crossentropy(logits, torch.arange(1024, device="cuda"), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it >= 2:
            t_build.append((t1 - t0) * 1e3)
            t_step.append((t2 - t1) * 1e3)
        sizes = [int(n.numel()) for n in ab.neighbours]
    # the reference builds its batches once and walks the same ones every epoch (node_classification.py: mkbatches
    # before the epoch loop): a step on a batch whose slice plans already exist
    def again():
        loss = categorical_crossentropy(model(X[ab.neighbours[-1]], ab), torch.arange(1024, device="cuda"), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    out["reused_batch_fwd_bwd_adam_ms"] = round(timed(again, iters=10, warm=2), 2)
    # the same re-sampled steps with the next batch (structure + slice plans) prepared by a worker thread on its own
    # stream while the current one trains: steady-state time per step, every step on a fresh batch
    steps, warm = 24, 6
    ys = torch.from_numpy(rng.integers(0, dims[-1][1], 1024)).cuda()
    rows1024 = torch.arange(1024, device="cuda")
    out["resampled_step_prefetched_ms"] = {}
    for workers in (1, 2, 3):
        pf = mb.BatchPrefetcher(dcsr, (np.sort(rng.choice(N, 1024, replace=False)) for _ in range(steps)), 2,
                                model=model, workers=workers)
        t0 = None
        for k, b in enumerate(pf):
            if k == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            loss = categorical_crossentropy(model(X[b.neighbours[-1]], b), rows1024, ys)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        out["resampled_step_prefetched_ms"][f"workers={workers}"] = round((time.perf_counter() - t0) / (steps - warm) * 1e3, 2)
        pf.close()
    # the batch as a masked pass over the FULL graph's plan (data.batch.A_BatchMasked, csrc/masked.hip): a re-sampled
    # step = two support builds on the existing plan + forward / backward / clip / Adam on compact arrays; no slices, no
    # per-batch plans, no worker threads
    from mrgcn_amd.plan import GraphPlan
    plan = GraphPlan.from_csr(A, N, R, value_mode="ref_int8", operand_row_bytes=model.operand_row_bytes())
    mb_build, mb_step = [], []
    for it in range(8):
        idx = np.sort(rng.choice(N, 1024, replace=False))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        am = mb.A_BatchMasked(plan, idx, 2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss = categorical_crossentropy(model(X[am.neighbours[-1]], am), rows1024, ys)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it >= 2:
            mb_build.append((t1 - t0) * 1e3)
            mb_step.append((t2 - t1) * 1e3)
        sup_sizes = [(s_.NR, s_.L, s_.E, s_.NL) for s_ in am.supports]
        am.close()
    from mrgcn_amd.train import train_step
    steps, warm = 24, 6
    idxs = [np.sort(rng.choice(N, 1024, replace=False)) for _ in range(steps)]

    def in_line(step_fn):
        for k, idx in enumerate(idxs):
            if k == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            am = mb.A_BatchMasked(plan, idx, 2)
            step_fn(am)
            am.close()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / (steps - warm) * 1e3, 2), am

    def raw_loop(am):  # the reference's loop shape: neighbours' rows gathered by the caller, backward, clip + Adam
        loss = categorical_crossentropy(model(X[am.neighbours[-1]], am), rows1024, ys)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    raw_ms, _ = in_line(raw_loop)
    # the package's step: weight_I's gradient stays row-sparse (fused clip + Adam on the rows that have one or ever
    # had), X handed over whole (the transform picks the neighbours' rows)
    step_ms, am = in_line(lambda am: train_step(model, lambda: model(X, am), rows1024, ys, opt))
    out["masked_pass"] = dict(
        resampled_step_in_line_ms=step_ms, resampled_step_in_line_raw_loop_ms=raw_ms,
        support_builds_ms=round(float(np.median(mb_build)), 2), fwd_bwd_adam_ms=round(float(np.median(mb_step)), 2),
        supports=[dict(rows=s_[0], live_cols=s_[1], entries=s_[2], live_nodes=s_[3]) for s_ in sup_sizes],
        note="every step on a fresh batch, nothing prepared ahead: the batch is a pair of forward supports on the full "
             "graph's plan (data.batch.A_BatchMasked), built in one call with one host wait; resampled_step_in_line_ms = "
             "mrgcn_amd.train.train_step on it, _raw_loop_ms = gather + backward + ClipAdam.step written out like the "
             "reference's loop (dense weight_I gradient)")
    out.update(neighbours=sizes, batch_build_ms=round(float(np.median(t_build)), 2),
               plans_fwd_bwd_adam_ms=round(float(np.median(t_step)), 2),
               note="a re-sampled batch every step: structure on the device (batch_build_ms), then its lean slice plans "
                    "(built side by side, blocks from the plan allocator's cache), forward, backward, clip, Adam "
                    "(plans_fwd_bwd_adam_ms); resampled_step_prefetched_ms = the whole of it per step (structure + plans + "
                    "training step, every step on a fresh batch) when worker threads prepare the next batches "
                    "meanwhile (data.batch.BatchPrefetcher, default 3 workers)")
    return out


def encoders():
    from mrgcn_amd import dense
    from mrgcn_amd.models.perceptron import MLP
    from mrgcn_amd.models.temporal_cnn import TCNN
    out = {}
    # fused MLP + gate + scatter: 1 M numeric literals, 4 -> 3
    n = 1_000_000
    torch.manual_seed(0)
    mlp = MLP(4, 3, num_layers=1).cuda()
    enc = torch.randn((n, 4), device="cuda")
    rows = torch.randperm(2 * n, device="cuda")[:n].sort().values
    gates = torch.full((2,), 0.1, device="cuda", requires_grad=True)
    lin = mlp.linears()

    def fused():
        XF = torch.zeros((2 * n, 8), device="cuda")
        y = dense.mlp_gate_scatter(XF, enc, rows, gates, 0, 2, [l.weight for l in lin], [l.bias for l in lin])
        y.square().sum().backward()

    def library():
        XF = torch.zeros((2 * n, 8), device="cuda")
        XF[rows, 2:5] = gates[0] * mlp(enc)
        XF.square().sum().backward()

    out["mlp_gate_scatter_1M_literals"] = {"hip_fwd_bwd_ms": round(timed(fused), 3),
                                           "torch_nn_fwd_bwd_ms": round(timed(library), 3)}
    # TCNN M: 2048 WKT literals, 37-symbol alphabet, 300 positions
    torch.manual_seed(0)
    m = TCNN(37, 16, p_dropout=0.0, size="M").cuda().train()
    x = (torch.rand((2048, 37, 300), device="cuda") < 0.03).float()

    def hip():
        for p in m.parameters():
            p.grad = None
        m(x).square().sum().backward()

    def lib():
        for p in m.parameters():
            p.grad = None
        y = m.conv(x)
        m.fc(y.view(y.size(0), -1)).square().sum().backward()

    # multiply-adds of the convolutions and the fully connected tail, forward; backward = 2x
    flops, T, cin = 0, 300, 37
    for mod in m.conv:
        if isinstance(mod, torch.nn.Conv1d):
            Tout = T + 2 * mod.padding[0] - mod.kernel_size[0] + 1
            flops += 2 * 2048 * Tout * mod.out_channels * cin * mod.kernel_size[0]
            T, cin = Tout, mod.out_channels
        elif isinstance(mod, torch.nn.MaxPool1d):
            T = (T - mod.kernel_size) // mod.kernel_size + 1
        elif isinstance(mod, torch.nn.AdaptiveMaxPool1d):
            T = mod.output_size
    flops += 2 * 2048 * (cin * cin + cin * 16)
    t_h, t_l = timed(hip, 5, 2), timed(lib, 5, 2)
    out["tcnn_M_2048x37x300"] = {"hip_fwd_bwd_ms": round(t_h, 2), "torch_nn_fwd_bwd_ms": round(t_l, 2),
                                 "gflop_fwd": round(flops / 1e9, 1),
                                 "hip_tflops_f32": round(3 * flops / (t_h * 1e-3) / 1e12, 1),
                                 "note": "fp32 throughout (v_mfma_f32_16x16x4_f32, exact; 157 TFLOP/s peak); the library "
                                         "path is MIOpen's fp32 Conv1d / BatchNorm and rocBLAS"}
    return out


def mrgcn_epoch():
    """BASELINE config 1 / 2 style: MRGCN(FullBatch) at the MUTAG shape with a numeric MLP over 6 000 literal nodes and
    a WKT TCNN (S) over 500, full-batch epoch = encoders + gates + 2 R-GCN layers + CE + backward + clip + Adam."""
    from mrgcn_amd.data.batch import FullBatch
    from mrgcn_amd.models.mrgcn import MRGCN
    from mrgcn_amd.train import ClipAdam, train_step
    g = synth.make_graph("mutag", seed=5, scale=1.0, value_mode="norm_f32")
    N, R = g.num_nodes, g.num_relations
    rng = np.random.default_rng(8)
    num_idx = np.sort(rng.choice(N, 6000, replace=False))
    num = rng.standard_normal((6000, 4)).astype(np.float32)
    wkt_idx = np.sort(rng.choice(N, 500, replace=False))
    wkt = (rng.random((500, 9, 20)) < 0.15).astype(np.float32)
    torch.manual_seed(12)
    emb_cfg = sorted([("ogc.wktLiteral", (9, 5, "S", 0.0), False), ("xsd.numeric", (4, 3, 0.0), False)],
                     key=lambda t: t[0])
    modules = [(8, 16, "mrgcn", torch.nn.ReLU()), (16, 2, "mrgcn", None)]
    model = MRGCN(modules, emb_cfg, R, N, num_bases=30, p_dropout=0.0, featureless=False, bias=True,
                  gcn_gpu_acceleration=True)
    A = sp.csr_matrix((g.vals, (g.rows, g.cols)), shape=(N, R * N))
    X = [np.empty((N, 0), dtype=np.float32),
         ["ogc.wktLiteral", [[wkt, wkt_idx, np.full(500, 20)]], False],
         ["xsd.numeric", [[num, num_idx, np.ones(6000, dtype=int)]], False]]
    batch = FullBatch(A, X, np.arange(N), value_mode="norm_f32")
    batch.as_tensors_()
    batch.to(model.devices)
    model.train()
    idx = torch.from_numpy(np.sort(rng.choice(N, 340, replace=False))).cuda()
    y = torch.from_numpy(rng.integers(0, 2, 340)).cuda()
    from mrgcn_amd.train import GraphedTrainStep
    opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0, capturable=True)
    ms = timed(lambda: train_step(model, lambda: model(batch), idx, y, opt), iters=30, warm=3)
    out = {"graph": f"mutag (N={N}, R={R}, nnz={A.nnz}), 30 bases, 8 -> 16 -> 2",
           "literals": "6000 numeric (MLP), 500 WKT (TCNN S)", "epoch_ms_eager": round(ms, 3)}
    try:
        step = GraphedTrainStep(model, lambda: model(batch), idx, y, opt, warmup=1)
        out["epoch_ms_hipgraph"] = round(timed(step, iters=30, warm=3), 3)
    except Exception as e:  # noqa: BLE001
        out["hipgraph_error"] = repr(e)[:200]
    return out


def ingestion():
    from mrgcn_amd.plan import GraphPlan
    g = synth.make_graph("am", seed=0, scale=1.0)
    N, R = g.num_nodes, g.num_relations
    A = sp.csr_matrix((g.vals, (g.rows, g.cols)), shape=(N, R * N))
    A.sort_indices()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p = GraphPlan.from_csr(A, N, R, value_mode="norm_f32")
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        mb_ = p.device_bytes / 2 ** 20
        del p
    return {"graph": f"am (N={N}, R={R}, nnz={A.nnz})", "csr_upload_and_plan_build_ms": round(min(ts), 1),
            "plan_device_mb": round(mb_, 1)}


def main():
    import sys
    res = {"device": torch.cuda.get_device_name(0)}
    only = set(sys.argv[1:])  # e.g. `next_rows_probe.py minibatch`
    for name, fn in (("minibatch", minibatch), ("encoders", encoders), ("mrgcn_with_encoders", mrgcn_epoch),
                     ("ingestion", ingestion)):
        if only and name not in only:
            continue
        try:
            res[name] = fn()
        except Exception as e:  # noqa: BLE001
            import traceback
            res[name] = {"error": repr(e)[:300], "where": traceback.format_exc()[-600:]}
        torch.cuda.empty_cache()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
