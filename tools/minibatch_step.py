#!/usr/bin/env python
"""Re-sampled mini-batch steps on the AM/4 graph (1 024 batch nodes, two layers) as masked passes over the full graph's
plan (data.batch.A_BatchMasked + train.train_step), every step on a fresh batch: per step the host and device time of the
support builds and of the training step.  Under rocprofv3 --kernel-trace its last step gives the launch sequence
(tools/epoch_sequence.py <dir> k_sup_rowcount 2 -> profiles/rNN_minibatch_step_sequence.md; MRGCN_SUP_TIMING=1 adds the
build's host phases).    usage: python tools/minibatch_step.py [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from mrgcn_amd import synth
from mrgcn_amd.host import fit_cpu_pool_to_quota
fit_cpu_pool_to_quota()   # (a container CPU quota: mrgcn_amd/host.py)
from mrgcn_amd.data import batch as mb
from mrgcn_amd.models.rgcn import RGCN
from mrgcn_amd.plan import GraphPlan
from mrgcn_amd.train import ClipAdam, categorical_crossentropy, train_step
g = synth.make_graph("am", seed=0, scale=0.25)
N, R = g.num_nodes, g.num_relations
A = sp.csr_matrix((g.vals, (g.rows, g.cols)), shape=(N, R * N))
rng = np.random.default_rng(0)
dims = synth.layer_dims("am")
mods = [(dims[0][0], dims[0][1], "mrgcn", torch.nn.ReLU()), (dims[1][0], dims[1][1], "mrgcn", None)]
torch.manual_seed(0)
model = RGCN(mods, R, N, synth.SHAPES["am"]["bases"], 0.0, False, True, False).cuda()
opt = ClipAdam(model.parameters(), lr=0.01, max_norm=1.0)
X = torch.randn((N, dims[0][0]), device="cuda")
plan = GraphPlan.from_csr(A, N, R, value_mode="ref_int8", operand_row_bytes=model.operand_row_bytes())
ys = torch.from_numpy(rng.integers(0, dims[-1][1], 1024)).cuda()
rows1024 = torch.arange(1024, device="cuda")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
tb = ts = 0.0
for k in range(steps):
    idx = np.sort(rng.choice(N, 1024, replace=False))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    am = mb.A_BatchMasked(plan, idx, 2)
    t1h = time.perf_counter()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss = train_step(model, lambda: model(X, am), rows1024, ys, opt)
    t2h = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    am.close()
    if k >= 4:
        print(f"build host {1e3*(t1h-t0):.2f} total {1e3*(t1-t0):.2f} | step host {1e3*(t2h-t1):.2f} total {1e3*(t2-t1):.2f}", flush=True)
