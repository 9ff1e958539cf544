"""debug: at full AM size, which engine's logits match the float64 oracle at sampled rows; loss sequences"""
import sys, os
import numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrgcn_amd import synth
from mrgcn_amd.models.rgcn import RGCN
from mrgcn_amd.train import ClipAdam, train_step, GraphedTrainStep
from oracle import rgcn_oracle as O

g = synth.make_graph("am", seed=0)
N, R = g.num_nodes, g.num_relations
A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals), (N, R * N)).cuda()
A_csr = sp.csr_matrix((g.vals.astype(np.float64), (g.rows, g.cols)), shape=(N, R * N))
dims = synth.layer_dims("am"); B = 40
idx_np, y_np = synth.make_labels("am", N, seed=0)
idx, y = torch.from_numpy(idx_np).cuda(), torch.from_numpy(y_np).cuda()
X = torch.randn((N, dims[0][0]), device="cuda", generator=torch.Generator("cuda").manual_seed(6))
mods = [(dims[0][0], dims[0][1], "mrgcn", torch.nn.ReLU()), (dims[1][0], dims[1][1], "mrgcn", None)]
rows = np.unique(np.concatenate([idx_np[:60], np.random.default_rng(0).choice(N, 60)]))
torch.manual_seed(5)
m = RGCN(mods, R, N, B, 0.0, False, True, False).cuda()
init = {k: v.detach().clone() for k, v in m.state_dict().items()}
state = {k: v.cpu().numpy() for k, v in init.items()}
cfgs = O.rgcn_cfgs(dims, R, N, B, True, False)
ref = O.rgcn_forward_at_rows(cfgs, O.split_params(state, 2), X.cpu().numpy(), A_csr, rows)
for eng in ("fused", "literal"):
    m.set_engine(eng)
    with torch.no_grad():
        got = m(X, A)[torch.from_numpy(rows).cuda()].cpu().numpy()
    print(eng, "max |logit - oracle|", float(np.abs(got - ref).max()), flush=True)
for name, engine, rs, graphed in (("fused/row-sparse/graph", "fused", None, True), ("fused/dense/eager", "fused", False, False),
                                  ("literal/dense/eager", "literal", False, False)):
    m.load_state_dict(init); m.set_engine(engine)
    opt = ClipAdam(m.parameters(), lr=0.01, max_norm=1.0, capturable=graphed)
    if graphed:
        st = GraphedTrainStep(m, lambda: m(X, A), idx, y, opt, warmup=1)
        losses = [float("nan")] + [float(st()) for _ in range(2)]
    else:
        losses = [float(train_step(m, lambda: m(X, A), idx, y, opt, row_sparse=rs)) for _ in range(3)]
    print(name, losses, flush=True)
