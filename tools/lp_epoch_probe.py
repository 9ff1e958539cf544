#!/usr/bin/env python
"""One full-batch link-prediction epoch at BASELINE config 4's shape (FB15k-237: 14 541 nodes, 237
predicates -> R = 475, 310 116 triples, one featureless R-GCN layer -> 200, 2 bases, DistMult
decoder; configs/fb15k-237.toml): encoder forward, 20 % in-batch negatives, scores, BCE, backward,
clip, Adam — then filtered + raw ranks of 500 test facts.  Synthetic graph of that shape.
    python tools/lp_epoch_probe.py [--steps 20] [--cpu-steps 1]"""
import argparse
import json
import time

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mrgcn_amd import synth
from mrgcn_amd.models.rgcn import RGCN
from mrgcn_amd.tasks import link_prediction as lp
from mrgcn_amd.train import ClipAdam


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--train-frac", type=float, default=0.877)  # 272 115 of 310 116 triples train
    ap.add_argument("--cpu-steps", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sh = synth.SHAPES["fb15k"]
    g = synth.make_graph("fb15k", seed=0, value_mode="ref_int8")
    N, R, H, B = g.num_nodes, g.num_relations, sh["hidden"], sh["bases"]
    tr = synth.make_triples(N, sh["P"], sh["T"], 0)
    rng = np.random.RandomState(0)
    perm = rng.permutation(len(tr))
    ntrain = int(a.train_frac * len(tr))
    train, test = tr[perm[:ntrain]], tr[perm[ntrain:ntrain + 500]]
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g.rows, g.cols])), torch.from_numpy(g.vals),
                                (N, R * N)).to(dev)
    torch.manual_seed(0)
    model = RGCN([(0, H, "rgcn", torch.nn.ReLU())], R, N, B, 0.0, True, False, True).to(dev)
    opt = ClipAdam(list(model.parameters()), lr=0.01)
    train_dev = torch.from_numpy(train).to(dev)

    def step():
        neg, Y = lp.sample_negatives(train, rng)          # host numpy, as the reference
        emb = model(None, A)
        t = torch.cat([train_dev, torch.from_numpy(neg).to(dev)])
        sc = lp.score_distmult_bc((t[:, 0], t[:, 1], t[:, 2]), emb, model.relations)
        loss = lp.binary_crossentropy(sc, torch.from_numpy(Y).to(dev))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    # the same step with the negative sampling hoisted (device-side work only)
    neg, Y = lp.sample_negatives(train, rng)
    t_all = torch.cat([train_dev, torch.from_numpy(neg).to(dev)])
    Yd = torch.from_numpy(Y).to(dev)

    def dev_step():
        emb = model(None, A)
        sc = lp.score_distmult_bc((t_all[:, 0], t_all[:, 1], t_all[:, 2]), emb, model.relations)
        l = lp.binary_crossentropy(sc, Yd)
        opt.zero_grad(set_to_none=True)
        l.backward()
        opt.step()

    dev_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        dev_step()
    torch.cuda.synchronize()
    dev_ms = (time.perf_counter() - t0) / a.steps * 1e3
    gen = torch.Generator(device=dev).manual_seed(0)

    def dev_sampled_step():  # negatives drawn on the device: nothing of the step runs on the host
        neg, Yn = lp.sample_negatives_device(train_dev, gen)
        t = torch.cat([train_dev, neg])
        emb = model(None, A)
        sc = lp.score_distmult_bc((t[:, 0], t[:, 1], t[:, 2]), emb, model.relations)
        l = lp.binary_crossentropy(sc, Yn)
        opt.zero_grad(set_to_none=True)
        l.backward()
        opt.step()

    dev_sampled_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        dev_sampled_step()
    torch.cuda.synchronize()
    dev_sampled_ms = (time.perf_counter() - t0) / a.steps * 1e3
    with torch.no_grad():
        emb = model(None, A)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        raw = lp.compute_ranks_fast(test, emb, model.relations, filtered=False)
        flt = lp.compute_ranks_fast(test, emb, model.relations, filtered=True)
        torch.cuda.synchronize()
        rank_ms = (time.perf_counter() - t0) * 1e3
    out = {"shape": {"N": N, "R": R, "nnz": len(g.rows), "hidden": H, "bases": B, "train_triples": ntrain},
           "epoch_ms_host_sampling": round(ms, 3), "epoch_ms_device_sampling": round(dev_sampled_ms, 3),
           "epoch_ms_fixed_negatives": round(dev_ms, 3), "loss": float(loss),
           "ranks_500_raw_plus_filtered_ms": round(rank_ms, 3),
           "mrr_raw": lp.mrr_hits(raw)[0], "mrr_flt": lp.mrr_hits(flt)[0]}
    if a.cpu_steps > 0:
        # the reference's op sequence for the same epoch on the host (oracle port of the encoder +
        # torch ops of tasks/link_prediction.py:239-300)
        from oracle import aten_literal as ref
        p = ref.make_params([(0, H)], R, N, B, False, True, seed=0)
        p["relations"] = torch.nn.init.xavier_uniform_(torch.empty((R, H))).requires_grad_(True)
        Ac = ref.coo_tensor(g.rows, g.cols, g.vals, (N, R * N))
        optc = torch.optim.Adam(list(p.values()), lr=0.01)
        crit = torch.nn.BCEWithLogitsLoss()
        tt = t_all.cpu()
        t0 = time.perf_counter()
        for _ in range(a.cpu_steps):
            emb = torch.relu(ref.layer_forward(p, "layers.layer_0.", None, Ac, R, N, B, True, True))
            sc = torch.sum(emb[tt[:, 0]] * p["relations"][tt[:, 1]] * emb[tt[:, 2]], dim=-1)
            l = crit(sc, Yd.cpu())
            optc.zero_grad()
            l.backward()
            torch.nn.utils.clip_grad_norm_(list(p.values()), 1.0)
            optc.step()
        out["cpu_epoch_ms"] = round((time.perf_counter() - t0) / a.cpu_steps * 1e3, 1)
        out["cpu_threads"] = torch.get_num_threads()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
