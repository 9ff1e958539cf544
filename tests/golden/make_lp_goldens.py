#!/usr/bin/env python
"""Golden vectors for the link-prediction decoder, AUTHORING CONTAINER ONLY:
imports the reference's `mrgcn/tasks/link_prediction.py` (rdflib stubbed) and records
score_distmult_bc (:645-665), BCEWithLogits (:57, :550-554) and compute_ranks_fast (:593-643),
raw and filtered, on seeded toy embedding tables.   python tests/golden/make_lp_goldens.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_goldens import _stub_rdflib  # noqa: E402


def main():
    _stub_rdflib()
    sys.path.insert(0, "/root/reference")
    for k in [k for k in sys.modules if k == "mrgcn" or k.startswith("mrgcn.")]:
        del sys.modules[k]
    import mrgcn.tasks.link_prediction as lp

    out = {}
    rng = np.random.default_rng(42)
    for tag, (N, P, H, nf, relu) in {"a": (60, 5, 16, 40, False), "b": (300, 11, 200, 120, True),
                                     "c": (30, 3, 8, 45, True)}.items():
        E = rng.standard_normal((N, H)).astype(np.float32)
        if relu:  # encoder outputs pass a ReLU: exact zeros -> exact ties
            E = np.maximum(E, 0).astype(np.float32)
            E[rng.choice(N, N // 10, replace=False)] = 0.0
        Rel = rng.standard_normal((2 * P + 1, H)).astype(np.float32)
        facts = np.stack([rng.integers(0, N, nf), rng.integers(0, P, nf), rng.integers(0, N, nf)], 1)
        facts = np.unique(facts, axis=0)
        # a few facts sharing (p, o) / (s, p) so that the filter has something to do
        extra = facts[: len(facts) // 4].copy()
        extra[:, 0] = rng.integers(0, N, len(extra))
        facts = np.unique(np.concatenate([facts, extra]), axis=0).astype(np.int64)
        Et, Rt, Ft = torch.from_numpy(E), torch.from_numpy(Rel), torch.from_numpy(facts)
        sc = lp.score_distmult_bc((Ft[:, 0], Ft[:, 1], Ft[:, 2]), Et, Rt)
        y = torch.from_numpy((rng.random(len(facts)) < 0.8).astype(np.float32))
        crit = torch.nn.BCEWithLogitsLoss()
        Eg, Rg = Et.clone().requires_grad_(True), Rt.clone().requires_grad_(True)
        loss = lp.binary_crossentropy(lp.score_distmult_bc((Ft[:, 0], Ft[:, 1], Ft[:, 2]), Eg, Rg), y, crit)
        loss.backward()
        out.update({f"{tag}.E": E, f"{tag}.Rel": Rel, f"{tag}.facts": facts, f"{tag}.scores": sc.numpy(),
                    f"{tag}.y": y.numpy(), f"{tag}.loss": np.float32(loss.item()),
                    f"{tag}.dE": Eg.grad.numpy(), f"{tag}.dRel": Rg.grad.numpy(),
                    f"{tag}.ranks_raw": lp.compute_ranks_fast(Ft, Et, Rt, 50, filtered=False).numpy(),
                    f"{tag}.ranks_flt": lp.compute_ranks_fast(Ft, Et, Rt, 50, filtered=True).numpy()})
        print(tag, "facts", len(facts), "mrr raw", float((1.0 / out[f"{tag}.ranks_raw"]).mean()),
              "flt", float((1.0 / out[f"{tag}.ranks_flt"]).mean()))
    np.savez_compressed(os.path.join(HERE, "lp_decoder.npz"), **out)


if __name__ == "__main__":
    main()
