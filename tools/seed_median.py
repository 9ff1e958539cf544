#!/usr/bin/env python3
"""the headline measured on the graphs of seeds 0, 1, 2 — one bench.py run per seed (each its own
process), per-seed lines kept, medians reported.
usage: python tools/seed_median.py [--steps 20] [--warmup 3] > profiles/rNN_seeds.json"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2])
    a = ap.parse_args()
    runs = []
    for seed in a.seeds:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup",
                              str(a.warmup), "--seed", str(seed), "--no-cpu-baseline", "--no-renumbered-extra",
                              "--no-literal-spmm", "--no-reference-loop"], capture_output=True, text=True, cwd=ROOT)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if out.returncode != 0 or not line:
            runs.append({"seed": seed, "error": (out.stderr or out.stdout)[-400:]})
            continue
        j = json.loads(line[-1])
        runs.append({"seed": seed, "ms_per_step": j["ms_per_step"], "spmm_avg_ms": j["roofline"]["avg_ms"],
                     "roofline_frac": j["roofline"]["frac"], "nnz": j["config"]["nnz"],
                     "ncols_touched": j["config"]["ncols_touched"], "final_loss": j["extra"]["final_loss"]})
    ok = [r for r in runs if "error" not in r]
    res = {"runs": runs}
    if ok:
        res["median"] = {k: statistics.median(r[k] for r in ok) for k in ("ms_per_step", "spmm_avg_ms", "roofline_frac")}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
