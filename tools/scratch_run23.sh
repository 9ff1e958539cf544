#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
F="--no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm"
for w in mutag aifb mutag; do timeout 300 python bench.py --workload $w --steps 200 --warmup 10 $F > $o/sm_$w.json 2>$o/sm_$w.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r6/sm_$w.json").read().strip().splitlines()[-1]); print("$w", round(d["ms_per_step"],4))
PY
done
timeout 900 python -m pytest tests/test_gpu_support.py tests/test_gpu_layers.py -x -q > $o/t18.txt 2>&1; tail -3 $o/t18.txt
