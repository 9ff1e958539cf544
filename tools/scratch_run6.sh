#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
F="--no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm"
for v in 0 2 1 4 0 2; do
  MRGCN_ADAM_ONCE=$v timeout 600 python bench.py $F --steps 40 > $o/bench_once_$v.json 2> $o/bench_once_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6/bench_once_$v.json").read().strip().splitlines()[-1]); print("adam_once=$v", round(d["ms_per_step"],4))
PY
done
timeout 1500 python -m pytest tests/test_gpu_support.py tests/test_gpu_step_oracle.py tests/test_gpu_layers.py -x -q > $o/t6.txt 2>&1; tail -6 $o/t6.txt
