#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_layers.py -x -q -k "line_aligned or fused_layer or ticket" > $o/t16.txt 2>&1; tail -4 $o/t16.txt
F="--no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm"
for v in 0 1 0 1; do
  MRGCN_X_LINE_ROWS=$v timeout 600 python bench.py $F --steps 40 > $o/xl_$v.json 2> $o/xl_$v.err
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/r6/xl_$v.json").read().strip().splitlines()[-1]); print("x_line_rows=$v epoch", round(d["ms_per_step"],4))
PY
done
