#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_layers.py tests/test_gpu_lp.py tests/test_gpu_step_oracle.py -x -q -k "wide or lp or fb15k or distmult" > $o/t12.txt 2>&1; tail -5 $o/t12.txt
for v in 1 0 1 0; do MRGCN_WIDE_DET=$v timeout 600 python bench.py --workload fb15k --no-cpu-baseline > $o/fb_$v.json 2>$o/fb_$v.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r6/fb_$v.json").read().strip().splitlines()[-1]); print("wide_det=$v", round(d["ms_per_step"],4), d["roofline"]["frac"])
PY
done
