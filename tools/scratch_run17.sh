#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/r6; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_support.py -x -q -k "ticket or row_adam or support" > $o/t14.txt 2>&1; tail -4 $o/t14.txt
MRGCN_SUP_MIX_ONCE=2 timeout 900 python -m pytest tests/test_gpu_support.py tests/test_gpu_step_oracle.py -x -q -k "support or am_gradients" > $o/t15.txt 2>&1; tail -4 $o/t15.txt
F="--no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm"
for v in 0 2 0 2; do
  MRGCN_SUP_MIX_ONCE=$v rocprofv3 --kernel-trace --stats --output-format csv -d $o/st_$v -o run -- python3 bench.py --steps 30 --warmup 3 $F > $o/st_$v.json 2> $o/st_$v.err
  python3 tools/prof_summary.py $o/st_$v 14 2>/dev/null | grep -E "k_mix_bwd|k_xform_mfma_dw<3|k_dcomp_chunks" | head -4
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/r6/st_$v.json").read().strip().splitlines()[-1]); print("sup_mix_once=$v epoch", round(d["ms_per_step"],4))
PY
  rm -rf $o/st_$v
done
