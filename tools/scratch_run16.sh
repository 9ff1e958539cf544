#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/r6; mkdir -p $o
F="--no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm"
for v in 0 1 0 1; do
  MRGCN_SUP_MIX_ONCE=$v rocprofv3 --kernel-trace --stats --output-format csv -d $o/st_$v -o run -- python3 bench.py --steps 30 --warmup 3 $F > $o/st_$v.json 2> $o/st_$v.err
  python3 tools/prof_summary.py $o/st_$v 14 2>/dev/null | grep -E "k_adam_rows|k_mix_fwd|k_mix_bwd|k_xform_mfma_fwd<1|k_xform_mfma_dw<3|k_dcomp" | head -7
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/r6/st_$v.json").read().strip().splitlines()[-1]); print("sup_mix_once=$v epoch", round(d["ms_per_step"],4))
PY
  rm -rf $o/st_$v
done
