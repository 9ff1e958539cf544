#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/r6; mkdir -p $o
F="--no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm"
for v in 0 16 0 16; do
  MRGCN_PAD_OUT_LD=$v rocprofv3 --kernel-trace --stats --output-format csv -d $o/st_$v -o run -- python3 bench.py --steps 30 --warmup 3 $F > $o/st_$v.json 2> $o/st_$v.err
  python3 tools/prof_summary.py $o/st_$v 16 2>/dev/null | grep -E "k_xform_cols_lds|k_spmm3|k_segment_sum_mask" | head -4
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/r6/st_$v.json").read().strip().splitlines()[-1]); print("pad_out_ld=$v epoch", round(d["ms_per_step"],4))
PY
  rm -rf $o/st_$v
done
