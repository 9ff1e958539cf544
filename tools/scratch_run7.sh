#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/r6; mkdir -p $o
F="--no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm"
for v in 0 1 2 0 1 2; do
  MRGCN_ADAM_ONCE=$v rocprofv3 --kernel-trace --stats --output-format csv -d $o/st_$v -o run -- python3 bench.py --steps 30 --warmup 3 $F > /dev/null 2> $o/st_$v.err
  python3 tools/prof_summary.py $o/st_$v 12 2>/dev/null | grep -E "k_adam_rows|k_mix_fwd|k_mix_bwd|k_xform_mfma_fwd|k_xform_mfma_dw" | head -6
  rm -rf $o/st_$v
  echo "---- adam_once=$v"
done
