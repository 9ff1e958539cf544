#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/r6; mkdir -p $o
F="--no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm"
for v in 0 1 0 1; do
  MRGCN_SPMM_REVERSE=$v rocprofv3 --kernel-trace --stats --output-format csv -d $o/st_$v -o run -- python3 bench.py --steps 30 --warmup 3 $F > $o/st_$v.json 2> $o/st_$v.err
  python3 tools/epoch_sequence.py $o/st_$v "k_xform_mfma_fwd<1, false, 10" 2>/dev/null | grep -E "k_spmm3|epoch ="
  rm -rf $o/st_$v
done
