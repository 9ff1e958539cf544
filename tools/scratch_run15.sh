#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/r6; mkdir -p $o
hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC tools/lab/spmm_hot_lab.hip -o tools/lab/libspmm_hot_lab.so
bash tools/pmc_passes.sh $o/pmc_hot mem -- python3 tools/lab/spmm_hot_lab.py --H 0 3400 --iters 3
python3 tools/pmc_summary.py $o k_hot > $o/spmm_hot_pmc.md 2>&1
python3 tools/pmc_summary.py $o k_spmm3 > $o/spmm3_beside_hot_pmc.md 2>&1
rm -rf $o/pmc_hot_*/
cat $o/spmm_hot_pmc.md
