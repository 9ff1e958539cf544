#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/lab/copy_lab.hip -o /tmp/copy_lab && /tmp/copy_lab > $o/copy_lab3.txt 2>&1; cat $o/copy_lab3.txt
