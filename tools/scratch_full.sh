#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
timeout 2400 python -m pytest tests -q -m gpu -x > $o/full_gpu.txt 2>&1; tail -12 $o/full_gpu.txt
