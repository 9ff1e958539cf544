#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
timeout 1800 python -m pytest tests/test_gpu_partition.py -x -q > $o/t11.txt 2>&1; tail -15 $o/t11.txt
