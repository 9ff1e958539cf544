#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_plan_spmm.py -x -q > $o/t9.txt 2>&1; tail -4 $o/t9.txt
timeout 600 python tools/kernel_probe.py --which spmm_t10,spmm_t11,spmm --ldm 10 --iters 20 --sweep "spmm_t_seg=0;spmm_t_seg=1;spmm_t_seg=0;spmm_t_seg=1" > $o/probe_tseg.txt 2>&1; grep -v amdgpu $o/probe_tseg.txt
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_support.py tests/test_gpu_bf16.py tests/test_gpu_partition.py -x -q > $o/t10.txt 2>&1; tail -4 $o/t10.txt
