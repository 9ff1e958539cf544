#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
for ld in 155 160 156 192; do
timeout 300 python tools/kernel_probe.py --which xf_fwd0,xf_bwd0,xf_fwd0 --ldm 10 --iters 20 --x-ld $ld 2>&1 | grep -E "xf_|N=" | sed "s/^/x-ld=$ld  /"
done
