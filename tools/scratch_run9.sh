#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/r6; mkdir -p $o
timeout 1500 python -m pytest tests/test_encoders.py tests/test_gpu_configs.py -x -q -m gpu > $o/t7.txt 2>&1; tail -5 $o/t7.txt
rocprofv3 --kernel-trace --output-format csv -d $o/stats_encb -o run -- python3 tools/am_encoders_step.py run bf16 > $o/am_encoders_step_bf16.txt 2> $o/am_encoders_step_bf16.err
python3 tools/am_encoders_step.py summary $o/stats_encb > $o/am_encoders_step_bf16.md 2>> $o/am_encoders_step_bf16.err
rm -rf $o/stats_encb
head -70 $o/am_encoders_step_bf16.md
