#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/r6; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_bf16_pipeline.py -x -q -k "model_in_bf16" > $o/t2.txt 2>&1; tail -5 $o/t2.txt
timeout 1200 python -m pytest tests/test_gpu_step_oracle.py -x -q -k "am_bf16 or am_gradients" > $o/t3.txt 2>&1; tail -8 $o/t3.txt
F="--no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_bf16 -o run -- python3 bench.py --operand bf16 --steps 20 --warmup 3 $F > $o/bench_bf16_prof.json 2> $o/bench_bf16_prof.err
python3 tools/epoch_sequence.py $o/stats_bf16 k_xform_bf16_fwd > $o/bf16_epoch_sequence.md 2>&1
rm -rf $o/stats_bf16
cat $o/bf16_epoch_sequence.md | head -60
