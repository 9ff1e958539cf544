#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/r6; mkdir -p $o
timeout 300 python tools/kernel_probe.py --which xf_fwd0,xf_fwd0_xb,xf_fwd1,xf_bwd0,xf_bwd1,xf_fwd0,xf_fwd0_xb,xf_fwd1,xf_bwd0 --ldm 10 > $o/probe2.txt 2>&1; cat $o/probe2.txt
timeout 1500 python -m pytest tests/test_gpu_bf16_pipeline.py tests/test_gpu_bf16.py tests/test_gpu_layers.py tests/test_gpu_support.py tests/test_minibatch.py -x -q > $o/t4.txt 2>&1; tail -8 $o/t4.txt
F="--no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm"
timeout 600 python bench.py --operand bf16 $F > $o/bench_bf16_b.json 2> $o/bench_bf16_b.err
timeout 600 python bench.py $F > $o/bench_f32_b.json 2> $o/bench_f32_b.err
python - <<'PY'
import json
for f in ["bench_bf16_b","bench_f32_b"]:
    try:
        d=json.loads(open(f"gpurun_out/r6/{f}.json").read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 1200 python -m pytest tests/test_gpu_step_oracle.py -x -q -k "am_bf16 or am_gradients" > $o/t5.txt 2>&1; tail -8 $o/t5.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_f32 -o run -- python3 bench.py --steps 20 --warmup 3 $F > $o/bench_f32_prof.json 2> $o/bench_f32_prof.err
python3 tools/epoch_sequence.py $o/stats_f32 "k_xform_mfma_fwd<1, false, 10" > $o/f32_epoch_sequence.md 2>&1
rm -rf $o/stats_f32
head -45 $o/f32_epoch_sequence.md
