#!/bin/bash
# first GPU contact of round 6: bf16 pipeline tests, kernel A/B, bf16 and f32 epoch
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_bf16_pipeline.py tests/test_gpu_bf16.py -x -q > $o/t1.txt 2>&1; tail -15 $o/t1.txt
timeout 300 python tools/kernel_probe.py --which xf_fwd0,cast_x,xf_fwd0_xb,xf_fwd0_xb_f32,mix_fwd_add,mix_fwd_add_b,mix_fwd_addb,xf_fwd0,xf_fwd0_xb --ldm 10 > $o/probe1.txt 2>&1; cat $o/probe1.txt
F="--no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm"
timeout 600 python bench.py --operand bf16 $F > $o/bench_bf16.json 2> $o/bench_bf16.err; python - <<'PY'
import json
for f in ["gpurun_out/r6/bench_bf16.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
MRGCN_BF16_PIPELINE=0 timeout 600 python bench.py --operand bf16 $F > $o/bench_bf16_off.json 2> $o/bench_bf16_off.err
timeout 600 python bench.py $F > $o/bench_f32.json 2> $o/bench_f32.err
python - <<'PY'
import json
for f in ["bench_bf16","bench_bf16_off","bench_f32"]:
    try:
        d=json.loads(open(f"gpurun_out/r6/{f}.json").read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
