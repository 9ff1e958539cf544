#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_layers.py tests/test_gpu_lp.py tests/test_gpu_support.py tests/test_gpu_configs.py tests/test_minibatch.py -x -q > $o/t17.txt 2>&1; tail -4 $o/t17.txt
timeout 900 python bench.py --no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-literal-spmm > $o/bench_q.json 2> $o/bench_q.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/bench_q.json").read().strip().splitlines()[-1])
print("value", d["value"], d["extra"]["seeds_ms_per_step"])
print({k: round(v["ms_per_step"],3) for k,v in d["extra"]["workloads"].items() if "ms_per_step" in v})
print({k: d["extra"].get(k) for k in ("device_copy_gbps_hip","triad_gbps","device_copy_gbps_hip_persistent","triad_gbps_persistent")})
PY
