#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/r6; mkdir -p $o
hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/lab/copy_lab.hip -o /tmp/copy_lab && /tmp/copy_lab > $o/copy_lab.txt 2>&1; cat $o/copy_lab.txt
timeout 600 python tools/lab/spmm_hot_lab.py > $o/spmm_hot_lab.txt 2>&1; tail -12 $o/spmm_hot_lab.txt
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; tail -3 $o/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "seeds", d["extra"].get("seeds_ms_per_step"))
print("roofline", {k: d["roofline"].get(k) for k in ("frac","avg_ms","in_epoch_frac","in_epoch","in_epoch_error")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:200])
w=d["extra"].get("workloads",{})
for k,v in w.items(): print(k, {kk: v.get(kk) for kk in ("ms_per_step","spmm_frac","error")})
print({k: d["extra"].get(k) for k in ("device_copy_gbps","device_copy_gbps_hip","triad_gbps","spmm_transposed_gbps","spmm_literal_gbps")})
PY
