#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/r6; mkdir -p $o
timeout 1500 python -m pytest tests/test_encoders.py tests/test_gpu_configs.py -x -q -m gpu > $o/t7.txt 2>&1; tail -8 $o/t7.txt
timeout 900 python - > $o/enc_twin.txt 2>&1 <<'PY'
import json, sys, torch
sys.argv = ["bench.py"]
import bench
args = bench.parse()
dev = torch.device("cuda:0")
for compute in ("f32", "bf16"):
    r = bench.am_encoders_record(args, dev, compute=compute)
    print(compute, json.dumps({k: r[k] for k in ("ms_per_step", "ms_per_step_eager", "ms_per_step_given_features", "launch", "final_loss", "ms_standin_backbones")}))
    import gc; gc.collect(); torch.cuda.empty_cache()
PY
tail -4 $o/enc_twin.txt
