#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
for v in 64 64; do MRGCN_WIDE_UNIT=$v timeout 600 python bench.py --workload fb15k --no-cpu-baseline > $o/fbu_$v.json 2>$o/fbu_$v.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r6/fbu_$v.json").read().strip().splitlines()[-1]); print("unit=$v", round(d["ms_per_step"],4))
PY
done
timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_lp.py -x -q -k "wide or lp" > $o/t13.txt 2>&1; tail -3 $o/t13.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_lp -o run -- python3 bench.py --workload fb15k --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2> $o/fb_prof.err
python3 tools/epoch_sequence.py $o/stats_lp k_corrupt_triples > $o/lp_epoch_sequence.md 2>&1
rm -rf $o/stats_lp; head -48 $o/lp_epoch_sequence.md | cut -c1-120
