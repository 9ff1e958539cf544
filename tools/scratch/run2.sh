cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_extras.py -m gpu -x -q 2>&1 | tail -3
for cfg in "1 2" "0 2"; do
  set -- $cfg
  export MRGCN_MIX_MFMA=$1 MRGCN_MIX_TN=$2 MRGCN_MIX_TILE=0
  d=gpurun_out/prof_m$1_n$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o run -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-renumbered-extra --no-literal-spmm > gpurun_out/bench_m$1_n$2.log 2>&1
  echo "== mfma=$1 tn=$2"
  python3 - <<PY
import csv,glob
f=glob.glob("$d/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:4]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
done
