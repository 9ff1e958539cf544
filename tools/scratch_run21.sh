#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/r6; mkdir -p $o
F="--no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm"
for w in aifb mutag; do
rocprofv3 --kernel-trace --stats --output-format csv -d $o/st_$w -o run -- python3 bench.py --workload $w --steps 50 --warmup 5 $F > $o/st_$w.json 2> $o/st_$w.err
python3 - <<PY
import csv, glob
f=sorted(glob.glob("gpurun_out/r6/st_$w/**/*kernel_trace.csv", recursive=True))[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# last 2 epochs: find repeating pattern by name of adam_multi
names=[r["Kernel_Name"] for r in rows]
idx=[i for i,n in enumerate(names) if "k_adam_multi" in n]
a,b=idx[-3]+1, idx[-2]+1
t0=int(rows[a]["Start_Timestamp"])
print("$w epoch launches", b-a, "span us", (int(rows[b]["Start_Timestamp"])-t0)/1e3)
for r in rows[a:b]:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:8.1f} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:7.1f}  {r["Kernel_Name"][:90]}')
PY
rm -rf $o/st_$w
done
