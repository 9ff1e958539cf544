#!/bin/bash
# Regenerates every measurement artefact under profiles/ from the tree it runs in (on a GPU box):
#   tools/regen_profiles.sh <round-tag, e.g. r02>      -> gpurun_out/<tag>/...   (copy into profiles/ afterwards)
# Passes: (1) SpMM PMC -> spmm_pmc_latest.json, (2) bench line, (3) rocprofv3 --kernel-trace --stats of the same command,
# (4) epoch PMC (memory + SQ sets) for the weight_I streamers and the transforms, (5) seeds 0,1,2, fb15k, ref_int8,
# (6) the encoders' product probe + MFMA counters, (7) the next-rows probe (mini-batch, encoders, ingestion).
tag=${1:-rXX}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/$tag
mkdir -p $o
bash tools/pmc_passes.sh $o/pmc_spmm mem -- python3 tools/spmm_probe.py --ld 10 --row-bytes 40 44 --iters 10
python3 tools/make_spmm_pmc_json.py $o/pmc_spmm "k_spmm3<" > $o/spmm_pmc_latest.json
python3 tools/pmc_summary.py $o k_spmm3 > $o/spmm_pmc.md
rm -rf $o/pmc_spmm_*/
cp $o/spmm_pmc_latest.json profiles/spmm_pmc_latest.json   # the bench line quotes it (digest-checked)
# clocks / power beside the bench line (one sample per 0.5 s while it runs; needs nothing but read access)
( for i in $(seq 1 200); do rocm-smi --showclocks --showpower --csv 2>/dev/null | tr '\n' ' '; echo; sleep 0.5; done > $o/rocm_smi_during_bench.txt ) &
smi_pid=$!
python3 bench.py --steps 20 --warmup 3 > $o/bench_line.json 2> $o/bench_line.err
kill $smi_pid 2>/dev/null; wait $smi_pid 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o run -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-renumbered-extra --no-reference-loop > $o/bench_under_rocprof.json 2> $o/bench_under_rocprof.err
python3 tools/prof_summary.py $o/stats 40 > $o/epoch_kernel_stats.md
python3 tools/trace_summary.py $o/stats > $o/epoch_kernel_trace_medians.md 2>/dev/null
python3 tools/epoch_sequence.py $o/stats > $o/epoch_sequence.md 2>/dev/null
rm -rf $o/stats
bash tools/pmc_passes.sh $o/pmc_epoch all -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-literal-spmm --no-graph
for k in k_mix_fwd k_mix_bwd k_adam_rows k_xform_mfma_fwd k_xform_mfma_dw k_spmm_t_live; do
  echo "## $k"; python3 tools/pmc_summary.py $o $k | tail -n +3
done > $o/epoch_pmc.md
rm -rf $o/pmc_epoch_*/
python3 tools/seed_median.py > $o/seeds.json 2> $o/seeds.err
python3 bench.py --workload fb15k > $o/bench_fb15k.json 2> $o/bench_fb15k.err
python3 bench.py --value-mode ref_int8 --no-cpu-baseline > $o/bench_ref_int8.json 2> $o/bench_ref_int8.err
for w in aifb mutag synth10m; do python3 bench.py --workload $w --no-cpu-baseline > $o/bench_$w.json 2> $o/bench_$w.err; done
# (6) the encoders' tiled product over the TCNN-M shapes: per-product rates, MFMA counters; (7) the next-rows probe
python3 tools/gemm_probe.py > $o/gemm_probe.txt 2> $o/gemm_probe.err
python3 tools/gemm_probe.py --json > $o/gemm_probe.json 2>> $o/gemm_probe.err
bash tools/pmc_passes.sh $o/pmc_mm mfma -- python3 tools/gemm_probe.py --iters 3
python3 tools/pmc_summary.py $o k_mm_tile > $o/mfma_mm.md
rm -rf $o/pmc_mm_*/
python3 tools/next_rows_probe.py > $o/next_rows.json 2> $o/next_rows.err
ls -la $o
