#!/bin/bash
# Regenerates every measurement artefact under profiles/ from the tree it runs in (on a GPU box):
#   tools/regen_profiles.sh <round-tag, e.g. r04>      -> gpurun_out/<tag>/...   (copy into profiles/ afterwards)
# Passes: (1) SpMM PMC for the two headline shapes -> spmm_pmc_latest.json (AM, F = 10, k_spmm3) and
# spmm_pmc_fb15k.json (FB15k-237, F = 200, k_spmm<64>): the bench lines quote them (digest-checked),
# (2) the default bench line (headline + seeds 0-2 + the six side workloads under extra.workloads),
# (3) rocprofv3 --kernel-trace --stats of the same command -> kernel stats, trace medians, the epoch's launch sequence,
# (4) epoch PMC (memory + SQ sets) for the weight_I streamers and the transforms -> epoch_pmc.md and, with (3),
# kernel_roofline.md, (5) the fb15k line + its launch sequence, (6) the encoders' product probe + MFMA counters,
# (7) the next-rows probe (mini-batch incl. the masked pass, encoders, ingestion) + the launch sequence of one
# re-sampled mini-batch step, (8) the halo-size probe, (9) one replayed step of the full-multimodal model kernel by kernel.
tag=${1:-rXX}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
o=gpurun_out/$tag
mkdir -p $o
bash tools/pmc_passes.sh $o/pmc_spmm mem -- python3 tools/spmm_probe.py --ld 10 --row-bytes 40 44 --iters 10
python3 tools/make_spmm_pmc_json.py $o/pmc_spmm "k_spmm3<" > $o/spmm_pmc_latest.json
python3 tools/pmc_summary.py $o k_spmm3 > $o/spmm_pmc.md
rm -rf $o/pmc_spmm_*/
bash tools/pmc_passes.sh $o/pmc_spfb mem -- python3 tools/spmm_probe.py --workload fb15k --F 200 --ld 200 --row-bytes 800 --iters 10
python3 tools/make_spmm_pmc_json.py $o/pmc_spfb "k_spmm<64" fb15k 200 > $o/spmm_pmc_fb15k.json
python3 tools/pmc_summary.py $o "k_spmm<64" > $o/spmm_pmc_fb15k.md
rm -rf $o/pmc_spfb_*/
cp $o/spmm_pmc_latest.json profiles/spmm_pmc_latest.json   # the bench lines quote them (digest-checked)
cp $o/spmm_pmc_fb15k.json profiles/spmm_pmc_fb15k.json
# clocks / power beside the bench line (one sample per 0.5 s while it runs; needs nothing but read access)
( for i in $(seq 1 400); do rocm-smi --showclocks --showpower --csv 2>/dev/null | tr '\n' ' '; echo; sleep 0.5; done > $o/rocm_smi_during_bench.txt ) &
smi_pid=$!
python3 bench.py > $o/bench_line.json 2> $o/bench_line.err
kill $smi_pid 2>/dev/null; wait $smi_pid 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o run -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads > $o/bench_under_rocprof.json 2> $o/bench_under_rocprof.err
python3 tools/prof_summary.py $o/stats 40 > $o/epoch_kernel_stats.md
python3 tools/trace_summary.py $o/stats > $o/epoch_kernel_trace_medians.md 2>/dev/null
python3 tools/epoch_sequence.py $o/stats "k_xform_mfma_fwd<1, false, 10" > $o/epoch_sequence.md 2>/dev/null
bash tools/pmc_passes.sh $o/pmc_epoch all -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-literal-spmm --no-graph --no-seeds --no-side-workloads
for k in k_mix_fwd k_mix_bwd_sup k_dcomp k_adam_rows k_xform_mfma_fwd k_xform_cols_lds k_xform_mfma_dw "k_spmm<" k_spmm3; do
  echo "## $k"; python3 tools/pmc_summary.py $o "$k" | tail -n +3
done > $o/epoch_pmc.md
python3 tools/roofline_table.py $o/stats $o/bench_line.json $o > $o/kernel_roofline.md 2> $o/kernel_roofline.err
rm -rf $o/stats $o/pmc_epoch_*/
python3 bench.py --workload fb15k > $o/bench_fb15k.json 2> $o/bench_fb15k.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_lp -o run -- python3 bench.py --workload fb15k --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2> $o/bench_fb15k_prof.err
python3 tools/epoch_sequence.py $o/stats_lp k_corrupt_triples > $o/lp_epoch_sequence.md 2>&1
rm -rf $o/stats_lp
# (6) the encoders' tiled product over the TCNN-M shapes: per-product rates, MFMA counters; (7) the next-rows probe
python3 tools/gemm_probe.py > $o/gemm_probe.txt 2> $o/gemm_probe.err
python3 tools/gemm_probe.py --mm bf16 >> $o/gemm_probe.txt 2>> $o/gemm_probe.err
python3 tools/gemm_probe.py --json > $o/gemm_probe.json 2>> $o/gemm_probe.err
bash tools/pmc_passes.sh $o/pmc_mm mfma -- python3 tools/gemm_probe.py --iters 3
python3 tools/pmc_summary.py $o k_mm_tile > $o/mfma_mm.md
rm -rf $o/pmc_mm_*/
python3 tools/next_rows_probe.py > $o/next_rows.json 2> $o/next_rows.err
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_mb -o run -- python3 tools/minibatch_step.py 10 > $o/minibatch_step.txt 2> $o/minibatch_step.err
python3 tools/epoch_sequence.py $o/stats_mb k_sup_rowcount 2 > $o/minibatch_step_sequence.md 2>&1
rm -rf $o/stats_mb
python3 tools/halo_probe.py > $o/halo.json 2> $o/halo.err
rocprofv3 --kernel-trace --output-format csv -d $o/stats_enc -o run -- python3 tools/am_encoders_step.py run > $o/am_encoders_step.txt 2> $o/am_encoders_step.err
python3 tools/am_encoders_step.py summary $o/stats_enc > $o/am_encoders_step.md 2>> $o/am_encoders_step.err
rm -rf $o/stats_enc
rocprofv3 --kernel-trace --output-format csv -d $o/stats_encb -o run -- python3 tools/am_encoders_step.py run bf16 > $o/am_encoders_step_bf16.txt 2> $o/am_encoders_step_bf16.err
python3 tools/am_encoders_step.py summary $o/stats_encb > $o/am_encoders_step_bf16.md 2>> $o/am_encoders_step_bf16.err
rm -rf $o/stats_encb
# (10) the bf16 pipeline's epoch in launch order; (11) the yardstick and LDS labs (tools/lab)
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_bf16 -o run -- python3 bench.py --operand bf16 --steps 20 --warmup 3 --no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm > $o/bench_bf16_under_rocprof.json 2> $o/bench_bf16_under_rocprof.err
python3 tools/epoch_sequence.py $o/stats_bf16 k_xform_bf16_fwd > $o/bf16_epoch_sequence.md 2>&1
rm -rf $o/stats_bf16
# (10b) the small graphs' replayed epochs in launch order (BASELINE configs 1 and 2: launch-bound)
for w in aifb mutag; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats_$w -o run -- python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm > $o/bench_${w}_under_rocprof.json 2> $o/bench_${w}_under_rocprof.err
  python3 tools/epoch_sequence.py $o/stats_$w k_xent_rows shortest > $o/${w}_epoch_sequence.md 2>&1
  rm -rf $o/stats_$w
done
hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/lab/copy_lab.hip -o /tmp/copy_lab 2>/dev/null && /tmp/copy_lab > $o/copy_lab.txt 2>&1
python3 tools/lab/spmm_hot_lab.py > $o/spmm_hot_lab.txt 2>&1
# the CPU suite last: the tree these artefacts describe is green
python3 -m pytest tests -q -x -m "not gpu" > $o/cpu_suite.txt 2>&1; tail -2 $o/cpu_suite.txt
ls -la $o
# (12) the driver's multi-rank command with two ranks on this one GPU (gloo: the ranks share the device, so the times
# are time-shared upper bounds): both partitioned engines priced, logits against the single-GPU model
timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-renumbered-extra --no-reference-loop --no-seeds --no-side-workloads --no-literal-spmm > $o/bench_two_ranks_one_gpu.json 2> $o/bench_two_ranks_one_gpu.err
ls -la $o
