#!/bin/bash
# Copies what tools/regen_profiles.sh left under gpurun_out/<tag>/ into profiles/<tag>_* (the tracked artefacts).
#   tools/collect_profiles.sh <tag of the regen run> [<tag under profiles/>, default the same]
src=${1:-rXX}
tag=${2:-$src}
o=gpurun_out/$src
for f in bench_line.json bench_fb15k.json bench_under_rocprof.json epoch_kernel_stats.md \
         epoch_kernel_trace_medians.md epoch_sequence.md epoch_pmc.md kernel_roofline.md lp_epoch_sequence.md spmm_pmc.md \
         spmm_pmc_fb15k.md next_rows.json minibatch_step_sequence.md halo.json gemm_probe.txt gemm_probe.json rocm_smi_during_bench.txt am_encoders_step.md; do
  [ -s $o/$f ] && cp $o/$f profiles/${tag}_$f
done
cp $o/spmm_pmc_latest.json profiles/spmm_pmc_latest.json
cp $o/spmm_pmc_fb15k.json profiles/spmm_pmc_fb15k.json
{
  echo "# MFMA counters of the tiled product (k_mm_tile) over the TCNN-M products (tools/gemm_probe.py, 2 048 literals of 37 x 300), rocprofv3 --pmc passes (tools/pmc_passes.sh <out> mfma)"
  echo
  echo "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs); v_mfma_f32_16x16x4_f32 peaks at 157.3 TFLOP/s.  Per-product times and TFLOP/s of the same build: ${tag}_gemm_probe.txt"
  echo
  cat $o/mfma_mm.md
} > profiles/${tag}_mfma_pmc.md
ls -la profiles/${tag}_* profiles/spmm_pmc_latest.json
