cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp MRGCN_MIX_TILE=0
bash tools/pmc_passes.sh gpurun_out/pmc_fwd mem -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-renumbered-extra --no-literal-spmm --no-graph
python3 tools/pmc_summary.py gpurun_out mix_fwd > gpurun_out/pmc_fwd_summary.md
python3 tools/pmc_summary.py gpurun_out mix_bwd > gpurun_out/pmc_bwd_summary.md
cat gpurun_out/pmc_fwd_summary.md
rm -rf gpurun_out/pmc_fwd_*/
