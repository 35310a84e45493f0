set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export FSGM_EPI_STRIPS=0 FSGM_SWEEP_GPW=1
timeout -k 10 500 bash tools/sq_counters.sh > gpurun_out/r02_sq10.log 2>&1 || { tail -20 gpurun_out/r02_sq10.log; exit 1; }
cp gpurun_out/sq_counters.md gpurun_out/sq_counters_blocks.md
grep -E "sweep_kernel|pair_|strip" gpurun_out/sq_counters.md | cut -c1-330
export FSGM_EPI_STRIPS=1 FSGM_EPI_LANES=1 FSGM_STRIP_DEEP=0
timeout -k 10 500 bash tools/sq_counters.sh > gpurun_out/r02_sq10b.log 2>&1 || { tail -20 gpurun_out/r02_sq10b.log; exit 1; }
cp gpurun_out/sq_counters.md gpurun_out/sq_counters_strips.md
grep -E "sweep_kernel|pair_|strip" gpurun_out/sq_counters.md | cut -c1-330
