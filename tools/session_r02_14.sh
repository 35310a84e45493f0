set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export FSGM_EPI_STRIPS=1 FSGM_EPI_LANES=1 FSGM_STRIP_DEEP=0
timeout -k 10 500 bash tools/sq_counters.sh > gpurun_out/r02_sq14.log 2>&1 || { tail -20 gpurun_out/r02_sq14.log; exit 1; }
cp gpurun_out/sq_counters.md gpurun_out/sq_counters_strips.md
grep -E "strip" gpurun_out/sq_counters.md | cut -c1-330
