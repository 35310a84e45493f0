set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_ng
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ng -- python3 scratch_dbg/ngbatch.py 8 > gpurun_out/prof_ng.log 2>&1
S=$(find gpurun_out/prof_ng -name "*kernel_stats.csv" | head -1)
cut -c1-150 $S | head -14
rm -rf gpurun_out/prof_ng
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ng -- python3 scratch_dbg/ngbatch.py 1 > gpurun_out/prof_ng1.log 2>&1
S=$(find gpurun_out/prof_ng -name "*kernel_stats.csv" | head -1)
cut -c1-150 $S | head -14
rm -rf gpurun_out/prof_ng
