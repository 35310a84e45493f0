# usage (through gpurun): bash tools/prof_script.sh scratch_dbg/foo.py  -> rocprofv3 kernel stats of a python script
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -- python3 $1 > gpurun_out/prof_tmp.log 2>&1
tail -2 gpurun_out/prof_tmp.log
S=$(find gpurun_out/prof_tmp -name "*kernel_stats.csv" | head -1)
cut -c1-150 $S | head -10
rm -rf gpurun_out/prof_tmp
