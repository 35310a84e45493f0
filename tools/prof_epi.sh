# rocprofv3 kernel stats of the headline bench + the plain bench line (run on the GPU box through gpurun)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
grep "^{" gpurun_out/bench_default.json | cut -c1-250
rm -rf gpurun_out/prof_epi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_epi -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_epi.log 2>&1
S=$(find gpurun_out/prof_epi -name "*kernel_stats.csv" | head -1)
cp $S gpurun_out/prof_epi_kernel_stats.csv
cut -c1-160 gpurun_out/prof_epi_kernel_stats.csv | head -8
grep "^{" gpurun_out/prof_epi.log | cut -c1-200
rm -rf gpurun_out/prof_epi
