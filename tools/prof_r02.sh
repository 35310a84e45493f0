# Round-2 evidence run (through gpurun): full GPU test suite, headline bench, rocprof kernel stats, PMC traffic,
# SQ counters, secondary workloads, micro-benchmarks.  Everything lands in gpurun_out/; copy to profiles/ afterwards.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_final.log 2>&1 || { tail -40 gpurun_out/r02_pytest_final.log; exit 1; }
tail -2 gpurun_out/r02_pytest_final.log
timeout -k 10 300 python3 bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err || { tail -20 gpurun_out/r02_bench_default.err; exit 1; }
cut -c1-400 gpurun_out/r02_bench_default.json
rm -rf gpurun_out/prof_epi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_epi -- python3 bench.py --no-cpu-baseline > gpurun_out/r02_prof_epi.log 2>&1
cp $(find gpurun_out/prof_epi -name "*kernel_stats.csv" | head -1) gpurun_out/r02_bench_default_kernel_stats.csv
rm -rf gpurun_out/prof_epi
cut -c1-140 gpurun_out/r02_bench_default_kernel_stats.csv | head -9
timeout -k 10 900 bash tools/pmc_traffic.sh > gpurun_out/r02_pmc_traffic.log 2>&1 || { tail -20 gpurun_out/r02_pmc_traffic.log; exit 1; }
tail -2 gpurun_out/r02_pmc_traffic.log
timeout -k 10 500 bash tools/sq_counters.sh > gpurun_out/r02_sq_final.log 2>&1 || { tail -20 gpurun_out/r02_sq_final.log; exit 1; }
for wl in "--paths 4" "--workload pyramid3" "--workload pyramid3_ng" "--workload postprocess"; do
  n=$(echo $wl | tr -d ' -' )
  timeout -k 10 300 python3 bench.py $wl --no-cpu-baseline > gpurun_out/r02_bench_$n.json 2> gpurun_out/r02_bench_$n.err || { tail -20 gpurun_out/r02_bench_$n.err; exit 1; }
  cut -c1-300 gpurun_out/r02_bench_$n.json
done
timeout -k 10 120 tools/ubench/pk_rates > gpurun_out/r02_ubench_pk_rates.txt 2>&1 || true
timeout -k 10 120 tools/ubench/copy_rates > gpurun_out/r02_ubench_copy_rates.txt 2>&1 || true
timeout -k 10 120 tools/ubench/h2d_rates > gpurun_out/r02_ubench_h2d_rates.txt 2>&1 || true
timeout -k 10 120 tools/ubench/pattern_rates > gpurun_out/r02_ubench_pattern_rates.txt 2>&1 || true
