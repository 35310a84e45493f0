# Round-3 evidence run (through gpurun): headline bench, rocprofv3 kernel stats of the same command, PMC traffic
# (8 and 4 paths), SQ counters with the calibration kernels, secondary workloads.  Everything lands in gpurun_out/;
# copy to profiles/ afterwards.  PARTS selects what runs (default: all).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PARTS="${PARTS:-bench stats traffic sq secondary crossover}"
for part in $PARTS; do
case $part in
bench)
  timeout -k 10 400 python3 bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err || { tail -20 gpurun_out/r03_bench_default.err; exit 1; }
  cut -c1-400 gpurun_out/r03_bench_default.json ;;
stats)
  rm -rf gpurun_out/prof_epi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_epi -- python3 bench.py --no-cpu-baseline > gpurun_out/r03_prof_epi.log 2>&1
  cp $(find gpurun_out/prof_epi -name "*kernel_stats.csv" | head -1) gpurun_out/r03_bench_default_kernel_stats.csv
  rm -rf gpurun_out/prof_epi
  cut -c1-160 gpurun_out/r03_bench_default_kernel_stats.csv | head -9 ;;
traffic)
  timeout -k 10 900 bash tools/pmc_traffic.sh > gpurun_out/r03_pmc_traffic.log 2>&1 || { tail -20 gpurun_out/r03_pmc_traffic.log; exit 1; }
  tail -2 gpurun_out/r03_pmc_traffic.log
  BENCH_ARGS="--paths 4" TAG=_paths4 timeout -k 10 900 bash tools/pmc_traffic.sh > gpurun_out/r03_pmc_traffic_paths4.log 2>&1 || { tail -20 gpurun_out/r03_pmc_traffic_paths4.log; exit 1; }
  tail -2 gpurun_out/r03_pmc_traffic_paths4.log
  BENCH_ARGS="--frames-per-gpu 40" TAG=_b40 timeout -k 10 900 bash tools/pmc_traffic.sh > gpurun_out/r03_pmc_traffic_b40.log 2>&1 || { tail -20 gpurun_out/r03_pmc_traffic_b40.log; exit 1; }
  tail -2 gpurun_out/r03_pmc_traffic_b40.log ;;
sq)
  LEGS="${LEGS:---steps%2%--warmup%1%--no-cpu-baseline default40 paths4 batch8 pyramid3 pyramid3_ng}" timeout -k 10 1000 bash tools/sq_counters.sh > gpurun_out/r03_sq_final.log 2>&1 || { tail -20 gpurun_out/r03_sq_final.log; exit 1; }
  grep -c "^|" gpurun_out/sq_counters.md ;;
secondary)
  for wl in "--paths 4" "--frames-per-gpu 40" "--frames-per-gpu 8" "--frames-per-gpu 1" "--workload pyramid3" "--workload pyramid3_ng" "--workload postprocess"; do
    n=$(echo $wl | tr -d ' -' )
    timeout -k 10 300 python3 bench.py $wl --no-cpu-baseline > gpurun_out/r03_bench_$n.json 2> gpurun_out/r03_bench_$n.err || { tail -20 gpurun_out/r03_bench_$n.err; exit 1; }
    cut -c1-300 gpurun_out/r03_bench_$n.json
  done ;;
crossover)
  timeout -k 10 600 python3 tools/crossover.py 1,2,4,8,16,40 > gpurun_out/r03_crossover.txt 2>&1 || tail -5 gpurun_out/r03_crossover.txt
  cat gpurun_out/r03_crossover.txt ;;
esac
done
