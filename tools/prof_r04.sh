# Round-4 evidence run (through gpurun): headline bench, rocprofv3 kernel stats of the same command, PMC traffic, cost-stage
# counters, secondary workloads, crossover tables.  Everything lands in gpurun_out/; copy to profiles/ afterwards.
# PARTS selects what runs (default: all).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PARTS="${PARTS:-bench stats traffic cost secondary crossover ng inprocess}"
for part in $PARTS; do
case $part in
bench)
  timeout -k 10 500 python3 bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err || { tail -20 gpurun_out/r04_bench_default.err; exit 1; }
  cut -c1-400 gpurun_out/r04_bench_default.json ;;
stats)
  rm -rf gpurun_out/prof_epi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_epi -- python3 bench.py --no-cpu-baseline > gpurun_out/r04_prof_epi.log 2>&1
  cp $(find gpurun_out/prof_epi -name "*kernel_stats.csv" | head -1) gpurun_out/r04_bench_default_kernel_stats.csv
  rm -rf gpurun_out/prof_epi
  cut -c1-160 gpurun_out/r04_bench_default_kernel_stats.csv | head -9 ;;
traffic)
  timeout -k 10 900 bash tools/pmc_traffic.sh > gpurun_out/r04_pmc_traffic.log 2>&1 || { tail -20 gpurun_out/r04_pmc_traffic.log; exit 1; }
  tail -2 gpurun_out/r04_pmc_traffic.log
  BENCH_ARGS="--paths 4" TAG=_paths4 timeout -k 10 900 bash tools/pmc_traffic.sh > gpurun_out/r04_pmc_traffic_paths4.log 2>&1 || { tail -20 gpurun_out/r04_pmc_traffic_paths4.log; exit 1; }
  tail -2 gpurun_out/r04_pmc_traffic_paths4.log ;;
cost)
  B=64 bash tools/prof_cost.sh > gpurun_out/r04_prof_cost.log 2>&1 || { tail -20 gpurun_out/r04_prof_cost.log; exit 1; }
  cp gpurun_out/cost_stage.txt gpurun_out/r04_cost_stage.txt; cp gpurun_out/cost_kernel_stats.csv gpurun_out/r04_cost_kernel_stats.csv; cp gpurun_out/cost_counters.txt gpurun_out/r04_cost_counters.txt
  cat gpurun_out/r04_cost_stage.txt; grep -E "costbox|rawcost|box5" gpurun_out/r04_cost_counters.txt | cut -c1-260 ;;
secondary)
  for wl in "--paths 4" "--frames-per-gpu 40" "--frames-per-gpu 8" "--frames-per-gpu 1" "--workload pyramid3" "--workload pyramid3_ng" "--workload postprocess"; do
    n=$(echo $wl | tr -d ' -' )
    timeout -k 10 400 python3 bench.py $wl --no-cpu-baseline > gpurun_out/r04_bench_$n.json 2> gpurun_out/r04_bench_$n.err || { tail -20 gpurun_out/r04_bench_$n.err; exit 1; }
    cut -c1-300 gpurun_out/r04_bench_$n.json
  done ;;
crossover)
  timeout -k 10 700 python3 tools/crossover.py 1,2,4,8,16,40 > gpurun_out/r04_crossover.txt 2>&1 || tail -5 gpurun_out/r04_crossover.txt
  cat gpurun_out/r04_crossover.txt ;;
ng)
  for B in 1 8; do
    rm -rf gpurun_out/prof_ng
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ng -- python3 tools/ng_batch.py $B > gpurun_out/prof_ng.log 2>&1
    cp $(find gpurun_out/prof_ng -name "*kernel_stats.csv" | head -1) gpurun_out/r04_ng_batch${B}_kernel_stats.csv
    rm -rf gpurun_out/prof_ng
    tail -1 gpurun_out/prof_ng.log; cut -c1-150 gpurun_out/r04_ng_batch${B}_kernel_stats.csv | head -6
  done
  NGB=8 bash tools/ng_sq_counters.sh > gpurun_out/r04_ng_sq.log 2>&1 || tail -5 gpurun_out/r04_ng_sq.log
  cp gpurun_out/ng_counters.txt gpurun_out/r04_ng_counters.txt; grep -E "compact_kernel<16>|dedupe|cost_hint|wta" gpurun_out/r04_ng_counters.txt | cut -c1-260 ;;
inprocess)
  timeout -k 10 400 python3 bench.py --in-process 2 --frames-per-gpu 128 --no-cpu-baseline > gpurun_out/r04_bench_inprocess2.json 2> gpurun_out/r04_bench_inprocess2.err || { tail -20 gpurun_out/r04_bench_inprocess2.err; exit 1; }
  cut -c1-700 gpurun_out/r04_bench_inprocess2.json ;;
esac
done
