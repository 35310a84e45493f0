set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest1.log 2>&1 || { tail -30 gpurun_out/r02_pytest1.log; exit 1; }
tail -3 gpurun_out/r02_pytest1.log
timeout -k 10 300 python3 bench.py > gpurun_out/r02_bench1.json 2> gpurun_out/r02_bench1.err || { tail -20 gpurun_out/r02_bench1.err; exit 1; }
cut -c1-1500 gpurun_out/r02_bench1.json
timeout -k 10 400 bash tools/sq_counters.sh > gpurun_out/r02_sq1.log 2>&1 || { tail -20 gpurun_out/r02_sq1.log; exit 1; }
timeout -k 10 400 bash tools/pmc_traffic.sh > gpurun_out/r02_pmc1.log 2>&1 || { tail -20 gpurun_out/r02_pmc1.log; exit 1; }
tail -3 gpurun_out/r02_pmc1.log
timeout -k 10 300 bash tools/trace_gaps.sh > gpurun_out/r02_gaps1.log 2>&1 || { tail -20 gpurun_out/r02_gaps1.log; exit 1; }
tail -8 gpurun_out/r02_gaps1.log
