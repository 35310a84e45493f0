set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_epi.py tests/test_gpu_fuzz.py tests/test_gpu_mex_gateways.py -m gpu -x -q > gpurun_out/r02_pytest2.log 2>&1 || { tail -40 gpurun_out/r02_pytest2.log; exit 1; }
tail -3 gpurun_out/r02_pytest2.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/r02_bench2.json 2> gpurun_out/r02_bench2.err || { tail -20 gpurun_out/r02_bench2.err; exit 1; }
cut -c1-1300 gpurun_out/r02_bench2.json
timeout -k 10 400 bash tools/sq_counters.sh > gpurun_out/r02_sq2.log 2>&1 || { tail -20 gpurun_out/r02_sq2.log; exit 1; }
grep -E "sweep_kernel|pair_" gpurun_out/sq_counters.md | cut -c1-200
