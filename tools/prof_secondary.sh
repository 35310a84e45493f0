set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --workload postprocess > gpurun_out/post_bench.json 2>&1; cat gpurun_out/post_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pyr -- python3 bench.py --workload pyramid3 --steps 8 > gpurun_out/prof_pyr.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_post -- python3 bench.py --workload postprocess --steps 8 > gpurun_out/prof_post.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pyrng -- python3 bench.py --workload pyramid3_ng --steps 8 > gpurun_out/prof_pyrng.log 2>&1
find gpurun_out/prof_pyr gpurun_out/prof_post gpurun_out/prof_pyrng -name "*kernel_stats.csv" | head
