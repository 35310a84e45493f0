set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest12.log 2>&1 || { tail -60 gpurun_out/r02_pytest12.log; exit 1; }
tail -2 gpurun_out/r02_pytest12.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/r02_bench12.json 2> gpurun_out/r02_bench12.err || { tail -20 gpurun_out/r02_bench12.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r02_bench12.json').read().strip().split('\n')[-1])
print('ms_per_step %.3f frac %.4f checked %s whole %.4f' % (d['ms_per_step'], d['roofline']['frac'], d['checked'], d['whole_mex']['ms_per_frame']))
print('host_call_ms', d['host_call_ms'])"
timeout -k 10 300 python3 bench.py --workload pyramid3_ng > gpurun_out/r02_bench12_ng.json 2> gpurun_out/r02_bench12_ng.err || { tail -20 gpurun_out/r02_bench12_ng.err; exit 1; }
cat gpurun_out/r02_bench12_ng.json | cut -c1-1500
timeout -k 10 300 python3 bench.py --no-cpu-baseline --total-frames 8 > gpurun_out/r02_bench12_tf8.json 2> gpurun_out/r02_bench12_tf8.err || { tail -20 gpurun_out/r02_bench12_tf8.err; exit 1; }
cut -c1-700 gpurun_out/r02_bench12_tf8.json
