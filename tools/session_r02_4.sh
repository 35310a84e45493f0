set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_epi.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r02_pytest4.log 2>&1 || { tail -60 gpurun_out/r02_pytest4.log; exit 1; }
tail -2 gpurun_out/r02_pytest4.log
for cfg in "1 2 32" "0 2 32" "1 1 32" "1 3 32" "1 2 48" "1 2 64" "1 1 64"; do
  set -- $cfg
  FSGM_SWEEP_GPW=1 FSGM_EPI_STRIPS=$1 FSGM_EPI_LANES=$2 timeout -k 10 300 python3 bench.py --no-cpu-baseline --frames-per-gpu $3 > gpurun_out/r02_bench4_$1_$2_$3.json 2> gpurun_out/r02_bench4.err || { tail -20 gpurun_out/r02_bench4.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r02_bench4_$1_$2_$3.json').read().strip().split('\n')[-1])
print('STRIPS $1 LANES $2 FRAMES $3: ms_per_step %.3f stage_ms %.3f frac %.4f checked %s whole_mex %.4f' % (d['ms_per_step'], d['roofline']['stage_ms'], d['roofline']['frac'], d['checked'], d['whole_mex']['ms_per_frame']))"
done
