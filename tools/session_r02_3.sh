set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_epi.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r02_pytest3.log 2>&1 || { tail -40 gpurun_out/r02_pytest3.log; exit 1; }
tail -2 gpurun_out/r02_pytest3.log
for cfg in "1 2" "2 2" "2 1" "2 3" "2 2"; do
  set -- $cfg
  FSGM_SWEEP_GPW=$1 FSGM_EPI_LANES=$2 timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/r02_bench3_$1_$2.json 2> gpurun_out/r02_bench3.err || { tail -20 gpurun_out/r02_bench3.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r02_bench3_$1_$2.json').read().strip().split('\n')[-1])
print('GPW $1 LANES $2: ms_per_step %.3f stage_ms %.3f frac %.4f checked %s whole_mex %.4f' % (d['ms_per_step'], d['roofline']['stage_ms'], d['roofline']['frac'], d['checked'], d['whole_mex']['ms_per_frame']))"
done
