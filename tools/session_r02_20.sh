set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_epi.py -m gpu -x -q -k "strip or sweep or batch" > gpurun_out/r02_pytest20.log 2>&1 || { tail -60 gpurun_out/r02_pytest20.log; exit 1; }
tail -2 gpurun_out/r02_pytest20.log
python3 - <<'PY'
import os, subprocess, json
def run(env, frames=40):
    e = dict(os.environ); e.update(env)
    out = subprocess.run(["python3", "bench.py", "--no-cpu-baseline", "--frames-per-gpu", str(frames), "--steps", "15"], env=e, capture_output=True, text=True, timeout=300)
    try:
        d = json.loads(out.stdout.strip().split("\n")[-1])
        print(env, frames, "ms_per_step %.3f stage %.3f frac %.4f checked %s" % (d["ms_per_step"], d["roofline"]["stage_ms"], d["roofline"]["frac"], d.get("checked")), flush=True)
    except Exception as ex:
        print(env, "FAILED", out.stderr[-300:], flush=True)
for rep in range(2):
    run({})
    run({"FSGM_EPI_STRIPS": "1", "FSGM_EPI_LANES": "1"})
    run({"FSGM_EPI_STRIPS": "1", "FSGM_EPI_LANES": "2"})
run({"FSGM_EPI_STRIPS": "1", "FSGM_EPI_LANES": "1"}, 32)
run({"FSGM_EPI_STRIPS": "1", "FSGM_EPI_LANES": "2"}, 32)
run({"FSGM_EPI_STRIPS": "1", "FSGM_EPI_LANES": "3"}, 48)
PY
