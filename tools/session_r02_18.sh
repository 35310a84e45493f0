set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_epi.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r02_pytest18.log 2>&1 || { tail -60 gpurun_out/r02_pytest18.log; exit 1; }
tail -2 gpurun_out/r02_pytest18.log
python3 - <<'PY'
import os, subprocess, json
def run(env, frames=32):
    e = dict(os.environ); e.update(env); e["FSGM_SWEEP_GPW"] = "1"
    out = subprocess.run(["python3", "bench.py", "--no-cpu-baseline", "--frames-per-gpu", str(frames), "--steps", "15"], env=e, capture_output=True, text=True, timeout=300)
    try:
        d = json.loads(out.stdout.strip().split("\n")[-1])
        print(env, frames, "ms_per_step %.3f stage %.3f frac %.4f checked %s" % (d["ms_per_step"], d["roofline"]["stage_ms"], d["roofline"]["frac"], d.get("checked")), flush=True)
    except Exception as ex:
        print(env, "FAILED", out.stderr[-300:], flush=True)
root = os.environ["GRAFT_REPO_ROOT"]
for rep in range(3):
    run({"FSGM_LIB_PATH": root + "/ab/libv2.so"})
    run({})
for fr in (36, 38, 40, 44):
    run({}, fr)
PY
