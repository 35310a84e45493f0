set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 120 tools/ubench/h2d_rates > gpurun_out/h2d_rates.log 2>&1 || true
cat gpurun_out/h2d_rates.log
python3 - <<'PY'
import os, subprocess, json
def run(env, frames=32):
    e = dict(os.environ); e.update(env); e["FSGM_SWEEP_GPW"] = "1"
    out = subprocess.run(["python3", "bench.py", "--no-cpu-baseline", "--frames-per-gpu", str(frames), "--steps", "15"], env=e, capture_output=True, text=True, timeout=300)
    try:
        d = json.loads(out.stdout.strip().split("\n")[-1])
        print(env, frames, "ms_per_step %.3f stage %.3f frac %.4f checked %s whole %.4f" % (d["ms_per_step"], d["roofline"]["stage_ms"], d["roofline"]["frac"], d.get("checked"), d["whole_mex"]["ms_per_frame"]), flush=True)
    except Exception as ex:
        print(env, "FAILED", out.stderr[-300:], flush=True)
root = os.environ["GRAFT_REPO_ROOT"]
for rep in range(3):
    run({"FSGM_LIB_PATH": root + "/ab/libv2.so"})
    run({})
    run({"FSGM_EPI_PAIRSPLIT": "0"})
for fr in (40, 48, 64):
    run({"FSGM_LIB_PATH": root + "/ab/libv2.so"}, fr)
    run({}, fr)
PY
