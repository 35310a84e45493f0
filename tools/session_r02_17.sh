set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 - <<'PY'
import os, subprocess, json
def run(env, frames=32):
    e = dict(os.environ); e.update(env); e["FSGM_SWEEP_GPW"] = "1"; e["FSGM_BENCH_NOCHECK"] = "1"
    out = subprocess.run(["python3", "bench.py", "--no-cpu-baseline", "--frames-per-gpu", str(frames), "--steps", "15"], env=e, capture_output=True, text=True, timeout=300)
    try:
        d = json.loads(out.stdout.strip().split("\n")[-1])
        print(env, "ms_per_step %.3f stage %.3f" % (d["ms_per_step"], d["roofline"]["stage_ms"]), flush=True)
    except Exception as ex:
        print(env, "FAILED", out.stderr[-300:], flush=True)
for skip in (0, 1, 2, 4, 6, 3, 5, 0):
    run({"FSGM_DBG_SKIP": str(skip)})
for skip in (0, 1, 6):
    run({"FSGM_DBG_SKIP": str(skip), "FSGM_EPI_LANES": "1"})
PY
