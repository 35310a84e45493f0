set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 - <<'PY'
import os, subprocess, json
def run(env, frames=32):
    e = dict(os.environ); e.update(env); e["FSGM_SWEEP_GPW"] = "1"
    out = subprocess.run(["python3", "bench.py", "--no-cpu-baseline", "--frames-per-gpu", str(frames), "--steps", "15"], env=e, capture_output=True, text=True, timeout=300)
    try:
        d = json.loads(out.stdout.strip().split("\n")[-1])
        print(env.get("FSGM_STRIP_NOWAIT"), "ms_per_step %.3f stage %.3f" % (d["ms_per_step"], d["roofline"]["stage_ms"]), flush=True)
    except Exception as ex:
        print(env, "FAILED", out.stderr[-300:], flush=True)
base = {"FSGM_EPI_STRIPS": "1", "FSGM_EPI_LANES": "1", "FSGM_STRIP_DEEP": "0", "FSGM_BENCH_NOCHECK": "1"}
run({"FSGM_EPI_STRIPS": "0"})
for nw in (3, 3 + 16, 3 + 32, 3 + 64, 3 + 128, 3 + 16 + 128, 3 + 32 + 64, 3 + 16 + 32 + 64 + 128):
    run(dict(base, FSGM_STRIP_NOWAIT=str(nw)))
PY
