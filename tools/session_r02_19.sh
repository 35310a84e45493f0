set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 - <<'PY'
import os, subprocess, json
def run(env, frames=40):
    e = dict(os.environ); e.update(env)
    out = subprocess.run(["python3", "bench.py", "--no-cpu-baseline", "--frames-per-gpu", str(frames), "--steps", "15"], env=e, capture_output=True, text=True, timeout=300)
    try:
        d = json.loads(out.stdout.strip().split("\n")[-1])
        print(env.get("FSGM_LIB_PATH", "base")[-16:], frames, "ms_per_step %.3f stage %.3f frac %.4f checked %s copy %.0f" % (d["ms_per_step"], d["roofline"]["stage_ms"], d["roofline"]["frac"], d.get("checked"), d["roofline"]["copy_GBps_measured"]), flush=True)
    except Exception as ex:
        print(env, "FAILED", out.stderr[-300:], flush=True)
root = os.environ["GRAFT_REPO_ROOT"]
for rep in range(2):
    run({})
    for v in ("occ4", "pf2", "l8"):
        run({"FSGM_LIB_PATH": f"{root}/ab/lib_{v}.so"})
PY
