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
        print(env, frames, "ms_per_step %.3f stage %.3f frac %.4f checked %s" % (d["ms_per_step"], d["roofline"]["stage_ms"], d["roofline"]["frac"], d.get("checked")), flush=True)
    except Exception as ex:
        print(env, "FAILED", out.stderr[-300:], flush=True)
run({"FSGM_PAIR_STREAM_PRIO": "0"})
run({"FSGM_PAIR_STREAM_PRIO": "1"})
S = {"FSGM_EPI_STRIPS": "1", "FSGM_EPI_LANES": "1"}
run(dict(S, FSGM_PAIR_STREAM_PRIO="0"))
run(dict(S, FSGM_PAIR_STREAM_PRIO="1"))
for pad in (16000, 24000, 36000, 60000):
    run(dict(S, FSGM_PAIR_STREAM_PRIO="1", FSGM_STRIP_LDS_PAD=str(pad)))
run(dict(S, FSGM_PAIR_STREAM_PRIO="0", FSGM_STRIP_LDS_PAD="36000"))
run(dict(S, FSGM_PAIR_STREAM_PRIO="1", FSGM_STRIP_LDS_PAD="36000", FSGM_EPI_LANES="2"))
PY
