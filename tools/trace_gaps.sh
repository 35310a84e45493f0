# kernel timeline of one bench step: per queue, busy time vs gaps (run on the GPU box through gpurun)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/trace_tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_tmp -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 2 > gpurun_out/trace_tmp.log 2>&1
T=$(find gpurun_out/trace_tmp -name "*kernel_trace.csv" | head -1)
python3 - "$T" <<'PY' | tee gpurun_out/trace_gaps.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
ev.sort()
# one step = from one hpair_ckpt start to the next one
ck = [e[0] for e in ev if "pair_ckpt" in e[2]]
lo, hi = ck[1], ck[2]
step = [e for e in ev if e[0] >= lo and e[0] < hi]
hi = max(e[1] for e in step)
print("step span us:", (hi - lo) / 1e3, "kernels:", len(step))
byq = collections.defaultdict(list)
for e in step: byq[(e[3], e[4])].append(e)
for q, es in sorted(byq.items()):
    busy = sum(e[1] - e[0] for e in es)
    gaps = [es[i + 1][0] - es[i][1] for i in range(len(es) - 1)]
    names = collections.Counter(e[2] for e in es)
    print("queue/stream", q, "n", len(es), "busy us %.1f" % (busy / 1e3), "first start %.1f" % ((es[0][0] - lo) / 1e3), "last end %.1f" % ((es[-1][1] - lo) / 1e3),
          "gap avg us %.2f max %.2f" % ((sum(gaps) / max(len(gaps), 1)) / 1e3, (max(gaps) if gaps else 0) / 1e3), dict(names))
# union busy of all kernels
iv = sorted((e[0], e[1]) for e in step)
tot = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: tot += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
tot += ce - cs
print("union busy us %.1f of span %.1f" % (tot / 1e3, (hi - lo) / 1e3))
PY
rm -rf gpurun_out/trace_tmp
