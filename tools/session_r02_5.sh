set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "1 1" "1 2" "0 2"; do
  set -- $cfg
  rm -rf gpurun_out/trace_tmp
  FSGM_SWEEP_GPW=1 FSGM_EPI_STRIPS=$1 FSGM_EPI_LANES=$2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_tmp -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 2 > gpurun_out/trace_tmp.log 2>&1
  T=$(find gpurun_out/trace_tmp -name "*kernel_trace.csv" | head -1)
  echo "=== STRIPS $1 LANES $2"
  python3 - "$T" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void fsgm::","")[:34], r.get("Stream_Id", "?")) for r in rows)
ck = [e[0] for e in ev if "pair_ckpt" in e[2]]
lo, hi = ck[1], ck[2]
step = [e for e in ev if lo <= e[0] < hi]
end = max(e[1] for e in step)
print("step span us %.1f kernels %d" % ((end - lo) / 1e3, len(step)))
if len(step) < 30:
    for e in step: print("  %-36s stream %s start %8.1f dur %8.1f" % (e[2], e[3], (e[0] - lo) / 1e3, (e[1] - e[0]) / 1e3))
else:
    by = collections.defaultdict(list)
    for e in step: by[(e[3], e[2])].append(e)
    for k, es in sorted(by.items()): print("  stream %s %-36s n %3d first %8.1f last end %8.1f sum dur %8.1f" % (k[0], k[1], len(es), (es[0][0]-lo)/1e3, (es[-1][1]-lo)/1e3, sum(x[1]-x[0] for x in es)/1e3))
PY
done
rm -rf gpurun_out/trace_tmp
