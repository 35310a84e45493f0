set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "1 1"; do
  set -- $cfg
  rm -rf gpurun_out/trace_tmp
  FSGM_EPI_STRIPS=$1 FSGM_EPI_LANES=$2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_tmp -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/trace_tmp.log 2>&1
  T=$(find gpurun_out/trace_tmp -name "*kernel_trace.csv" | head -1)
  python3 - "$T" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void fsgm::","")[:34], r.get("Stream_Id", "?")) for r in rows)
ck = [i for i, e in enumerate(ev) if "pair_ckpt" in e[2]]
i0 = ck[2]
lo = ev[i0][0]
for e in ev[max(0, i0 - 3): i0 + 12]:
    print("  %-36s stream %s start %9.1f end %9.1f dur %8.1f" % (e[2], e[3], (e[0] - lo) / 1e3, (e[1] - lo) / 1e3, (e[1] - e[0]) / 1e3))
PY
done
rm -rf gpurun_out/trace_tmp
