# cost stage: timings, rocprofv3 kernel stats, SQ counters, HBM traffic of the fused kernel (run through gpurun)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B=${B:-64}
if [ -z "$COUNTERS_ONLY" ]; then
python3 tools/cost_stage.py $B | tee gpurun_out/cost_stage.txt
python3 tools/cost_stage.py 1 | tee -a gpurun_out/cost_stage.txt
python3 tools/cost_stage.py 8 | tee -a gpurun_out/cost_stage.txt
rm -rf gpurun_out/prof_cost
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cost -- python3 tools/cost_stage.py $B > gpurun_out/prof_cost.log 2>&1
cp $(find gpurun_out/prof_cost -name "*kernel_stats.csv" | head -1) gpurun_out/cost_kernel_stats.csv
rm -rf gpurun_out/prof_cost
cut -c1-170 gpurun_out/cost_kernel_stats.csv | head -12
fi
C1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
C2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS"
for P in 1 2 3 4; do
  case $P in 1) C="$C1";; 2) C="$C2";; 3) C="FETCH_SIZE";; 4) C="WRITE_SIZE";; esac
  rm -rf gpurun_out/pmc_cost_$P
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_cost_$P -- python3 tools/cost_stage.py $B ${MODES:-fused} axis,general 2 > gpurun_out/pmc_cost_$P.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/cost_counters.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_cost_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fsgm::", "")[:48]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not any(s in k for s in ("costbox", "census", "rawcost", "box5")): continue
    a = acc[k]
    n = len(a["GRBM_GUI_ACTIVE"]) or 1
    # dispatches: 2 map kinds x (1 warm-up + 2 timed); report first half (axis) and second half (general) separately
    def mean(name, half):
        v = a.get(name, [])
        h = len(v) // 2
        v = v[:h] if half == 0 else v[h:]
        return sum(v) / len(v) if v else float("nan")
    for half, kind in ((0, "axis"), (1, "general")):
        cyc = mean("GRBM_GUI_ACTIVE", half) / 8
        print(f"{k} [{kind}]: cycles {cyc:.3e} VALU insts/wave {mean('SQ_INSTS_VALU', half) / max(mean('SQ_WAVES', half), 1):.0f} waves {mean('SQ_WAVES', half):.0f} "
              f"VALU busy {mean('SQ_ACTIVE_INST_VALU', half) * 4 / (cyc * 1024):.3f} wait_any/wave_cycles {mean('SQ_WAIT_ANY', half) / mean('SQ_WAVE_CYCLES', half):.3f} "
              f"wait_inst_any {mean('SQ_WAIT_INST_ANY', half) / mean('SQ_WAVE_CYCLES', half):.3f} LDS busy {mean('SQ_ACTIVE_INST_LDS', half) * 4 / (cyc * 1024):.3f} "
              f"bank conflict/LDS active {mean('SQ_LDS_BANK_CONFLICT', half) / max(mean('SQ_LDS_IDX_ACTIVE', half), 1):.3f} "
              f"VMEM rd/wr insts per wave {mean('SQ_INSTS_VMEM_RD', half) / max(mean('SQ_WAVES', half), 1):.0f}/{mean('SQ_INSTS_VMEM_WR', half) / max(mean('SQ_WAVES', half), 1):.0f} "
              f"FETCH x2 MiB {2 * mean('FETCH_SIZE', half) / 1024:.1f} WRITE MiB {mean('WRITE_SIZE', half) / 1024:.1f}")
PY
rm -rf gpurun_out/pmc_cost_*
