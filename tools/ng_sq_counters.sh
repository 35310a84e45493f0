set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
C1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
C2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS"
for P in 1 2; do
  case $P in 1) C="$C1";; 2) C="$C2";; esac
  rm -rf gpurun_out/pmc_ng_$P
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_ng_$P -- python3 tools/ng_batch.py ${NGB:-8} > gpurun_out/pmc_ng_$P.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/ng_counters.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_ng_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fsgm::", "")[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not k.startswith("ng_"): continue
    a = acc[k]
    def top(name):      # the largest dispatch (level 1)
        v = a.get(name, [])
        return max(v) if v else float("nan")
    cyc = top("GRBM_GUI_ACTIVE") / 8
    if not (cyc > 2e4): continue
    print(f"{k}: cycles {cyc:.3e} waves {top('SQ_WAVES'):.0f} VALU/wave {top('SQ_INSTS_VALU') / max(top('SQ_WAVES'), 1):.0f} VALU busy {top('SQ_ACTIVE_INST_VALU') * 4 / (cyc * 1024):.3f} "
          f"wait_any {top('SQ_WAIT_ANY') / top('SQ_WAVE_CYCLES'):.3f} wait_inst_any {top('SQ_WAIT_INST_ANY') / top('SQ_WAVE_CYCLES'):.3f} wait_inst_lds {top('SQ_WAIT_INST_LDS') / top('SQ_WAVE_CYCLES'):.3f} "
          f"LDS busy {top('SQ_ACTIVE_INST_LDS') * 4 / (cyc * 1024):.3f} LDS insts/wave {top('SQ_INSTS_LDS') / max(top('SQ_WAVES'), 1):.0f} conflict {top('SQ_LDS_BANK_CONFLICT') / max(top('SQ_LDS_IDX_ACTIVE'), 1):.3f} "
          f"VMEM rd/wr per wave {top('SQ_INSTS_VMEM_RD') / max(top('SQ_WAVES'), 1):.0f}/{top('SQ_INSTS_VMEM_WR') / max(top('SQ_WAVES'), 1):.0f} SALU/wave {top('SQ_INSTS_SALU') / max(top('SQ_WAVES'), 1):.0f} occupancy(wave_cycles*4/(cyc*1024)) {top('SQ_WAVE_CYCLES') * 4 / (cyc * 1024):.2f}")
PY
rm -rf gpurun_out/pmc_ng_*
