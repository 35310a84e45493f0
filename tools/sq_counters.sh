# SQ counters of the headline bench's kernels (rocprofv3 --pmc, two passes of 8 SQ counters; the program directly
# after `--`).  Under counter collection the dispatches are serialised, so the numbers are per kernel in isolation.
# Run on the GPU box through gpurun; summary -> gpurun_out/sq_counters.md (copy to profiles/rNN_sq_counters.md).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ARGS="${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu-baseline}"
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq1 -- python3 bench.py $ARGS > gpurun_out/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_sq2 -- python3 bench.py $ARGS > gpurun_out/pmc_sq2.log 2>&1
python3 - <<'PY' | tee gpurun_out/sq_counters.md
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
dur = collections.defaultdict(list)
for d in ("pmc_sq1", "pmc_sq2"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for f in glob.glob(f"gpurun_out/{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
         "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS",
         "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_WAIT_INST_LDS"]
print("per dispatch averages (serialised dispatches); cycles of SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are quad-cycles summed over waves\n")
print("| kernel | n | avg us | " + " | ".join(names) + " |")
print("|---|---|---|" + "---|" * len(names))
for k in sorted(acc):
    if not any(s in k for s in ("sweep", "strip", "pair", "agg_", "wta", "fwd", "bwd", "copy16")): continue
    n = max(cnt[k].values())
    row = [f"{acc[k][c] / max(cnt[k][c], 1):.4g}" if cnt[k][c] else "-" for c in names]
    d = dur[k]
    print(f"| {k} | {n} | {sum(d) / max(len(d), 1):.1f} | " + " | ".join(row) + " |")
print("\nderived (per kernel): VALU instructions per wave; share of wave time with an instruction active / parked in s_waitcnt or barrier / issue-stalled;")
print("VALU issue cycles per SIMD-cycle while the kernel ran = SQ_ACTIVE_INST_VALU*4 / (GRBM_GUI_ACTIVE * 1024 SIMDs)\n")
print("| kernel | VALU inst/wave | active% | wait_any% | wait_inst% | VALU busy of SIMD time | LDS conflict cycles / LDS active |")
print("|---|---|---|---|---|---|---|")
for k in sorted(acc):
    a = acc[k]
    if not a.get("SQ_WAVES") or not any(s in k for s in ("sweep", "strip", "pair", "agg_", "wta", "fwd", "bwd")): continue
    wc = a["SQ_WAVE_CYCLES"] or 1
    gui = a.get("GRBM_GUI_ACTIVE", 0)
    print(f"| {k} | {a['SQ_INSTS_VALU'] / a['SQ_WAVES']:.0f} | {100 * a['SQ_ACTIVE_INST_ANY'] / wc:.1f} | {100 * a['SQ_WAIT_ANY'] / wc:.1f} | "
          f"{100 * a['SQ_WAIT_INST_ANY'] / wc:.1f} | {(a['SQ_ACTIVE_INST_VALU'] * 4 / (gui * 1024)) if gui else float('nan'):.3f} | "
          f"{a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 0), 1):.3f} |")
PY
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
