# SQ counters of the library's kernels (rocprofv3 --pmc, two passes of 8 SQ counters each; the program directly after
# `--`).  Under counter collection the dispatches are serialised, so the numbers are per kernel in isolation.
# Run on the GPU box through gpurun; summary -> gpurun_out/sq_counters.md (copy to profiles/rNN_sq_counters.md).
#
# Legs (each a command profiled twice): the calibration kernels of tools/ubench/counter_calib.hip (known behaviour: what
# the counters read for a saturated SIMD, for conflict-free b128 LDS traffic, for a 32-way conflict), then the benches
# named in LEGS (default: headline 8 paths, 4 paths, 8-frame batch, pyramid3, pyramid3_ng).
#
# Normalisation: rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md, "DVFS give-back"), so the cycles
# a kernel ran are GRBM_GUI_ACTIVE / 8 and the SIMD-cycles available to it GRBM_GUI_ACTIVE / 8 * 1024.  SQ_ACTIVE_INST_*
# count quad-cycles: "VALU busy" = SQ_ACTIVE_INST_VALU * 4 / (GRBM_GUI_ACTIVE / 8 * 1024).  Round 2's table divided by
# GRBM_GUI_ACTIVE * 1024 and so read 8x too low; the calibration rows are the sanity check (calib_valu_pk must read
# what a saturated SIMD reads).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
[ -x tools/ubench/counter_calib ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o tools/ubench/counter_calib tools/ubench/counter_calib.hip
C1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
C2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS"
rm -rf gpurun_out/pmc_sq
mkdir -p gpurun_out/pmc_sq
rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d gpurun_out/pmc_sq/calib_1 -- tools/ubench/counter_calib > gpurun_out/pmc_sq/calib_1.log 2>&1
rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d gpurun_out/pmc_sq/calib_2 -- tools/ubench/counter_calib > gpurun_out/pmc_sq/calib_2.log 2>&1
echo "calibration leg done"
LEGS="${LEGS:-default paths4 batch8 pyramid3 pyramid3_ng}"
for leg in $LEGS; do
    case $leg in
        default)     A="--steps 2 --warmup 1 --no-cpu-baseline" ;;
        default40)   A="--steps 2 --warmup 1 --no-cpu-baseline --frames-per-gpu 40" ;;
        paths4)      A="--steps 2 --warmup 1 --no-cpu-baseline --paths 4" ;;
        batch8)      A="--steps 2 --warmup 1 --no-cpu-baseline --frames-per-gpu 8" ;;
        batch1)      A="--steps 2 --warmup 1 --no-cpu-baseline --frames-per-gpu 1" ;;
        pyramid3)    A="--steps 4 --workload pyramid3" ;;
        pyramid3_ng) A="--steps 4 --workload pyramid3_ng" ;;
        *)           A="${leg//%/ }" ;;                 # a custom leg: bench.py arguments with % in place of spaces
    esac
    tag=$(echo "$leg" | tr -c 'A-Za-z0-9\n' '_' | cut -c1-40)
    rocprofv3 --kernel-trace --pmc $C1 --output-format csv -d gpurun_out/pmc_sq/${tag}_1 -- python3 bench.py $A > gpurun_out/pmc_sq/${tag}_1.log 2>&1
    rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d gpurun_out/pmc_sq/${tag}_2 -- python3 bench.py $A > gpurun_out/pmc_sq/${tag}_2.log 2>&1
    echo "leg $leg done"
done
python3 - <<'PY' | tee gpurun_out/sq_counters.md
import csv, glob, collections, os
legs = sorted({os.path.basename(d)[:-2] for d in glob.glob("gpurun_out/pmc_sq/*_1")}, key=lambda s: (s != "calib", s))
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
         "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS",
         "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_WAIT_INST_LDS"]
keep = ("calib_", "sweep", "strip", "band", "pair", "agg_", "wta", "copy16", "rawcost", "box5x5", "census", "pyd_rows", "pyd_", "ng_", "finish")
print("per dispatch averages (dispatches serialised by the counter collection).  SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are quad-cycles")
print("summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs (cycles the kernel ran = GRBM_GUI_ACTIVE / 8).\n")
for leg in legs:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.Counter())
    dur = collections.defaultdict(list)
    # a kernel launched with several grid sizes in one leg (the bench's whole-MEX extras run the cost stage on 1, 8 and
    # 512 frames) gets one row per grid size: a mean over dispatches of different sizes says nothing
    def grid_of(r):
        if r.get("Grid_Size"): return int(r["Grid_Size"])
        if r.get("Grid_Size_X"): return int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1)
        return 0
    def base_of(r): return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fsgm::", "")[:64]
    rows_c, rows_t, sizes = [], [], collections.defaultdict(set)
    for d in (f"{leg}_1", f"{leg}_2"):
        for f in glob.glob(f"gpurun_out/pmc_sq/{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                rows_c.append(r); sizes[base_of(r)].add(grid_of(r))
        for f in glob.glob(f"gpurun_out/pmc_sq/{d}/**/*kernel_trace.csv", recursive=True):
            rows_t.extend(csv.DictReader(open(f)))
    def key_of(r):
        b = base_of(r)
        return f"{b} [{grid_of(r) // 64} waves]" if len(sizes[b]) > 1 else b
    for r in rows_c:
        k = key_of(r)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for r in rows_t:
        dur[key_of(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    ks = [k for k in sorted(acc) if any(s in k for s in keep) and acc[k].get("SQ_WAVES")]
    if not ks:
        continue
    print(f"## leg `{leg}`\n")
    print("| kernel | n | avg us | clock GHz | VALU inst/wave | active% | wait_any% (s_waitcnt/barrier) | wait_inst% (issue stall) | VALU busy of SIMD time | LDS busy of CU time | LDS conflict / LDS active | VMEM rd+wr inst/wave |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for k in ks:
        a = {c: acc[k][c] / max(cnt[k][c], 1) for c in names}      # per dispatch
        n = max(cnt[k].values())
        us = sum(dur[k]) / max(len(dur[k]), 1)
        wc = a["SQ_WAVE_CYCLES"] or 1
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0                              # cycles the kernel ran (per XCD)
        ghz = cyc / (us * 1e3) if us else float("nan")
        valu_busy = a["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024) if cyc else float("nan")
        lds_busy = a["SQ_LDS_IDX_ACTIVE"] / (cyc * 256) if cyc else float("nan")   # LDS-array cycles per CU-cycle
        print(f"| {k} | {n} | {us:.1f} | {ghz:.2f} | {a['SQ_INSTS_VALU'] / a['SQ_WAVES']:.0f} | {100 * a['SQ_ACTIVE_INST_ANY'] / wc:.1f} | "
              f"{100 * a['SQ_WAIT_ANY'] / wc:.1f} | {100 * a['SQ_WAIT_INST_ANY'] / wc:.1f} | {valu_busy:.3f} | {lds_busy:.3f} | "
              f"{a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1):.3f} | {(a['SQ_INSTS_VMEM_RD'] + a['SQ_INSTS_VMEM_WR']) / a['SQ_WAVES']:.0f} |")
    print("\nraw per-dispatch averages:\n")
    print("| kernel | " + " | ".join(names) + " |")
    print("|---|" + "---|" * len(names))
    for k in ks:
        print(f"| {k} | " + " | ".join(f"{acc[k][c] / max(cnt[k][c], 1):.4g}" if cnt[k][c] else "-" for c in names) + " |")
    print()
PY
rm -rf gpurun_out/pmc_sq/*_1 gpurun_out/pmc_sq/*_2
