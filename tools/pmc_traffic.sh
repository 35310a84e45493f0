# HBM/fabric traffic of the headline bench's aggregation stage: rocprofv3 PMC passes, FETCH_SIZE and WRITE_SIZE in
# separate passes (they do not fit one), as MI355X_MICROARCH.md prescribes.  Each counter is collected for two runs of
# the bench that differ only in the number of timed steps; the difference of the totals is exactly (S2 - S1) steps of
# the stage, whatever else the bench launches around them.  Run on the GPU box through gpurun; results land in
# gpurun_out/pmc_traffic$TAG.{txt,json} (copy to profiles/rNN_pmc_traffic$TAG.*).
#   BENCH_ARGS="--paths 4" TAG=_paths4 bash tools/pmc_traffic.sh      the shipped 4-path configuration
#   BENCH_ARGS="--frames-per-gpu 256" TAG=_b256 ...                    another batch size / pipeline
set -e
BENCH_ARGS="${BENCH_ARGS:-}"
TAG="${TAG:-}"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S1=2; S2=6
for C in FETCH_SIZE WRITE_SIZE; do
  for S in $S1 $S2; do
    rm -rf gpurun_out/pmc_${C}_$S
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_${C}_$S -- python3 bench.py --steps $S --warmup 1 --no-cpu-baseline --no-extras $BENCH_ARGS > gpurun_out/pmc_${C}_$S.log 2>&1
  done
done
python3 - $S1 $S2 "$TAG" <<'PY' | tee gpurun_out/pmc_traffic$TAG.txt
import csv, glob, collections, json, re, sys
S1, S2, TAG = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
def load(d, name):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)[0]
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    return acc, cnt
f1, fc1 = load(f"pmc_FETCH_SIZE_{S1}", "FETCH_SIZE"); f2, fc2 = load(f"pmc_FETCH_SIZE_{S2}", "FETCH_SIZE")
w1, wc1 = load(f"pmc_WRITE_SIZE_{S1}", "WRITE_SIZE"); w2, wc2 = load(f"pmc_WRITE_SIZE_{S2}", "WRITE_SIZE")
b = json.loads([l for l in open(f"gpurun_out/pmc_FETCH_SIZE_{S2}.log") if l.startswith("{")][-1])
vox = b["config"]["frames_per_gpu"] * 1242 * 375 * 128
n = S2 - S1
print(f"difference of a {S2}-step and a {S1}-step run = {n} steps of {b['config']['frames_per_gpu']} frames; counter unit KiB; FETCH_SIZE doubled (gfx950)")
print("kernel | dispatches per step | FETCH_SIZE x2 per dispatch MiB | WRITE_SIZE per dispatch MiB | bytes per voxel per step")
tot = 0.0
kern = {}
for k in sorted(set(f2) | set(w2)):
    dn = fc2.get(k, 0) - fc1.get(k, 0)
    if dn <= 0 or not re.search(r"sweep_kernel|strip_kernel|band_kernel|pairx?_(ckpt|sum)_kernel|sweep_finish|agg_packed_kernel|wta_packed_kernel|wta_sweep_kernel", k): continue
    fb = 2 * (f2.get(k, 0) - f1.get(k, 0)) * 1024; wb = (w2.get(k, 0) - w1.get(k, 0)) * 1024
    if "sweep_finish" not in k: tot += fb + wb
    kern[k] = {"dispatches_per_step": dn / n, "fetch_MiB_x2_per_dispatch": fb / dn / 2**20, "write_MiB_per_dispatch": wb / dn / 2**20,
               "bytes_per_voxel": (fb + wb) / n / vox}
    print(f"{k} | {dn / n:g} | {fb / dn / 2**20:.1f} | {wb / dn / 2**20:.1f} | {(fb + wb) / n / vox:.3f}")
import hashlib, datetime
out = {"date": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"), "lib_sha16": hashlib.sha256(open("fsgm_amd/libfsgm_hip.so", "rb").read()).hexdigest()[:16],
       "pipeline": b["config"]["kernel"], "paths": int(re.search(r"(\d) paths", b["config"]["workload"]).group(1)), "frames_per_gpu": b["config"]["frames_per_gpu"], "bytes_per_voxel": tot / n / vox,
       "steps_profiled": n, "kernels": kern,
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md); per step = difference of two bench runs that differ only in --steps; aggregation-stage kernels only (the finish kernel is listed, not summed)",
       "command": "tools/pmc_traffic.sh", "bench_args": b.get("argv", "")}
json.dump(out, open(f"gpurun_out/pmc_traffic{TAG}.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("pipeline", "frames_per_gpu", "bytes_per_voxel", "steps_profiled")}))
PY
rm -rf gpurun_out/pmc_FETCH_SIZE_* gpurun_out/pmc_WRITE_SIZE_*
