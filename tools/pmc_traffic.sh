# HBM/fabric traffic of the headline bench: two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE do
# not fit one pass), as MI355X_MICROARCH.md prescribes.  Run on the GPU box through gpurun; the per-kernel
# sums land in gpurun_out/pmc_traffic.txt.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_w.log 2>&1
python3 - <<'PY' | tee gpurun_out/pmc_traffic.txt
import csv, glob, collections
def load(d, name):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)[0]
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"][:70]
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    return acc, cnt
fa, fc = load("pmc_f", "FETCH_SIZE")
wa, wc = load("pmc_w", "WRITE_SIZE")
print("kernel | dispatches | FETCH_SIZE/dispatch MiB (raw) | x2 | WRITE_SIZE/dispatch MiB")
for k in sorted(set(fa) | set(wa)):
    n = fc.get(k, wc.get(k, 1))
    print(f"{k} | {n} | {fa.get(k,0)/n/1024:.1f} | {2*fa.get(k,0)/n/1024:.1f} | {wa.get(k,0)/max(wc.get(k,1),1)/1024:.1f}")
PY
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
