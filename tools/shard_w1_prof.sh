# kernels of the sharded bootstrap filter at world = 1 over real RCCL (what does a step cost beyond the unsharded filter's kernel?)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/shw1 -- python3 bench.py --mode sharded --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/shw1.json 2> gpurun_out/shw1.err
cat gpurun_out/shw1.json | cut -c1-400
python3 - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/shw1/**/*kernel_stats.csv", recursive=True))[-1]
for row in list(csv.DictReader(open(f)))[:8]:
    print("  ", row["Name"][:90], row["Calls"], row["AverageNs"], row["Percentage"])
PY
