set -o pipefail
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for r in 64 498; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/cal_$r -- python3 $root/scripts/k10_bench.py 1813 $r > $root/gpurun_out/cal_$r.log 2>&1 || exit 1
  python3 - <<PY
import csv,glob,statistics
v=[float(r["Counter_Value"]) for f in glob.glob("$root/gpurun_out/cal_$r/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "rank_update" in r["Kernel_Name"]]
print("r=$r FETCH_SIZE KiB: n", len(v), "median", statistics.median(v), "min", min(v), "max", max(v))
PY
  rm -rf $root/gpurun_out/cal_$r
done
