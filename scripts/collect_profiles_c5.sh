#!/bin/bash
# PMC passes of the C5 frame (large-system route): FETCH_SIZE, WRITE_SIZE, MFMA busy -> gpurun_out/prof_c5/pmc_summary.csv
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_c5
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --workload C5 --steps 6 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $B > "$out/trace.log" 2>&1 || exit 1
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- $B > "$out/pmc_fetch.log" 2>&1 || exit 1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- $B > "$out/pmc_write.log" 2>&1 || exit 1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_mfma" -- $B > "$out/pmc_mfma.log" 2>&1 || exit 1
echo "mfma done"
cd "$root"
python3 scripts/pmc_summary.py "$out/pmc_summary.csv" "$out"/pmc_fetch/* "$out"/pmc_write/* "$out"/pmc_mfma/* 2>&1 | tail -2
stats=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
cp "$stats" "$out/kernel_stats.csv"
python3 -c "import sys; sys.path.insert(0, '$root'); import bench; print(bench.kernel_sources_digest())" > "$out/pmc_summary.src_sha256"
rm -rf "$out"/trace "$out"/pmc_fetch "$out"/pmc_write "$out"/pmc_mfma
