#!/bin/bash
# rocprofv3 evidence of one round (run on the MI355X box through gpurun): kernel trace + three PMC passes of the bench
# command at C3.  usage: scripts/collect_profiles.sh r02   ->  gpurun_out/prof_r02/...  (copy the summaries into profiles/)
set -o pipefail
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $B --steps 50 --warmup 5 > "$out/trace.log" 2>&1 || exit 1
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- $B --steps 20 --warmup 3 --no-graph > "$out/pmc_fetch.log" 2>&1 || exit 1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- $B --steps 20 --warmup 3 --no-graph > "$out/pmc_write.log" 2>&1 || exit 1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_mfma" -- $B --steps 20 --warmup 3 --no-graph > "$out/pmc_mfma.log" 2>&1 || exit 1
echo "mfma done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$out/pmc_valu" -- $B --steps 20 --warmup 3 --no-graph > "$out/pmc_valu.log" 2>&1 || echo "valu pass failed (counters unavailable?)"
cd "$root"
stats=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
cp "$stats" "$out/kernel_stats.csv"
python3 scripts/pmc_summary.py "$out/pmc_summary.csv" "$out"/pmc_fetch/* "$out"/pmc_write/* "$out"/pmc_mfma/* "$out"/pmc_valu/* 2>&1 | tail -2
python3 -c "import sys; sys.path.insert(0, '$root'); import bench; print(bench.kernel_sources_digest())" > "$out/pmc_summary.src_sha256"
# keep the merge-back small: the raw traces stay on the box
rm -rf "$out/trace" "$out"/pmc_fetch "$out"/pmc_write "$out"/pmc_mfma "$out"/pmc_valu
ls -la "$out"
