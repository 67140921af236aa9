#!/bin/bash
# everything profiles/r06_* is made from (run through gpurun in two calls: `scripts/collect_r06.sh a` and `... b`; copy
# gpurun_out/r06/* into profiles/ afterwards)
set -o pipefail
part=${1:-a}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r06
mkdir -p "$out"
cd "$root"
if [ "$part" = a ]; then
  timeout -k 10 600 scripts/collect_profiles.sh r06 > "$out/collect_c3.log" 2>&1 || { tail -5 "$out/collect_c3.log"; exit 1; }
  cp gpurun_out/prof_r06/kernel_stats.csv "$out/r06_bench_c3_kernel_stats.csv"
  cp gpurun_out/prof_r06/pmc_summary.csv "$out/r06_bench_c3_pmc_summary.csv"
  cp gpurun_out/prof_r06/pmc_summary.src_sha256 "$out/r06_bench_c3_pmc_summary.src_sha256"
  echo "c3 profiles done"
  timeout -k 10 600 scripts/collect_profiles_c5.sh > "$out/collect_c5.log" 2>&1 || { tail -5 "$out/collect_c5.log"; exit 1; }
  cp gpurun_out/prof_c5/kernel_stats.csv "$out/r06_bench_c5_kernel_stats.csv"
  cp gpurun_out/prof_c5/pmc_summary.csv "$out/r06_bench_c5_pmc_summary.csv"
  cp gpurun_out/prof_c5/pmc_summary.src_sha256 "$out/r06_bench_c5_pmc_summary.src_sha256"
  echo "c5 profiles done"
  timeout -k 10 120 python scripts/sweep_stamps.py > "$out/r06_sweep_stamps_c3_compat1.txt" 2>&1 || exit 1
  timeout -k 10 120 python scripts/sweep_stamps.py --compat 0 > "$out/r06_sweep_stamps_c3_compat0.txt" 2>&1 || exit 1
  echo "stamps done"
else
  { timeout -k 10 500 python scripts/soak.py 1 100000; timeout -k 10 400 python scripts/soak.py 0 100000; timeout -k 10 400 python scripts/soak.py 0 100000 300 1000 3;
    timeout -k 10 300 python scripts/soak.py 1 2000 1000 1000 4; timeout -k 10 300 python scripts/soak.py 0 1500 1000 1000 4; } > "$out/r06_soak.txt" 2>&1 || { tail -3 "$out/r06_soak.txt"; exit 1; }
  echo "soak done"
  timeout -k 10 900 python scripts/fuzz_parity.py 400 6000 520 > "$out/r06_fuzz.txt" 2>&1 || { tail -5 "$out/r06_fuzz.txt"; exit 1; }
  echo "fuzz done"
fi
