#!/bin/bash
# everything profiles/r05_* is made from, in one GPU call (run through gpurun; copy gpurun_out/r05/* into profiles/ afterwards)
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r05
mkdir -p "$out"
cd "$root"
timeout -k 10 600 scripts/collect_profiles.sh r05 > "$out/collect_c3.log" 2>&1 || { tail -5 "$out/collect_c3.log"; exit 1; }
cp gpurun_out/prof_r05/kernel_stats.csv "$out/r05_bench_c3_kernel_stats.csv"
cp gpurun_out/prof_r05/pmc_summary.csv "$out/r05_bench_c3_pmc_summary.csv"
cp gpurun_out/prof_r05/pmc_summary.src_sha256 "$out/r05_bench_c3_pmc_summary.src_sha256"
echo "c3 profiles done"
timeout -k 10 600 scripts/collect_profiles_c5.sh > "$out/collect_c5.log" 2>&1 || { tail -5 "$out/collect_c5.log"; exit 1; }
cp gpurun_out/prof_c5/kernel_stats.csv "$out/r05_bench_c5_kernel_stats.csv"
cp gpurun_out/prof_c5/pmc_summary.csv "$out/r05_bench_c5_pmc_summary.csv"
cp gpurun_out/prof_c5/pmc_summary.src_sha256 "$out/r05_bench_c5_pmc_summary.src_sha256"
echo "c5 profiles done"
timeout -k 10 120 python scripts/sweep_stamps.py > "$out/r05_sweep_stamps_c3_compat1.txt" 2>&1 || exit 1
timeout -k 10 120 python scripts/sweep_stamps.py --compat 0 > "$out/r05_sweep_stamps_c3_compat0.txt" 2>&1 || exit 1
echo "stamps done"
{ timeout -k 10 300 python scripts/ab_c5.py 1; timeout -k 10 300 python scripts/ab_c5.py 0; } > "$out/r05_c5_ab.txt" 2>&1 || exit 1
for v in 0 1 2; do
  timeout -k 10 200 scripts/c5_ab_trace.sh $v 1 > /dev/null 2>&1 || exit 1
  cp gpurun_out/c5ab/steps_v${v}_c1.txt "$out/r05_c5_trace_v${v}.txt"
done
echo "c5 ab + traces done"
{ echo "== scripts/probes/cu_mask 32"; timeout -k 10 60 scripts/probes/cu_mask 32; echo; echo "== scripts/probes/mask_alone"; timeout -k 10 60 scripts/probes/mask_alone;
  echo; echo "== scripts/probes/rocblas_yardstick (vendor library: a yardstick, never linked into the product)"; timeout -k 10 120 scripts/probes/rocblas_yardstick;
  echo; echo "== scripts/probes/gemm128_loop"; timeout -k 10 120 scripts/probes/gemm128_loop; } > "$out/r05_probes.txt" 2>&1 || exit 1
echo "probes done"
{ timeout -k 10 300 python scripts/ab_front.py; timeout -k 10 300 python scripts/ab_front.py 1000 1000; } > "$out/r05_front_ab.txt" 2>&1 || { tail -3 "$out/r05_front_ab.txt"; exit 1; }
echo "front-end ab done"
{ timeout -k 10 500 python scripts/soak.py 1 100000; timeout -k 10 400 python scripts/soak.py 0 100000; timeout -k 10 400 python scripts/soak.py 0 100000 300 1000 3;
  timeout -k 10 300 python scripts/soak.py 1 2000 1000 1000 4; timeout -k 10 300 python scripts/soak.py 0 1500 1000 1000 4; } > "$out/r05_soak.txt" 2>&1 || { tail -3 "$out/r05_soak.txt"; exit 1; }
echo "soak done"
