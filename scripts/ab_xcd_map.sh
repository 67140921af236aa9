#!/bin/bash
# Round 6: the tile workers' XCD-aware tile map (kernels.hip WkMap) against the round-3 round-robin assignment (RSLAM_SWEEP_EXP
# bit 12), same diagnostic library, inside ONE gpurun call: ms per frame (hipGraph replay, interleaved), the stamps of the HI
# launch, and FETCH_SIZE / WRITE_SIZE of the fused launch under rocprofv3 --pmc (separate passes).
#   scripts/ab_xcd_map.sh [lib.so]        -> gpurun_out/xcdmap/
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
lib=$(realpath ${1:-$root/ransac_slam_amd/librslam_hip_dbg.so})
out=$root/gpurun_out/xcdmap
mkdir -p "$out"
cd "$root"
for rep in 1 2 3; do
  for mask in 0 4096; do
    for compat in 1 0; do
      RSLAM_SWEEP_EXP=$mask RSLAM_HIP_LIB_DEBUG=$lib timeout -k 10 120 python scripts/ab_frame.py --debug $compat 2>&1 | grep -E "ms/frame|factor_hi" | sed "s/^/mask $mask: /" || exit 1
    done
  done
done | tee "$out/timing.txt"
for mask in 0 4096; do
  RSLAM_SWEEP_EXP=$mask RSLAM_HIP_LIB_DEBUG=$lib timeout -k 10 120 python scripts/sweep_stamps.py > "$out/stamps_$mask.txt" 2>&1 || exit 1
  grep -A9 "tile worker 0" "$out/stamps_$mask.txt" | head -10; grep "LAST END" "$out/stamps_$mask.txt" | head -1
done
cd /tmp && export TMPDIR=/tmp
for mask in 0 4096; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    RSLAM_SWEEP_EXP=$mask RSLAM_HIP_LIB_DEBUG=$lib rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/pmc_${mask}_$ctr" -- python3 "$root/scripts/ab_frame.py" --debug 1 > "$out/pmc_${mask}_$ctr.log" 2>&1 || { tail -5 "$out/pmc_${mask}_$ctr.log"; exit 1; }
  done
done
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, statistics, sys
from collections import Counter
out = sys.argv[1]
for mask in (0, 4096):
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        vs = []
        for f in glob.glob("%s/pmc_%d_%s/**/*counter_collection.csv" % (out, mask, ctr), recursive=True):
            for r in csv.DictReader(open(f)):
                if "sweep_persistent" in r["Kernel_Name"]:
                    vs.append(float(r["Counter_Value"]))
        c = Counter(int(round(v / 500.0) * 500) for v in vs)
        top = max(b for b, n in c.items() if n >= 10)
        print("mask %4d %-10s fused HI launch: median %.0f KiB over %d launches" % (mask, ctr, statistics.median([v for v in vs if abs(v - top) <= 500]), len(vs)))
PY
rm -rf "$out"/pmc_*_FETCH_SIZE "$out"/pmc_*_WRITE_SIZE
