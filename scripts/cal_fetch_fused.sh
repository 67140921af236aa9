#!/bin/bash
# Is the x2 correction of FETCH_SIZE (gfx950: coalesced streaming reads count half) also right for the FUSED HI launch?
# The same frames twice under rocprofv3 --pmc: once as the product runs them (rank update inside the sweep's launch) and once
# with the rank update as a launch of its own (diagnostic library, RSLAM_SWEEP_EXP bit 7) -- whose x2 was calibrated on a launch
# that must read all of P (scripts/cal_fetch.sh).  Raw counter of the fused launch against sweep-only + stand-alone rank update.
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for mode in fused apart; do
  if [ $mode = apart ]; then export RSLAM_SWEEP_EXP=128; else unset RSLAM_SWEEP_EXP; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $root/gpurun_out/calf_${mode}_$ctr -- python3 $root/scripts/ab_frame.py --debug 1 > $root/gpurun_out/calf_${mode}_$ctr.log 2>&1 || exit 1
  done
done
unset RSLAM_SWEEP_EXP
cd $root
python3 - <<PY
import csv, glob, statistics
def rows(mode, ctr):
    out = {}
    for f in glob.glob("gpurun_out/calf_%s_%s/**/*counter_collection.csv" % (mode, ctr), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            out.setdefault(k, []).append(float(r["Counter_Value"]))
    return out
from collections import Counter
for mode in ("fused", "apart"):
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, vs in rows(mode, ctr).items():
            if "sweep_persistent" in k or "rank_update" in k:
                print(mode, ctr, k[:40], sorted(Counter(int(round(v / 500.0) * 500) for v in vs).items()))
def hi_mode(vs):
    """the HI launch = the largest value that occurs often (the check frames of scripts/ab_frame.py run the product library)"""
    c = Counter(int(round(v / 500.0) * 500) for v in vs)
    top = max(b for b, n in c.items() if n >= 10)
    return statistics.median([v for v in vs if abs(v - top) <= 500])
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    fu, ap = rows("fused", ctr), rows("apart", ctr)
    sw_f = hi_mode([v for k, vs in fu.items() if "sweep_persistent" in k for v in vs])
    sw_a = hi_mode([v for k, vs in ap.items() if "sweep_persistent" in k for v in vs])
    ru_a = hi_mode([v for k, vs in ap.items() if "rank_update_kernel<true>" in k for v in vs])
    print("%s KiB, raw counter, HI launches: fused %.0f | sweep alone %.0f + stand-alone rank update %.0f = %.0f | fused - sum = %+.0f"
          % (ctr, sw_f, sw_a, ru_a, sw_a + ru_a, sw_f - sw_a - ru_a))
PY
rm -rf gpurun_out/calf_*_FETCH_SIZE gpurun_out/calf_*_WRITE_SIZE
