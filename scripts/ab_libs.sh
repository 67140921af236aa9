#!/bin/bash
# interleaved A/B of development libraries (ransac_slam_amd/_dev/<name>.so) inside ONE gpurun call: ms per frame in hipGraph
# replay, both arithmetic modes, `reps` rounds over all names; the posterior of each against the product library's.
#   [AB_EAGER=1] scripts/ab_libs.sh <reps> name1 name2 ...      (AB_EAGER: stream-ordered launches instead of hipGraph replays)
reps=$1; shift
for r in $(seq 1 $reps); do
  for n in "$@"; do
    for compat in 1 0; do
      RSLAM_HIP_LIB_DEBUG=ransac_slam_amd/_dev/$n.so timeout -k 10 120 python scripts/ab_frame.py --debug ${AB_EAGER:+--eager} $compat 2>&1 | grep -E "ms/frame|MISMATCH" | sed "s/^/$n: /" || exit 1
    done
  done
done
