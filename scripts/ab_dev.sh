#!/bin/bash
# A/B of kernel experiments inside ONE gpurun call (boxes differ by a few per cent):
#   scripts/ab_dev.sh base peel early ...     (libraries ransac_slam_amd/_dev/<name>.so, see ransac_slam_amd/build.py build_dev)
# per library: ms/frame in hipGraph replay (compat 1 and 0) and the chain's per-block stamps of one frame
for n in "$@"; do
  lib=ransac_slam_amd/_dev/$n.so
  echo "=== $n"
  RSLAM_HIP_LIB_DEBUG=$lib timeout -k 10 120 python scripts/ab_frame.py --debug 1 || exit 1
  RSLAM_HIP_LIB_DEBUG=$lib timeout -k 10 120 python scripts/ab_frame.py --debug 0 || exit 1
  RSLAM_HIP_LIB_DEBUG=$lib timeout -k 10 120 python scripts/sweep_stamps.py > gpurun_out/stamps_$n.txt 2>&1 || { tail -5 gpurun_out/stamps_$n.txt; exit 1; }
  sed -n 2,11p gpurun_out/stamps_$n.txt
  grep -A9 "tile worker 0" gpurun_out/stamps_$n.txt | head -10
  grep "LAST END" gpurun_out/stamps_$n.txt
done
