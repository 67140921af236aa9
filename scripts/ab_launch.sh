#!/bin/bash
# interleaved A/B inside ONE gpurun call: how the frames of a resident sequence are handed to the device -- hipGraph replay
# (rslam_step_frame(use_graph = 1)) against stream-ordered launches (use_graph = 0), hipGraph replay of the C3 frame, both modes.
#   scripts/ab_launch.sh <reps>
reps=$1; shift
for r in $(seq 1 $reps); do
  for compat in 1 0; do
    timeout -k 10 120 python scripts/ab_frame.py $compat 2>&1 | grep -E "ms/frame" | sed "s/^/hipGraph replay:         /" || exit 1
    timeout -k 10 120 python scripts/ab_frame.py --eager $compat 2>&1 | grep -E "ms/frame" | sed "s/^/stream-ordered launches: /" || exit 1
  done
done
