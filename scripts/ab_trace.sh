#!/bin/bash
# kernel-trace medians of the resident C3 frame for the current library, optionally with an environment switch:
#   scripts/ab_trace.sh tag [VAR=1]
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$root/gpurun_out/abtrace_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
# (AB_DEBUG=1: the diagnostic variant of the library, which reads RSLAM_SWEEP_EXP; AB_EAGER=1: stream-ordered launches instead of
#  hipGraph replays; AB_COMPAT: arithmetic mode, default 1)
rocprofv3 --kernel-trace --output-format csv -d "$out/t" -- python3 $root/scripts/ab_frame.py ${AB_DEBUG:+--debug} ${AB_EAGER:+--eager} ${AB_COMPAT:-1} > "$out/log.txt" 2>&1 || exit 1
cd "$root"
f=$(find "$out/t" -name "*kernel_trace.csv" | head -1)
echo "== $tag $*"; python3 scripts/kernel_medians.py "$f"
rm -rf "$out/t"
