#!/bin/bash
# kernel trace (launches of the last frame, start order, queue, duration) of ONE variant of scripts/ab_c5.py
#   scripts/c5_ab_trace.sh <variant 0..3> [compat]      -> gpurun_out/c5ab/steps_v<variant>_c<compat>.txt
set -o pipefail
v=${1:-1}; compat=${2:-1}; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/c5ab
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
AB_ONLY=$v rocprofv3 --kernel-trace --output-format csv -d "$out/t$v" -- python3 $root/scripts/ab_c5.py $compat 1000 "$@" > "$out/log_v${v}_c$compat.txt" 2>&1 || { tail -5 "$out/log_v${v}_c$compat.txt"; exit 1; }
cd "$root"
f=$(find "$out/t$v" -name "*kernel_trace.csv" | head -1)
python3 - "$f" > "$out/steps_v${v}_c$compat.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import os
starts = [i for i, r in enumerate(rows) if "predict_kernel<false" in r["Kernel_Name"]]
which = int(os.environ.get("TRACE_FRAME", "-1"))          # -1: the last frame (a timed one: events between the stages); e.g. 12: a frame of the untimed back-to-back batches
start = starts[which]
rows = rows[start:(starts[which + 1] if which >= 0 and which + 1 < len(starts) else None)]
t0 = int(rows[0]["Start_Timestamp"])
queues = {}
for r in rows:
    q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("rslam::", "").replace("void ", "")
    print(f"q{q} {(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  grid {int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) // max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)))):>6} {name[:48]}")
print("frame: %.1f us" % ((max(int(r["End_Timestamp"]) for r in rows) - t0) / 1e3))
PY
rm -rf "$out/t$v"
tail -40 "$out/steps_v${v}_c$compat.txt"
