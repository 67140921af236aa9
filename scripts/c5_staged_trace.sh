#!/bin/bash
# kernel trace of the staged large-system route (C5): the launches of the last frame in start order with their queue
# (= stream: context stream, S stage, group inverses, R stage), start and duration
#   scripts/c5_staged_trace.sh [compat]        -> gpurun_out/c5staged/steps_compat<compat>.txt
set -o pipefail
compat=${1:-1}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/c5staged
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/t$compat" -- python3 $root/bench.py --workload C5 --compat $compat --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$out/log$compat.txt" 2>&1 || exit 1
cd "$root"
f=$(find "$out/t$compat" -name "*kernel_trace.csv" | head -1)
python3 - "$f" > "$out/steps_compat$compat.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame: from the last launch of the first prediction kernel on
start = max(i for i, r in enumerate(rows) if "predict_kernel<false" in r["Kernel_Name"] or "predict_kernelILb0" in r["Kernel_Name"])
rows = rows[start:]
t0 = int(rows[0]["Start_Timestamp"])
queues = {}
for r in rows:
    q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("rslam::", "").replace("void ", "")
    print(f"q{q} {(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  grid {int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) // max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)))):>6} {name[:48]}")
print("frame: %.1f us" % ((max(int(r["End_Timestamp"]) for r in rows) - t0) / 1e3))
PY
rm -rf "$out/t$compat"
tail -150 "$out/steps_compat$compat.txt"
