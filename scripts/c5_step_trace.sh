#!/bin/bash
# per-launch durations of the large-system factor sweep (C5): kernel trace of a few frames, last frame listed in launch order
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/c5trace
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/t" -- python3 $root/bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$out/log.txt" 2>&1 || exit 1
cd "$root"
f=$(find "$out/t" -name "*kernel_trace.csv" | head -1)
python3 - "$f" > "$out/steps.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 80 launches
t0 = None
for r in rows[-80:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None: t0 = s
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} {r['Kernel_Name'][:40]}")
PY
rm -rf "$out/t"
tail -70 "$out/steps.txt"
