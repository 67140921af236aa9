#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r8w; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 $root/scripts/repeat_score.py > $out/log.txt 2>&1 || exit 1
cd $root
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, statistics as st
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    seq.append((n, d))
# find runs of consecutive score launches
per = {}
it = []
for n, d in seq:
    if "sweep_persistent" in n:
        if len(it) == 6:                  # (the frame's own scoring launch + the five behind the frame)
            for k, v in enumerate(it): per.setdefault(k, []).append(v)
        it = []
    elif "score_" in n:
        it.append(d)
names = {0: "inside the frame, behind its P H^T launch", 1: "behind the frame and three predictions", 2: "again", 3: "again", 4: "again, behind an idle device", 5: "again, behind an idle device and another (tiny) kernel"}
for k in sorted(per): print("scoring launch #%d (%s): n=%d median %.2f us" % (k, names[k], len(per[k]), st.median(per[k])))
runs = {1: [], 2: [], 3: []}
i = 0
while i < len(seq):
    if "predict_kernel" in seq[i][0]:
        j = i
        while j < len(seq) and "predict_kernel" in seq[j][0]: j += 1
        if j - i == 3:
            for k in range(3): runs[k + 1].append(seq[i + k][1])
        i = j
    else:
        i += 1
for k in (1, 2, 3): print("prediction launch #%d in a row: n=%d median %.2f us" % (k, len(runs[k]), st.median(runs[k]) if runs[k] else float("nan")))
PY
rm -rf $out/t
