#!/bin/bash
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r7z
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$out/counters.txt" 2>&1 || true
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/p_$tag" -- python3 $root/scripts/score_bench.py > "$out/log_$tag.txt" 2>&1 || echo "pass failed: $set"
done
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "score_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
            acc[(int(r["Grid_Size"]), int(r.get("Workgroup_Size", 0) or 0))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g in sorted(acc):
    print("grid", g, {k: round(sum(v) / len(v)) for k, v in sorted(acc[g].items())})
PY
f=$(find "$out" -name "*counter_collection.csv" | head -1); head -3 "$f" > "$out/sample_rows.txt"; python3 -c "
import csv,sys,collections
c=collections.Counter()
for r in csv.DictReader(open('$f')):
    c[(r['Kernel_Name'][:30], r['Grid_Size'])]+=1
print(c.most_common(12))" >> "$out/sample_rows.txt"
rm -rf "$out"/p_*
