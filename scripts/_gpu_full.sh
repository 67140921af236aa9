set -e
out=gpurun_out/full2
mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $out/tests.txt 2>&1 || { tail -40 $out/tests.txt; exit 1; }
tail -2 $out/tests.txt
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/full2/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d.get("stage_us")); print(d["summary"]); print(d.get("sequence",{}).get("graph_captures"), d.get("sequence",{}).get("sweep_reruns"))
PY
