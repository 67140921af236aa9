#!/bin/bash
# bench.py with hipGraph replay (--graph) against stream-ordered launches (the default since round 6), interleaved inside ONE gpurun call
for r in 1 2 3; do
  for mode in "--graph" ""; do
    timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $mode 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line); s = d['summary']
        print('mode=%-10s' % ('$mode' or 'stream'), 'c3 %.4f compat0 %.4f c4 %.4f c5 %.3f' % (s['c3_ms'], s['c3_compat0_ms'], s['c4_ms'], s['c5_ms']), 'repeats', d.get('timed_region_repeats_ms_per_step'))
" || exit 1
  done
done
