"""median duration per kernel (and per launch position inside a frame) from a rocprofv3 kernel trace CSV"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"].split("(")[0][-30:], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
nframes = max((len(d) for k, d in by.items() if "best_mask" in k), default=0)       # (one consensus launch per frame)
for k, d in by.items():
    if len(d) > 500:
        if any(x in k for x in ("sweep", "predict", "pht")) and "rescue_pht" not in k and len(d) > 1.5 * nframes:
            a, b = st.median(d[0::2]), st.median(d[1::2]); tot += a + b
            print("  %-30s n=%d  1st %.2f  2nd %.2f" % (k, len(d), a, b))
        else:
            a = st.median(d); tot += a
            print("  %-30s n=%d  median %.2f" % (k, len(d), a))
print("  sum of medians per frame %.2f us" % tot)
# the idle time in front of each launch (its start minus the previous launch's end), median per kernel, steady-state frames only
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gaps = {}
for a, b in zip(rows[:-1], rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g < 50.0:          # (not the pauses between timed regions)
        gaps.setdefault(b["Kernel_Name"].split("(")[0][-30:], []).append(g)
gtot = 0.0
for k, d in gaps.items():
    if len(d) > 500:
        per_frame = len(d) / max(nframes, 1)
        gtot += st.median(d) * round(per_frame)
        print("  gap in front of %-30s n=%d  median %.2f  p90 %.2f" % (k, len(d), st.median(d), sorted(d)[int(0.9 * len(d))]))
print("  sum of median gaps per frame %.2f us" % gtot)
