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
