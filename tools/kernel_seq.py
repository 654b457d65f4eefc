#!/usr/bin/env python3
"""Per-launch durations of one forward in launch order, from a rocprofv3 --kernel-trace CSV (no in-library events between
the launches): python tools/kernel_seq.py <kernel_trace.csv> [launches per step]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void se::", "").replace("se::", "")[:34] for r in rows]
# the last full step: find the last pack_m_kernel (first kernel of a forward)
starts = [i for i, n in enumerate(names) if n.startswith("pack_m")]
a = starts[-2] if len(starts) >= 2 else 0
b = starts[-1] if len(starts) >= 2 else len(rows)
prev_end = None
for i in range(a, b):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%3d %-34s %8.1f us  gap %6.1f us" % (i - a, names[i], (e - s) / 1e3, gap))
    prev_end = e
print("step span %.1f us" % ((int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3))
