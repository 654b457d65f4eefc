#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats result database into a small CSV summary (profiles/*.csv).

    python tools/rocprof_summary.py gpurun_out/prof_r01/r01_results.db profiles/r01_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name, calls, "%.3f" % tot, "%.3f" % avg, "%.3f" % pct])
    print("wrote %s (%d kernels)" % (out_path, len(rows)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
