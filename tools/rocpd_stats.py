#!/usr/bin/env python3
"""Write the per-kernel summary (calls, total / average duration, share) of a rocprofv3 rocpd
SQLite result file as CSV:  python tools/rocpd_stats.py results.db out.csv"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.3f" % r[4]])
    print("%d kernels, %.1f ms total" % (len(rows), sum(r[2] for r in rows) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
