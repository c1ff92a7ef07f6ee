#!/usr/bin/env python
"""Kernel summary (name, calls, total / average duration) from a rocprofv3 `*_results.db` (rocpd sqlite output), written as
the same CSV columns rocprofv3's --stats kernel_stats.csv uses.   python tools/rocpd_stats.py results.db [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute(f"pragma table_info({sym})")]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    rows = db.execute(f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                      f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, c, t, lo, hi in rows:
        out.writerow([n, c, t, round(t / c, 1), round(100.0 * t / total, 2), lo, hi])


if __name__ == "__main__":
    main()
