"""rocprofv3 results .db (rocpd sqlite, the default output of ROCm 7.2) -> the kernel_stats.csv layout of `--stats --output-format csv`
(Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev): db_to_stats_csv.py results.db out.csv"""
import csv
import math
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), avg((end-start)*(end-start)) "
                  "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, n, t, a, mn, mx, a2 in rows:
        w.writerow([name, n, t, f"{a:.6f}", f"{100.0 * t / tot:.4f}", mn, mx, f"{math.sqrt(max(0.0, a2 - a * a)):.6f}"])
