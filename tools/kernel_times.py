"""Print per-kernel totals from a rocprofv3 results .db (rocpd sqlite): kernel_times.py results.db [steps] [substring ...]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pats = sys.argv[3:]
rows = db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e6/steps:.3f} ms/step over {sum(r[1] for r in rows)//steps} launches/step")
for name, n, t, a in rows:
    if pats and not any(p in name for p in pats):
        continue
    print(f"{name[:96]:96s} {n/steps:7.1f} {t/1e6/steps:8.3f} ms {a/1e3:8.1f} us")
