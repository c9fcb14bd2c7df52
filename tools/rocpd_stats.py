"""Per-kernel totals of a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`): python3 tools/rocpd_stats.py <results.db> [top]."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, count(*), avg(d.end - d.start), sum(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id "
                  "group by s.kernel_name order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)
print(f"total {tot / 1e3:.1f} us in {sum(r[1] for r in rows)} dispatches")
for n, c, a, s in rows[:top]:
    print(f"  {c:5d} x {a / 1e3:9.1f} us  {s / tot * 100:5.1f} %  {n[:120]}")
