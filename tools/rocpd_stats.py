"""Per-kernel summary (calls, total/avg/min/max ns, share) of a rocprofv3 rocpd .db, the same
columns as rocprofv3's kernel_stats.csv.  usage: python tools/rocpd_stats.py results.db [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
    "from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print(f"{'Name':70s} {'Calls':>7s} {'TotalNs':>13s} {'AvgNs':>11s} {'MinNs':>10s} {'MaxNs':>10s} {'Pct':>6s}")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else None]:
    print(f"{r[0][:70]:70s} {r[1]:7d} {r[2]:13d} {r[3]:11.0f} {r[4]:10d} {r[5]:10d} {100 * r[2] / total:6.2f}")
print(f"{'TOTAL':70s} {sum(r[1] for r in rows):7d} {total:13d}")
