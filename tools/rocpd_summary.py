"""Turns a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the text summary committed
under profiles/:  python tools/rocpd_summary.py <results.db> > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                  "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
print("# rocprofv3 --kernel-trace --stats summary (durations in us)")
print("%-58s %7s %12s %10s %10s %10s %6s %5s %5s %7s %9s %4s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds_B", "grid", "wg"))
for r in rows:
    name = r[0].split("(")[0][:58]
    print("%-58s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %7d %9d %4d" % (
        name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0))
