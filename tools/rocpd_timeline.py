"""Condensed kernel timeline of a rocprofv3 --kernel-trace database: consecutive solve kernels of one queue are folded into one SOLVE line."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = cur.execute(f"select d.start, d.end, d.queue_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
t0 = rows[0][0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
solve = ('k_eliminate_grouped', 'k_backsub', 'k_reduced_solve', 'k_lm_update', 'k_linearise', 'k_slab')
runs = []
for st, en, q, name in rows:
    a, b = (st - t0) / 1e6, (en - t0) / 1e6
    n = name
    for pre in ('void slslam::', 'slslam::', '_ZN6slslam', '_ZN12_GLOBAL__N_1'):
        n = n.replace(pre, '')
    n = n.split('(')[0][:30]
    if any(x in n for x in solve):
        if runs and runs[-1][3] == 'SOLVE' and runs[-1][2] == q and a - runs[-1][1] < 2:
            runs[-1][1] = max(runs[-1][1], b)
            continue
        runs.append([a, b, q, 'SOLVE'])
    else:
        runs.append([a, b, q, n])
# copies of the copy engines (rocprofv3 --memory-copy-trace), when the database has them: size and rate of every copy above 1 MB
mc = [t for t in tabs if 'memory_copy' in t and 'rocpd_memory_copy' in t] or [t for t in tabs if 'memory_copy' in t]
if mc:
    cols = [r[1] for r in cur.execute(f"pragma table_info({mc[0]})")]
    if all(c in cols for c in ('start', 'end', 'size')):
        for st, en, size in cur.execute(f"select start, end, size from {mc[0]} order by start"):
            a, b = (st - t0) / 1e6, (en - t0) / 1e6
            if size >= (1 << 20) and b > a:
                runs.append([a, b, -1, "COPY %.1f MB at %.1f GB/s" % (size / 1e6, size / 1e9 / ((b - a) / 1e3))])
        runs.sort(key=lambda r: r[0])
    else:
        print("# memory copy table %s has columns %s" % (mc[0], cols))
for a, b, q, n in runs:
    if b - a > 0.25 and a >= lo and a <= hi:
        print("%9.2f %9.2f  %7.2f ms  q%d  %s" % (a, b, b - a, q, n))
