"""Per-kernel PMC averages from a rocprofv3 --pmc run (rocpd sqlite):  python tools/rocpd_pmc.py <db>"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
rows = db.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ix else [c for c in cols if "name" in c and "counter" not in c][0]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for r in rows:
    k = str(r[ix[name_col]]).split("(")[0]
    a = agg[k][r[ix["counter_name"]]]
    a[0] += float(r[ix["value"]])
    a[1] += 1
for k, d in sorted(agg.items()):
    print(k)
    for c, (s, n) in sorted(d.items()):
        print("   %-28s avg/dispatch %16.1f   dispatches %d" % (c, s / n, n))
