"""Per-launch durations of a rocprofv3 kernel trace in start order (rocpd sqlite): one line per LM iteration of the last bench step.
   python tools/rocpd_sequence.py <results.db> [launches of the sweep per step, default 10]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = db.execute("select name, start, duration from kernels order by start").fetchall()
fam = {"k_eliminate_grouped": "elim", "k_linearise_schur": "elim", "k_reduced_solve": "solve", "k_backsub": "backsub"}
seq = {}
for name, start, dur in rows:
    for k, f in fam.items():
        if k in name:
            seq.setdefault(f, []).append(dur / 1e3)
n = len(seq.get("elim", []))
steps = n // per
print("# %d elimination launches = %d steps of %d; per-launch durations (us) of the last step, then the mean over steps" % (n, steps, per))
for f in ("elim", "solve", "backsub"):
    v = seq.get(f, [])
    if len(v) < per:
        continue
    last = v[-per:]
    mean = [sum(v[s * per + i] for s in range(steps)) / steps for i in range(per)]
    print("%-8s last: %s" % (f, " ".join("%7.1f" % x for x in last)))
    print("%-8s mean: %s   sum %.1f" % (f, " ".join("%7.1f" % x for x in mean), sum(mean)))
