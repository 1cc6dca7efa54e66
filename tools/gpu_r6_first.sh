#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6_first}
rm -rf gpurun_out/${TAG}_prof
timeout 900 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_prof -o t -- python tools/first_solve_after_refill.py > gpurun_out/${TAG}.log 2>&1
python - <<PY
import sqlite3
db=sqlite3.connect('gpurun_out/${TAG}_prof/t_results.db'); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
seq=[]; cur_solve=[]
for st,en,n in rows:
    if 'k_eliminate_grouped' in n:
        if 'true' in n or 'ILb1' in n:
            if cur_solve: seq.append(cur_solve)
            cur_solve=[]
        cur_solve.append((en-st)/1e3)
    elif 'k_backsub' in n and cur_solve is not None:
        cur_solve.append(-(en-st)/1e3)
if cur_solve: seq.append(cur_solve)
for k,s in enumerate(seq[-9:]):
    el=[x for x in s if x>0]; bs=[-x for x in s if x<0]
    print("solve %d: eliminate %s | backsub %s" % (k, " ".join("%.0f"%x for x in el), " ".join("%.0f"%x for x in bs)))
PY
