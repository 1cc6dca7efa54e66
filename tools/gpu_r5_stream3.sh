#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_stream3}
SLSLAM_REFILL_TIMING=1 timeout 600 python tools/stream_probe.py --batches 8 > gpurun_out/${TAG}_probe.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/${TAG}_tr -o t -- python tools/stream_probe.py --batches 4 > gpurun_out/${TAG}_tr.log 2>&1
ls -la gpurun_out/${TAG}_tr/* | head; 
python - <<'PY'
import sqlite3, glob, sys
db = glob.glob('gpurun_out/r5_stream3_tr/*.db') + glob.glob('gpurun_out/r5_stream3_tr/*/*.db')
print(db)
if db:
    c = sqlite3.connect(db[0])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    print([t for t in tabs if 'copy' in t.lower() or 'kernel' in t.lower()][:20])
PY
tail -25 gpurun_out/${TAG}_probe.txt
