#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6_trace}
rm -rf gpurun_out/${TAG}_prof
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/${TAG}_prof -o t -- python tools/stream_probe.py --batches 8 --mode ${2:-pinned} --host-threads 1 > gpurun_out/${TAG}_prof.log 2>&1
python tools/rocpd_timeline.py gpurun_out/${TAG}_prof/t_results.db > gpurun_out/${TAG}_timeline.txt; tail -60 gpurun_out/${TAG}_timeline.txt
