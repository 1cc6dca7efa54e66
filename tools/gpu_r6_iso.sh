#!/bin/bash
# round 6: the device build's kernels in isolation (a depth-1 stream: nothing overlaps)
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6_iso}
export GPU_MAX_HW_QUEUES=8
rm -rf gpurun_out/${TAG}_prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o t -- python tools/stream_probe.py --batches 6 --depth 1 --mode pinned --host-threads 1 > gpurun_out/${TAG}_prof.log 2>&1
tail -3 gpurun_out/${TAG}_prof.log | cut -c1-300
python tools/rocpd_summary.py gpurun_out/${TAG}_prof/t_results.db > gpurun_out/${TAG}_kernels.txt 2>&1; head -24 gpurun_out/${TAG}_kernels.txt
