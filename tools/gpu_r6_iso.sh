#!/bin/bash
# the streamed leg's kernels alone (depth-1 stream: nothing overlaps) + the steady period of the depth-3 stream
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=8
TAG=${1:-r6_iso}
rm -rf gpurun_out/${TAG}_iso
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_iso -o t -- python tools/stream_probe.py --batches 6 --depth 1 --mode pinned --host-threads 1 > gpurun_out/${TAG}_iso.log 2>&1
python tools/rocpd_summary.py gpurun_out/${TAG}_iso/t_results.db > gpurun_out/${TAG}_stream_kernels_alone.txt 2>&1
rm -rf gpurun_out/${TAG}_iso
grep "k_build\|k_ingest\|k_reset\|k_export\|^   \|^ " gpurun_out/${TAG}_stream_kernels_alone.txt | cut -c1-150
for MODE in packed pinned; do timeout 300 python tools/stream_probe.py --mode $MODE --batches 24 --host-threads 1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$MODE ms_per_batch %.2f  steady %.2f' % (d['ms_per_batch'], d['steady_ms_per_batch']))"; done
