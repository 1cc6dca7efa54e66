#!/bin/bash
# round 6: the streamed leg with the build stage on the device - kernel timeline (rocprofv3) and per-call host timings, three modes
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r6_stream}
export GPU_MAX_HW_QUEUES=8
for MODE in pinned pageable; do
  SLSLAM_REFILL_TIMING=1 timeout 600 python tools/stream_probe.py --batches 12 --mode $MODE --host-threads 2 > gpurun_out/${TAG}_probe_$MODE.txt 2>&1
  echo "mode $MODE"; tail -4 gpurun_out/${TAG}_probe_$MODE.txt | cut -c1-300
done
rm -rf gpurun_out/${TAG}_prof
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o t -- python tools/stream_probe.py --batches 8 --mode pinned --host-threads 1 > gpurun_out/${TAG}_prof.log 2>&1
find gpurun_out/${TAG}_prof -name "*kernel_stats*" | head -2
python - <<PY
import glob, csv
for f in glob.glob('gpurun_out/${TAG}_prof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), ('%.1f' % (float(r['AverageNs'])/1e3)).rjust(10), 'us avg', ('%.1f' % (float(r['TotalDurationNs'])/1e6)).rjust(9), 'ms total')
PY
