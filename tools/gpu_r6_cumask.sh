#!/bin/bash
# steady batch period of the streamed leg (packed indices: the host is out of the way) with the build stream confined to n compute units
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=8
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for n in 0 128 64 32 16 0; do
  echo "SLSLAM_BUILD_CUS=$n"
  SLSLAM_BUILD_CUS=$n timeout 300 python tools/stream_probe.py --mode packed --batches 24 --host-threads 1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ms_per_batch %.2f  steady %.2f' % (d['ms_per_batch'], d['steady_ms_per_batch']))"
done 2>&1 | tee gpurun_out/r6_cumask.txt
