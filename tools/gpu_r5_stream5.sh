#!/bin/bash
# streamed leg after the packer's host-side diet: refill split per batch at 16 / 12 / 8 host threads, the bench's streamed block, the stream tests
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_stream5}
for T in 16 12 8; do
  SLSLAM_REFILL_TIMING=1 timeout 600 python tools/stream_probe.py --batches 16 --host-threads $T > gpurun_out/${TAG}_probe_t$T.txt 2>&1
  echo "threads $T"; grep "slslam refill" gpurun_out/${TAG}_probe_t$T.txt | tail -3; tail -1 gpurun_out/${TAG}_probe_t$T.txt | cut -c1-200
done
bash tools/pack_bench.sh > gpurun_out/${TAG}_pack_bench.txt 2>&1; grep "distinct" gpurun_out/${TAG}_pack_bench.txt
(timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -x -q 2>&1 | tail -3) > gpurun_out/${TAG}_tests.log; cat gpurun_out/${TAG}_tests.log
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-run --no-extra-configs > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
j=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
print('value', j['value'], 'ms', j['ms_per_step']); print(json.dumps(j.get('streamed'))[:1200])
PY
