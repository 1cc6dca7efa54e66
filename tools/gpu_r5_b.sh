#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r5_b}
bash tools/pack_bench.sh > gpurun_out/${TAG}_pack_bench.txt 2>&1
(timeout 900 python -m pytest tests/test_gpu_lba.py -m gpu -q -x -s -k "mixed_precision" 2>&1 | tail -25) > gpurun_out/${TAG}_mixed.log
(timeout 900 python -m pytest tests/test_gpu_stream.py -m gpu -q -x 2>&1 | tail -8) > gpurun_out/${TAG}_streamtests.log
(timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -q -x -k "c_level" 2>&1 | tail -15) > gpurun_out/${TAG}_disttest.log
timeout 600 python tools/make_bench_digest.py > gpurun_out/${TAG}_digest.log 2>&1; cp tests/golden/bench_digest.json gpurun_out/${TAG}_bench_digest.json
SLSLAM_REFILL_TIMING=1 timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-run > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cat gpurun_out/${TAG}_pack_bench.txt; cat gpurun_out/${TAG}_mixed.log; tail -4 gpurun_out/${TAG}_streamtests.log; cat gpurun_out/${TAG}_disttest.log; tail -3 gpurun_out/${TAG}_digest.log; tail -6 gpurun_out/${TAG}_bench.err
python - <<PY
import json
j=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
print('value', j['value'], 'ms', j['ms_per_step'], j['kernel_ms_per_step'])
print(json.dumps(j.get('streamed'), indent=1)); print(json.dumps(j.get('mixed_precision'), indent=1))
PY
